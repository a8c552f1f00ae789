// oracle/island_ref.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of Bullet's constraint solver for one simulation island of SEVERAL Dynamic bodies: the contacts of Dynamic
// boxes with each other on top of what boxbox_ref.h solves for one body (the reference steps a btDiscreteDynamicsWorld with
// btSequentialImpulseConstraintSolver, src/physics/PhysicsSystem.cpp:122-131; every RigidBody is a btRigidBody in that world,
// :421-474 — two Dynamic boxes of a scene collide with each other there).  Bullet's source is not under /root/reference;
// restated from its PUBLISHED source and then put into the COMPILED form of the reference's exe:
//   BulletDynamics/ConstraintSolver/btSequentialImpulseConstraintSolver.cpp
//       convertContact (rel_pos1/2, getVelocityInLocalPointNoDelta of both bodies, the lateral direction),
//       setupContactConstraint / setupFrictionConstraint with rb0 AND rb1 (torqueAxis1 = rel_pos2 x -n, angularComponentB,
//       denom0 + denom1, vel1Dotn + vel2Dotn, warm start into both solver bodies), solveGroupCacheFriendlySplitImpulseIterations,
//       solveSingleIteration (all contact rows, then all friction rows with the limits of their contact row),
//       solveGroupCacheFriendlyFinish (write-back, split-impulse transform)
//   BulletDynamics/ConstraintSolver/btSolverBody.h   internalApplyImpulse / internalApplyPushImpulse for both bodies of a row
// PARITY STATUS: unpinned in the strict sense (boxbox_ref.h's header).  The two-body forms of setupContactConstraint,
// setupFrictionConstraint and convertContact's loop body are read off the exe by oracle/tools/check_solver_setup.py (paths
// "two rigid bodies": the association of every sum below is the compiled one); the row solvers' body-B side is the same four
// dpps / four fused multiply-adds check_solver_rows.py counts.  With `b < 0` (the other body is a fixed solver body) every
// expression is boxbox_ref.h's SolveBody of round 3 — which is now a wrapper around this function — bit for bit.
// Specification choices (Bullet's own order is history: manifold pool order with swap-removal, then an unstable quickSort by
// island id): bodies of an island in ascending entity id; a pair's body A is the lower entity id; the island's manifolds by
// owner — for every body in turn its plane manifold, its manifolds with Static / Kinematic boxes (ascending entity id), then
// its manifolds with Dynamic boxes of HIGHER entity id (ascending).
#pragma once

#include <vector>

#include "boxbox_ref.h"

namespace orc {
namespace ct {

struct IslandBody {
    BodyState* state = nullptr; // in / out
    float invMass = 0.0f;
    Vec3 invInertiaLocal{0, 0, 0};
    Vec3 force{0, 0, 0};        // m_totalForce (gravity / invMass; zero for a body that was asleep when applyGravity ran)
    bool moved = false;         // out: the split impulse corrected the pose
};

struct IslandPoint {
    Vec3 worldA, worldB, normal;
    float distance;
    float* applied;
    float* appliedLat;
};

struct IslandManifold {
    int a = 0, b = -1; // indices into the island's bodies; b < 0: a static object (the fixed solver body)
    float friction = 0.0f, restitution = 0.0f;
    int n = 0;
    IslandPoint p[4];
};

namespace isl {

struct Row2 : SolverRow {
    int a = 0, b = -1;
    Vec3 normal2{0, 0, 0};       // m_contactNormal2 = -normal
    Vec3 relpos2CrossN{0, 0, 0}; // m_relpos2CrossNormal
    Vec3 angularCompB{0, 0, 0};  // m_angularComponentB
};

inline float Dpps(const Vec3& u, const Vec3& v) { return (u.x * v.x + u.y * v.y) + u.z * v.z; }

// gResolveSingleConstraintRow{LowerLimit,Generic}_sse4_1_fma3 (contact_ref.h ResolveRow) with both bodies
inline void ResolveRow2(SolverBody* sb, Row2& c, bool withUpperLimit)
{
    SolverBody& a = sb[c.a];
    float deltaImpulse = c.rhs - c.applied * c.cfm;
    const float dv1 = Dpps(c.relposCrossN, a.dAng) + Dpps(c.normal, a.dLin);
    const float dv2 = c.b >= 0 ? Dpps(c.normal2, sb[c.b].dLin) + Dpps(c.relpos2CrossN, sb[c.b].dAng) : 0.0f + 0.0f;
    deltaImpulse = Fma(-dv1, c.jacDiagABInv, deltaImpulse);
    deltaImpulse = Fma(-dv2, c.jacDiagABInv, deltaImpulse);
    const float sum = c.applied + deltaImpulse;
    if (c.lower < sum) {
        if (withUpperLimit && !(sum < c.upper)) {
            deltaImpulse = c.upper - c.applied;
            c.applied = c.upper;
        } else {
            c.applied = sum;
        }
    } else {
        deltaImpulse = c.lower - c.applied;
        c.applied = c.lower;
    }
    a.dLin = V(Fma(c.normal.x * a.invMass.x, deltaImpulse, a.dLin.x), Fma(c.normal.y * a.invMass.y, deltaImpulse, a.dLin.y),
               Fma(c.normal.z * a.invMass.z, deltaImpulse, a.dLin.z));
    a.dAng = V(Fma(c.angularComp.x, deltaImpulse, a.dAng.x), Fma(c.angularComp.y, deltaImpulse, a.dAng.y), Fma(c.angularComp.z, deltaImpulse, a.dAng.z));
    if (c.b >= 0) {
        SolverBody& b = sb[c.b];
        b.dLin = V(Fma(c.normal2.x * b.invMass.x, deltaImpulse, b.dLin.x), Fma(c.normal2.y * b.invMass.y, deltaImpulse, b.dLin.y),
                   Fma(c.normal2.z * b.invMass.z, deltaImpulse, b.dLin.z));
        b.dAng = V(Fma(c.angularCompB.x, deltaImpulse, b.dAng.x), Fma(c.angularCompB.y, deltaImpulse, b.dAng.y), Fma(c.angularCompB.z, deltaImpulse, b.dAng.z));
    }
}

// gResolveSplitPenetrationImpulse_sse2 (contact_ref.h ResolveSplitPenetration) with both bodies
inline void ResolveSplitPenetration2(SolverBody* sb, Row2& c)
{
    if (!c.rhsPenetration) return;
    SolverBody& a = sb[c.a];
    auto dot3 = [](const Vec3& u, const Vec3& v) { return u.x * v.x + (u.y * v.y + u.z * v.z); };
    float deltaImpulse = c.rhsPenetration - c.appliedPush * c.cfm;
    const float dv1 = dot3(c.normal, a.push) + dot3(c.relposCrossN, a.turn);
    const float dv2 = c.b >= 0 ? dot3(c.normal2, sb[c.b].push) + dot3(c.relpos2CrossN, sb[c.b].turn) : 0.0f + 0.0f;
    deltaImpulse = deltaImpulse - dv1 * c.jacDiagABInv;
    deltaImpulse = deltaImpulse - dv2 * c.jacDiagABInv;
    const float sum = c.appliedPush + deltaImpulse;
    if (sum < c.lower) {
        deltaImpulse = c.lower - c.appliedPush;
        c.appliedPush = c.lower;
    } else {
        c.appliedPush = sum;
    }
    const Vec3 lin = V(c.normal.x * a.invMass.x, c.normal.y * a.invMass.y, c.normal.z * a.invMass.z);
    a.push = Add(a.push, Scale(lin, deltaImpulse));
    a.turn = Add(a.turn, Scale(c.angularComp, deltaImpulse));
    if (c.b >= 0) {
        SolverBody& b = sb[c.b];
        const Vec3 lin2 = V(c.normal2.x * b.invMass.x, c.normal2.y * b.invMass.y, c.normal2.z * b.invMass.z);
        b.push = Add(b.push, Scale(lin2, deltaImpulse));
        b.turn = Add(b.turn, Scale(c.angularCompB, deltaImpulse));
    }
}

} // namespace isl

// solveGroup for one island.  btContactSolverInfo as the reference's exe constructs it (boxbox_ref.h).
inline void SolveIsland(IslandBody* bodies, int nBodies, IslandManifold* manifolds, int nManifolds, float dt)
{
    constexpr int kIterations = 10;
    constexpr float kErp2 = 0.2f;
    constexpr float kSplitThreshold = -0.04f;
    constexpr float kSplitTurnErp = 0.1f;
    constexpr float kWarmstart = 0.85f;
    constexpr float kSor = 1.0f;
    constexpr float kRestitutionVelocityThreshold = 0.2f;

    // convertBodies
    std::vector<SolverBody> sb(static_cast<size_t>(nBodies));
    std::vector<Mat3> invI(static_cast<size_t>(nBodies));
    for (int i = 0; i < nBodies; ++i) {
        const IslandBody& ib = bodies[i];
        const BodyState& b = *ib.state;
        invI[i] = InvInertiaWorld(b.basis, ib.invInertiaLocal);
        sb[i].invMass = V(ib.invMass, ib.invMass, ib.invMass);
        sb[i].linVel = b.linVel;
        sb[i].angVel = b.angVel;
        sb[i].extForce = Scale(Scale(ib.force, ib.invMass), dt);
        sb[i].extTorque = V(0.0f, 0.0f, 0.0f);
        sb[i].extTorque = Add(sb[i].extTorque, GyroscopicImpulse(ib.invInertiaLocal, b.angVel, b.orn, dt));
    }

    int nRows = 0;
    for (int k = 0; k < nManifolds; ++k) nRows += manifolds[k].n;
    std::vector<isl::Row2> normalRow(static_cast<size_t>(nRows)), frictionRow(static_cast<size_t>(nRows));
    std::vector<const IslandPoint*> ref(static_cast<size_t>(nRows));
    const float invTimeStep = 1.0f / dt;
    int j = 0;
    for (int k = 0; k < nManifolds; ++k) {
        const IslandManifold& m = manifolds[k];
        const bool two = m.b >= 0;
        for (int q = 0; q < m.n; ++q, ++j) {
            const IslandPoint& cp = m.p[q];
            ref[j] = &cp;
            const Vec3 n = cp.normal;
            SolverBody& A = sb[m.a];
            const BodyState& sa = *bodies[m.a].state;
            const float invMassA = bodies[m.a].invMass;
            isl::Row2& c = normalRow[j];
            c = isl::Row2{};
            c.a = m.a;
            c.b = m.b;
            const Vec3 rel_pos1 = Sub(cp.worldA, sa.origin);
            const Vec3 vel1 = Add(Add(A.linVel, A.extForce), Cross(Add(A.angVel, A.extTorque), rel_pos1)); // getVelocityInLocalPointNoDelta
            Vec3 rel_pos2 = V(0.0f, 0.0f, 0.0f), vel2 = V(0.0f, 0.0f, 0.0f);
            if (two) {
                const SolverBody& B = sb[m.b];
                rel_pos2 = Sub(cp.worldB, bodies[m.b].state->origin);
                vel2 = Add(Add(B.linVel, B.extForce), Cross(Add(B.angVel, B.extTorque), rel_pos2));
            }
            const Vec3 vel = Sub(vel1, vel2);
            const float rel_vel = Dot(n, vel);
            const float relaxation = kSor;
            const Vec3 torqueAxis0 = Cross(rel_pos1, n);
            c.angularComp = MatVec(invI[m.a], torqueAxis0);
            Vec3 torqueAxis1 = V(0.0f, 0.0f, 0.0f);
            {
                const Vec3 vec = Cross(c.angularComp, rel_pos1);
                const float denom0 = InvMassPlusDot(invMassA, n, vec);
                float denom1 = 0.0f;
                if (two) {
                    torqueAxis1 = Cross(n, rel_pos2); // rel_pos2 x -n
                    c.angularCompB = MatVec(invI[m.b], torqueAxis1);
                    denom1 = InvMassPlusDot(bodies[m.b].invMass, n, Cross(rel_pos2, c.angularCompB)); // n . (-angularComponentB x rel_pos2)
                }
                const float cfm0 = 0.0f * invTimeStep;
                c.jacDiagABInv = relaxation / (denom0 + denom1 + cfm0);
            }
            c.normal = n;
            c.relposCrossN = torqueAxis0;
            c.normal2 = V(-n.x, -n.y, -n.z);
            c.relpos2CrossN = torqueAxis1;
            const float penetration = cp.distance + 0.0f;
            c.friction = m.friction;
            // setupContactConstraint's own relative velocity: the rigid bodies', without the external force impulse
            float restitution = 0.0f;
            if (m.restitution != 0.0f) {
                const Vec3 rbVel1 = Add(sa.linVel, Cross(sa.angVel, rel_pos1)); // rb0->getVelocityInLocalPoint(rel_pos1)
                Vec3 rbVel2 = V(0.0f, 0.0f, 0.0f);
                if (two) rbVel2 = Add(bodies[m.b].state->linVel, Cross(bodies[m.b].state->angVel, rel_pos2));
                const float rbRelVel = Dot(n, Sub(rbVel1, rbVel2));
                restitution = std::fabs(rbRelVel) < kRestitutionVelocityThreshold ? 0.0f : m.restitution * -rbRelVel; // restitutionCurve
                if (restitution <= 0.0f) restitution = 0.0f;
            }
            c.applied = *cp.applied * kWarmstart;
            {
                const Vec3 lin = V(c.normal.x * A.invMass.x, c.normal.y * A.invMass.y, c.normal.z * A.invMass.z);
                A.dLin = Add(A.dLin, Scale(lin, c.applied));
                A.dAng = Add(A.dAng, Scale(c.angularComp, c.applied * 1.0f));
                if (two) {
                    SolverBody& B = sb[m.b];
                    const Vec3 linB = V(B.invMass.x * n.x, B.invMass.y * n.y, B.invMass.z * n.z);
                    B.dLin = Sub(B.dLin, Scale(linB, c.applied));
                    B.dAng = Add(B.dAng, Scale(c.angularCompB, c.applied * 1.0f));
                }
            }
            c.appliedPush = 0.0f;
            {
                const float vel1Dotn = DotXZY(c.normal, Add(A.linVel, A.extForce)) + DotXZY(c.relposCrossN, Add(A.angVel, A.extTorque));
                float vel2Dotn = 0.0f + 0.0f;
                if (two) {
                    const SolverBody& B = sb[m.b];
                    const Vec3 l = Add(B.linVel, B.extForce);
                    vel2Dotn = DotXZY(c.relpos2CrossN, Add(B.angVel, B.extTorque)) + ((-(l.x * n.x) - l.z * n.z) - l.y * n.y);
                }
                const float rel_vel2 = vel1Dotn + vel2Dotn;
                float positionalError = 0.0f;
                float velocityError = restitution - rel_vel2;
                if (penetration > 0.0f) {
                    positionalError = 0.0f;
                    velocityError -= penetration * invTimeStep;
                } else {
                    positionalError = -penetration * kErp2 * invTimeStep;
                }
                const float penetrationImpulse = positionalError * c.jacDiagABInv;
                const float velocityImpulse = velocityError * c.jacDiagABInv;
                if (penetration > kSplitThreshold) {
                    c.rhs = penetrationImpulse + velocityImpulse;
                    c.rhsPenetration = 0.0f;
                } else {
                    c.rhs = velocityImpulse;
                    c.rhsPenetration = penetrationImpulse;
                }
                c.cfm = 0.0f * c.jacDiagABInv;
                c.lower = 0.0f;
                c.upper = 1e10f;
            }
            Vec3 dir = Sub(vel, Scale(n, rel_vel));
            const float lat_rel_vel = Dot(dir, dir);
            if (lat_rel_vel > bt::kEpsilon) {
                dir = Scale(dir, 1.0f / std::sqrt(lat_rel_vel));
            } else {
                dir = PlaneSpace1(n); // (for the plane's (0, 1, 0): (-1, 0, 0), contact_ref.h's FallbackFrictionDir)
            }
            isl::Row2& f = frictionRow[j];
            f = isl::Row2{};
            f.a = m.a;
            f.b = m.b;
            f.friction = m.friction;
            f.normal = dir;
            f.normal2 = V(-dir.x, -dir.y, -dir.z);
            f.relposCrossN = Cross(rel_pos1, dir);
            f.angularComp = MatVec(invI[m.a], f.relposCrossN);
            {
                const Vec3 vec = Cross(f.angularComp, rel_pos1);
                const float denom0 = InvMassPlusDot(invMassA, dir, vec);
                float denom1 = 0.0f;
                if (two) {
                    f.relpos2CrossN = Cross(dir, rel_pos2);
                    f.angularCompB = MatVec(invI[m.b], f.relpos2CrossN);
                    denom1 = InvMassPlusDot(bodies[m.b].invMass, dir, Cross(rel_pos2, f.angularCompB));
                }
                f.jacDiagABInv = relaxation / (denom0 + denom1);
            }
            {
                const float vel1Dotn = DotXZY(f.normal, Add(A.linVel, A.extForce)) + DotXZY(f.relposCrossN, A.angVel);
                float rv;
                if (two) {
                    const SolverBody& B = sb[m.b];
                    const Vec3 l = Add(B.linVel, B.extForce);
                    rv = DotXZY(f.relpos2CrossN, B.angVel) + ((vel1Dotn - l.z * dir.z) + (-(l.x * dir.x) - l.y * dir.y));
                } else {
                    const float vel2Dotn = 0.0f + 0.0f;
                    rv = vel1Dotn + vel2Dotn;
                }
                const float velocityError = 0.0f - rv;
                const float velocityImpulse = velocityError * f.jacDiagABInv;
                f.rhs = 0.0f + velocityImpulse;
                f.rhsPenetration = 0.0f;
                f.cfm = 0.0f;
                f.lower = -f.friction;
                f.upper = f.friction;
            }
            f.applied = 0.0f; // setFrictionConstraintImpulse of this Bullet: frictionConstraint1.m_appliedImpulse = 0.f, no warm start
        }
    }
    for (int it = 0; it < kIterations; ++it) {
        for (int r = 0; r < nRows; ++r) isl::ResolveSplitPenetration2(sb.data(), normalRow[r]);
    }
    for (int it = 0; it < kIterations; ++it) {
        for (int r = 0; r < nRows; ++r) isl::ResolveRow2(sb.data(), normalRow[r], false);
        for (int r = 0; r < nRows; ++r) {
            const float totalImpulse = normalRow[r].applied;
            if (totalImpulse > 0.0f) {
                frictionRow[r].lower = -(frictionRow[r].friction * totalImpulse);
                frictionRow[r].upper = frictionRow[r].friction * totalImpulse;
                isl::ResolveRow2(sb.data(), frictionRow[r], true);
            }
        }
    }
    for (int r = 0; r < nRows; ++r) {
        *ref[r]->applied = normalRow[r].applied;
        *ref[r]->appliedLat = frictionRow[r].applied;
    }
    for (int i = 0; i < nBodies; ++i) {
        SolverBody& s = sb[i];
        BodyState& b = *bodies[i].state;
        s.linVel = Add(s.linVel, s.dLin);
        s.angVel = Add(s.angVel, s.dAng);
        bodies[i].moved = false;
        if (s.push.x != 0.0f || s.push.y != 0.0f || s.push.z != 0.0f || s.turn.x != 0.0f || s.turn.y != 0.0f || s.turn.z != 0.0f) {
            b.origin = Add(b.origin, Scale(s.push, dt));
            b.orn = bt::IntegrateOrientation(b.orn, Scale(s.turn, kSplitTurnErp), dt);
            b.basis = bt::MatFromQuat(b.orn);
            bodies[i].moved = true;
        }
        b.linVel = Add(s.linVel, s.extForce);
        b.angVel = Add(s.angVel, s.extTorque);
    }
}

// the rows of a body's own manifolds (plane, then Static / Kinematic boxes in ascending entity id) as island manifolds of body `a`
inline void AppendOwnManifolds(std::vector<IslandManifold>& out, int a, Manifold* ground, BoxManifold* boxes, int nBoxes, float bodyFriction)
{
    if (ground) {
        IslandManifold m;
        m.a = a;
        m.friction = std::max(-10.0f, std::min(10.0f, bodyFriction * 1.0f));
        m.restitution = 0.0f;
        m.n = ground->n;
        for (int j = 0; j < ground->n; ++j) {
            ContactPoint& cp = ground->p[j];
            m.p[j] = IslandPoint{cp.worldA, V(0.0f, 0.0f, 0.0f), V(0.0f, 1.0f, 0.0f), cp.distance, &cp.appliedImpulse, &cp.appliedImpulseLateral1};
        }
        out.push_back(m);
    }
    for (int k = 0; k < nBoxes; ++k) {
        IslandManifold m;
        m.a = a;
        m.friction = boxes[k].friction;
        m.restitution = boxes[k].restitution;
        m.n = boxes[k].n;
        for (int j = 0; j < boxes[k].n; ++j) {
            BoxPoint& cp = boxes[k].p[j];
            m.p[j] = IslandPoint{cp.worldA, cp.worldB, cp.normalB, cp.distance, &cp.appliedImpulse, &cp.appliedImpulseLateral1};
        }
        out.push_back(m);
    }
}

// solveGroup for the island {body}: its manifold with the ground plane (when the plane is on) followed by its manifolds with
// Static / Kinematic boxes, in ascending entity id.  The other body of every row is a fixed solver body (static and kinematic
// objects share the zero-velocity one: the reference never gives a Kinematic body a velocity, it teleports it), so its side of
// every row contributes exactly zero.  btContactSolverInfo as the reference's exe constructs it (VA 0x1401b8b5b: tau 0.6 ...
// m_restitutionVelocityThreshold 0.2 at +0x108 of the world — the field exists, so the build is bullet3 >= 2.88).
inline bool SolveBody(BodyState& b, Manifold* ground, BoxManifold* boxes, int nBoxes, float invMassScalar, const Vec3& invInertiaLocal,
                      float bodyFriction, const Vec3& force, float dt)
{
    IslandBody ib;
    ib.state = &b;
    ib.invMass = invMassScalar;
    ib.invInertiaLocal = invInertiaLocal;
    ib.force = force;
    std::vector<IslandManifold> ms;
    AppendOwnManifolds(ms, 0, ground, boxes, nBoxes, bodyFriction);
    SolveIsland(&ib, 1, ms.data(), static_cast<int>(ms.size()), dt);
    return ib.moved;
}

} // namespace ct
} // namespace orc
