// oracle/contact_ref.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of what Bullet does, inside stepSimulation (src/physics/PhysicsSystem.cpp:863), for ONE Dynamic body
// and the static ground plane the reference adds to every world (src/physics/PhysicsSystem.cpp:149-166:
// btStaticPlaneShape((0,1,0), 0), friction 1, restitution 0, group StaticFilter, mask AllFilter) — SURVEY.md §8(f) rank 4.
// Bodies do not collide with each other here (no convex-convex narrowphase): every Dynamic body is an island of its own
// whose only manifold is the one with the plane, so the whole step is independent per body.
//
// Bullet's source is not under /root/reference; the functions below restate its PUBLISHED algorithm (bullet3 file
// names given per function) in the order internalSingleStepSimulation runs them:
//   performDiscreteCollisionDetection   btConvexPlaneCollisionAlgorithm::processCollision: collideSingleContact (one contact
//                                       per step: the convex shape's support vertex against the plane; the perturbed
//                                       multi-point queries are off: btDefaultCollisionConfiguration creates the algorithm
//                                       with minimumPointsPerturbationThreshold 0), btManifoldResult::addContactPoint,
//                                       btPersistentManifold (4-point cache: getCacheEntry / replaceContactPoint /
//                                       addManifoldPoint / sortCachedPoints), then refreshContactPoints
//   solveConstraints                    btSequentialImpulseConstraintSolver, default btContactSolverInfo (10 iterations,
//                                       erp2 0.2, split impulse below -0.04 with turn erp 0.1, warm starting 0.85, one
//                                       velocity-dependent friction direction): convertBodies (external force impulse,
//                                       implicit gyroscopic impulse), convertContact / setupContactConstraint /
//                                       setupFrictionConstraint / setFrictionConstraintImpulse, split-impulse iterations,
//                                       velocity iterations, write-back
//   integrateTransforms / updateActivationState stay in physics_ref.h (the free-body step), fed with the solver's velocities.
//
// PARITY STATUS: "parity unpinned" (Bullet is an unpinned third-party dependency and the reference holds no fixtures).
// Pinned from the reference's committed build/bin/RelWithDebInfo/SandboxCity.exe, read as bytes with objdump:
//   * the two row solvers the velocity iterations call on any x86-64 CPU with FMA3 + SSE4.1 (the solver's constructor
//     selects them at run time): gResolveSingleConstraintRowLowerLimit_sse4_1_fma3 at VA 0x1401c9230 and
//     gResolveSingleConstraintRowGeneric_sse4_1_fma3 at VA 0x1401c8ae0 — dpps 0x7f dot products ((x·x' + y·y') + z·z'),
//     deltaImpulse = fnmadd(deltaVelDotn, jacDiagABInv, rhs - appliedImpulse·cfm), the limit selection by blendvps
//     (lower < sum, sum < upper), velocity updates by fmadd(normal·invMass, deltaImpulse, deltaLinearVelocity) — ResolveRow
//     below follows that instruction sequence (and with it the btSolverConstraint / btSolverBody field layout);
//   * the split-impulse row Bullet installs when USE_SIMD is defined (MSVC x64), gResolveSplitPenetrationImpulse_sse2 at
//     VA 0x1401c9790 (the one that bumps gNumSplitImpulseRecoveries and reads m_rhsPenetration at +0x98): mulps + shufps
//     dot products summed (z·z' + y·y') + x·x', linear dot + angular dot, deltaImpulse = ((rhsPenetration − push·cfm) −
//     dv1·jacDiagABInv) − dv2·jacDiagABInv, the lower-limit select by cmpltps/andps/andnps, velocity updates as separate
//     mulps then addps ((normal·invMass)·deltaImpulse + pushVelocity; no FMA) — ResolveSplitPenetration below follows it
//     (the exe's scalar variant at VA 0x1401c9380 is not the one installed);
//   * contraction: a scan of the whole exe's disassembly finds exactly 12 fused multiply-adds (vfmadd / vfnmadd), all
//     inside those two _sse4_1_fma3 row solvers, and no other VEX-encoded arithmetic — every other float operation of the
//     engine, glm and Bullet in that build is an unfused SSE mul/add/sub/div, which is what -ffp-contract=off restates.
// Everything else is from the published source with scalar left-to-right arithmetic; the compiled operation order of
// setupContactConstraint, the manifold functions and computeGyroscopicImpulseImplicit_Body has NOT been checked.
//
// Simplifications (stated, not hidden): the world inverse inertia tensor is rebuilt from the pose at solve time (Bullet
// keeps the one of the last integrateTransforms: different only in the sub-step right after a teleport of a body that is
// in contact); contact life time counters are not kept (nothing reads them).
#pragma once

#include <algorithm>
#include <cmath>

#include "bullet_math.h"

namespace orc {
namespace ct {

using bt::Mat3;
using bt::Quat;
using bt::Vec3;

inline Vec3 V(float x, float y, float z) { return Vec3{x, y, z}; }
inline Vec3 Add(const Vec3& a, const Vec3& b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 Sub(const Vec3& a, const Vec3& b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 Scale(const Vec3& a, float s) { return V(a.x * s, a.y * s, a.z * s); }
inline float Dot(const Vec3& a, const Vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // btVector3::dot, scalar
inline Vec3 Cross(const Vec3& a, const Vec3& b)
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); // btVector3::cross
}
// btMatrix3x3 * btVector3: (row0.dot(v), row1.dot(v), row2.dot(v))
inline Vec3 MatVec(const Mat3& m, const Vec3& v)
{
    return V(m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z, m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z,
             m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z);
}
// btMatrix3x3::transpose() * v, as btTransform::invXform forms it: (col0.dot(v), col1.dot(v), col2.dot(v))
inline Vec3 MatTVec(const Mat3& m, const Vec3& v)
{
    return V(m.m[0][0] * v.x + m.m[1][0] * v.y + m.m[2][0] * v.z, m.m[0][1] * v.x + m.m[1][1] * v.y + m.m[2][1] * v.z,
             m.m[0][2] * v.x + m.m[1][2] * v.y + m.m[2][2] * v.z);
}

// What the reference's colliders are in Bullet (src/physics/PhysicsSystem.cpp:686-707): btBoxShape(halfExtents) with the
// safe margin of bullet_math.h BoxAabbHalfExtents, or btCapsuleShape(radius, 2·halfHeight) (up axis Y, margin = radius).
struct Shape {
    bool capsule = false;
    Vec3 dims{0.5f, 0.5f, 0.5f}; // box: getHalfExtentsWithMargin(); capsule: (radius, halfHeight, radius) = m_implicitShapeDimensions
};

// btBoxShape::calculateLocalInertia / btCapsuleShape::calculateLocalInertia (BulletCollision/CollisionShapes)
inline Vec3 LocalInertia(const Shape& s, float mass)
{
    if (s.capsule) {
        const float radius = s.dims.x;
        const float hx = radius, hy = radius + s.dims.y, hz = radius; // halfExtents[upAxis] += getHalfHeight()
        const float lx = 2.0f * hx, ly = 2.0f * hy, lz = 2.0f * hz;
        const float x2 = lx * lx, y2 = ly * ly, z2 = lz * lz;
        const float scaledmass = mass * 0.08333333f;
        return V(scaledmass * (y2 + z2), scaledmass * (x2 + z2), scaledmass * (x2 + y2));
    }
    const float lx = 2.0f * s.dims.x, ly = 2.0f * s.dims.y, lz = 2.0f * s.dims.z;
    return V(mass / 12.0f * (ly * ly + lz * lz), mass / 12.0f * (lx * lx + lz * lz), mass / 12.0f * (lx * lx + ly * ly));
}

// btRigidBody::setMassProps: m_invInertiaLocal
inline Vec3 InvInertiaLocal(const Vec3& inertia)
{
    return V(inertia.x != 0.0f ? 1.0f / inertia.x : 0.0f, inertia.y != 0.0f ? 1.0f / inertia.y : 0.0f, inertia.z != 0.0f ? 1.0f / inertia.z : 0.0f);
}

// btRigidBody::updateInertiaTensor: basis.scaled(invInertiaLocal) * basis.transpose()
inline Mat3 InvInertiaWorld(const Mat3& b, const Vec3& il)
{
    Mat3 s; // scaled: column c times il[c]
    for (int r = 0; r < 3; ++r) {
        s.m[r][0] = b.m[r][0] * il.x;
        s.m[r][1] = b.m[r][1] * il.y;
        s.m[r][2] = b.m[r][2] * il.z;
    }
    Mat3 o; // s * b^T: element (r, c) = s.row(r) . b.row(c), accumulated as btMatrix3x3::tdot does (x, y, z in order)
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) o.m[r][c] = s.m[r][0] * b.m[c][0] + s.m[r][1] * b.m[c][1] + s.m[r][2] * b.m[c][2];
    }
    return o;
}

// btCollisionShape::getContactBreakingThreshold(gContactBreakingThreshold) = getAngularMotionDisc() · 0.02 of the convex
// shape (the plane's is astronomically larger; the manifold takes the minimum: btCollisionDispatcher::getNewManifold with
// CD_USE_RELATIVE_CONTACT_BREAKING_THRESHOLD).  getAngularMotionDisc = bounding-sphere radius + |centre| with the sphere
// taken from the identity-transform AABB (btCollisionShape::getBoundingSphere): centre is 0 for both shapes.
inline float ContactBreakingThreshold(const Shape& s)
{
    const float ex = s.capsule ? s.dims.x : s.dims.x;
    const float ey = s.capsule ? s.dims.x + s.dims.y : s.dims.y;
    const float ez = s.capsule ? s.dims.x : s.dims.z;
    const Vec3 mn = V(0.0f - ex, 0.0f - ey, 0.0f - ez), mx = V(0.0f + ex, 0.0f + ey, 0.0f + ez);
    const Vec3 d = Sub(mx, mn);
    const float radius = std::sqrt(Dot(d, d)) * 0.5f;
    const Vec3 c = Scale(Add(mn, mx), 0.5f);
    const float disc = radius + std::sqrt(Dot(c, c));
    return disc * bt::kContactBreakingThreshold;
}

// convexShape->localGetSupportingVertex(dir) (local frame)
//   btBoxShape: per component  dir >= 0 ? +h : -h  with h = halfExtentsWithMargin (btFsels)
//   btCapsuleShape (btConvexInternalShape::localGetSupportingVertex): the segment end point that maximises the dot product
//   with the normalised direction, plus margin (= radius) times the normalised direction
inline Vec3 SupportVertex(const Shape& s, const Vec3& dir)
{
    if (!s.capsule) return V(dir.x >= 0.0f ? s.dims.x : -s.dims.x, dir.y >= 0.0f ? s.dims.y : -s.dims.y, dir.z >= 0.0f ? s.dims.z : -s.dims.z);
    Vec3 vec = dir;
    const float lenSqr = Dot(vec, vec);
    if (lenSqr < 0.0001f) {
        vec = V(1.0f, 0.0f, 0.0f);
    } else {
        const float rlen = 1.0f / std::sqrt(lenSqr);
        vec = Scale(vec, rlen);
    }
    Vec3 sup = V(0.0f, 0.0f, 0.0f);
    float maxDot = -1.0e18f; // -BT_LARGE_FLOAT
    {
        const Vec3 vtx = V(0.0f, s.dims.y, 0.0f);
        const float d = Dot(vec, vtx);
        if (d > maxDot) {
            maxDot = d;
            sup = vtx;
        }
    }
    {
        const Vec3 vtx = V(0.0f, -s.dims.y, 0.0f);
        const float d = Dot(vec, vtx);
        if (d > maxDot) {
            maxDot = d;
            sup = vtx;
        }
    }
    // margin part of localGetSupportingVertex: the direction is normalised again from the ORIGINAL vector
    Vec3 vecnorm = dir;
    if (Dot(vecnorm, vecnorm) < bt::kEpsilon * bt::kEpsilon) vecnorm = V(-1.0f, -1.0f, -1.0f);
    vecnorm = Scale(vecnorm, 1.0f / std::sqrt(Dot(vecnorm, vecnorm))); // btVector3::normalize: *this /= length() -> operator/= multiplies by 1/s
    return Add(sup, Scale(vecnorm, s.dims.x));
}

// btManifoldPoint, the fields that outlive a step
struct ContactPoint {
    Vec3 localA{0, 0, 0};  // on the convex body, body frame
    Vec3 localB{0, 0, 0};  // on the plane, world frame (the plane's transform is the identity)
    float appliedImpulse = 0.0f;
    float appliedImpulseLateral1 = 0.0f;
    // refreshed every step
    Vec3 worldA{0, 0, 0}, worldB{0, 0, 0};
    float distance = 0.0f;
};

struct Manifold {
    int n = 0;
    ContactPoint p[4];
    void Clear() { n = 0; }
};

// btPersistentManifold::sortCachedPoints (gContactCalcArea3Points): which of the four cached points the new one replaces
inline int SortCachedPoints(const Manifold& m, const ContactPoint& pt)
{
    int maxPenetrationIndex = -1;
    float maxPenetration = pt.distance;
    for (int i = 0; i < 4; ++i) {
        if (m.p[i].distance < maxPenetration) {
            maxPenetrationIndex = i;
            maxPenetration = m.p[i].distance;
        }
    }
    float res[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const Vec3& p0 = m.p[0].localA;
    const Vec3& p1 = m.p[1].localA;
    const Vec3& p2 = m.p[2].localA;
    const Vec3& p3 = m.p[3].localA;
    auto len2 = [](const Vec3& v) { return Dot(v, v); };
    if (maxPenetrationIndex != 0) res[0] = len2(Cross(Sub(pt.localA, p1), Sub(p3, p2)));
    if (maxPenetrationIndex != 1) res[1] = len2(Cross(Sub(pt.localA, p0), Sub(p3, p2)));
    if (maxPenetrationIndex != 2) res[2] = len2(Cross(Sub(pt.localA, p0), Sub(p3, p1)));
    if (maxPenetrationIndex != 3) res[3] = len2(Cross(Sub(pt.localA, p0), Sub(p2, p1)));
    // btVector4::closestAxis4 = absolute4().maxAxis4()
    int maxIndex = -1;
    float maxVal = -1.0e18f;
    for (int i = 0; i < 4; ++i) {
        const float a = std::fabs(res[i]);
        if (a > maxVal) {
            maxIndex = i;
            maxVal = a;
        }
    }
    return maxIndex;
}

// btConvexPlaneCollisionAlgorithm::processCollision for plane normal (0,1,0), constant 0, identity plane transform.
// The products with the plane's 0 / 1 components are exact, so they are written out: the direction handed to the support
// function is -(second row of the basis), the distance is the vertex' world y, its projection has y = 0.
inline void CollideWithGround(Manifold& m, const Shape& shape, float breaking, const Vec3& origin, const Mat3& basis)
{
    // collideSingleContact
    const Vec3 dirLocal = V(-basis.m[1][0], -basis.m[1][1], -basis.m[1][2]); // planeInConvex.getBasis() * -planeNormal
    const Vec3 vtx = SupportVertex(shape, dirLocal);
    const Vec3 vtxInPlane = Add(MatVec(basis, vtx), origin); // convexInPlaneTrans(vtx)
    const float distance = vtxInPlane.y;                      // planeNormal.dot(vtxInPlane) - planeConstant
    if (distance < breaking) {
        // btManifoldResult::addContactPoint(normalOnB = (0,1,0), pointInWorld = projection, depth = distance)
        const Vec3 pointInWorld = V(vtxInPlane.x, vtxInPlane.y - distance, vtxInPlane.z); // vtxInPlane - distance * planeNormal
        if (!(distance > breaking)) {
            ContactPoint np;
            const Vec3 pointA = V(pointInWorld.x, pointInWorld.y + distance, pointInWorld.z); // pointInWorld + normalOnB * depth
            np.localA = MatTVec(basis, Sub(pointA, origin));                                   // convex transform .invXform(pointA)
            np.localB = pointInWorld;
            np.worldA = pointA;
            np.worldB = pointInWorld;
            np.distance = distance;
            // getCacheEntry: the cached point whose localA is nearest, within the breaking threshold
            float shortest = breaking * breaking;
            int nearest = -1;
            for (int i = 0; i < m.n; ++i) {
                const Vec3 diffA = Sub(m.p[i].localA, np.localA);
                const float d2 = Dot(diffA, diffA);
                if (d2 < shortest) {
                    shortest = d2;
                    nearest = i;
                }
            }
            if (nearest >= 0) {
                // replaceContactPoint: the applied impulses survive
                np.appliedImpulse = m.p[nearest].appliedImpulse;
                np.appliedImpulseLateral1 = m.p[nearest].appliedImpulseLateral1;
                m.p[nearest] = np;
            } else {
                // addManifoldPoint
                int insert = m.n;
                if (insert == 4) {
                    insert = SortCachedPoints(m, np);
                } else {
                    m.n++;
                }
                if (insert < 0) insert = 0;
                m.p[insert] = np;
            }
        }
    }
    // refreshContactPoints(convex transform, plane transform)
    for (int i = m.n - 1; i >= 0; --i) {
        ContactPoint& c = m.p[i];
        c.worldA = Add(MatVec(basis, c.localA), origin);
        c.worldB = c.localB;
        c.distance = Dot(Sub(c.worldA, c.worldB), V(0.0f, 1.0f, 0.0f));
    }
    for (int i = m.n - 1; i >= 0; --i) {
        ContactPoint& c = m.p[i];
        bool remove = !(c.distance <= breaking); // validContactDistance
        if (!remove) {
            const Vec3 projectedPoint = Sub(c.worldA, Scale(V(0.0f, 1.0f, 0.0f), c.distance));
            const Vec3 projectedDifference = Sub(c.worldB, projectedPoint);
            const float distance2d = Dot(projectedDifference, projectedDifference);
            remove = distance2d > breaking * breaking;
        }
        if (remove) {
            // removeContactPoint: the last point takes the slot
            const int last = m.n - 1;
            if (i != last) m.p[i] = m.p[last];
            m.p[last] = ContactPoint{};
            m.n--;
        }
    }
}

// quatRotate(rotation, v) (LinearMath/btQuaternion.h): q = rotation * v; q *= rotation.inverse(); vector part
inline Quat QuatTimesVec(const Quat& q, const Vec3& w)
{
    return Quat{q.w * w.x + q.y * w.z - q.z * w.y, q.w * w.y + q.z * w.x - q.x * w.z, q.w * w.z + q.x * w.y - q.y * w.x,
                -q.x * w.x - q.y * w.y - q.z * w.z};
}
inline Vec3 QuatRotate(const Quat& rotation, const Vec3& v)
{
    const Quat q = QuatTimesVec(rotation, v);
    const Quat inv{-rotation.x, -rotation.y, -rotation.z, rotation.w};
    const Quat r = bt::QuatMul(q, inv);
    return V(r.x, r.y, r.z);
}

// btMatrix3x3::solve33 (Cramer's rule by cofactors) for J x = b
inline Vec3 Solve33(const Mat3& J, const Vec3& b)
{
    const Vec3 col1 = V(J.m[0][0], J.m[1][0], J.m[2][0]);
    const Vec3 col2 = V(J.m[0][1], J.m[1][1], J.m[2][1]);
    const Vec3 col3 = V(J.m[0][2], J.m[1][2], J.m[2][2]);
    float det = Dot(col1, Cross(col2, col3));
    if (std::fabs(det) > bt::kEpsilon) det = 1.0f / det;
    return V(det * Dot(b, Cross(col2, col3)), det * Dot(col1, Cross(b, col3)), det * Dot(col1, Cross(col2, b)));
}

// btRigidBody::computeGyroscopicImpulseImplicit_Body(step): one Newton step of the implicit gyroscopic term in the body
// frame (BT_ENABLE_GYROSCOPIC_FORCE_IMPLICIT_BODY is a btRigidBody default flag)
inline Vec3 GyroscopicImpulse(const Vec3& idl, const Vec3& omega1, const Quat& q, float step)
{
    const Quat qinv{-q.x, -q.y, -q.z, q.w};
    Vec3 omegab = QuatRotate(qinv, omega1);
    const Vec3 ibo = V(idl.x * omegab.x, idl.y * omegab.y, idl.z * omegab.z); // Ib * omegab, Ib diagonal (the zero products vanish)
    const Vec3 f = Scale(Cross(omegab, ibo), step);
    // skew0 = [omegab]x, skew1 = [Ib omegab]x;  J = Ib + (skew0 * Ib - skew1) * step
    auto skew = [](const Vec3& v) {
        Mat3 s;
        s.m[0][0] = 0.0f; s.m[0][1] = -v.z; s.m[0][2] = v.y;
        s.m[1][0] = v.z; s.m[1][1] = 0.0f; s.m[1][2] = -v.x;
        s.m[2][0] = -v.y; s.m[2][1] = v.x; s.m[2][2] = 0.0f;
        return s;
    };
    const Mat3 s0 = skew(omegab), s1 = skew(ibo);
    Mat3 J;
    const float il[3] = {idl.x, idl.y, idl.z};
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            const float s0Ib = s0.m[r][c] * il[c]; // (skew0 * Ib)(r,c): Ib is diagonal
            const float ib = r == c ? il[c] : 0.0f;
            J.m[r][c] = ib + (s0Ib - s1.m[r][c]) * step;
        }
    }
    const Vec3 omega_div = Solve33(J, f);
    omegab = Sub(omegab, omega_div);
    const Vec3 omega2 = QuatRotate(q, omegab);
    return Sub(omega2, omega1);
}

// btPlaneSpace1((0,1,0)) -> first tangent, evaluated: a = 0*0 + 1*1 = 1, k = 1/sqrt(1) = 1, p = (-n.y*k, n.x*k, 0)
inline Vec3 FallbackFrictionDir() { return V(-1.0f, 0.0f, 0.0f); }

struct SolverRow {
    Vec3 normal{0, 0, 0};         // m_contactNormal1
    Vec3 relposCrossN{0, 0, 0};   // m_relpos1CrossNormal
    Vec3 angularComp{0, 0, 0};    // m_angularComponentA
    float jacDiagABInv = 0.0f, rhs = 0.0f, rhsPenetration = 0.0f, cfm = 0.0f;
    float lower = 0.0f, upper = 0.0f, friction = 0.0f;
    float applied = 0.0f, appliedPush = 0.0f;
};

struct SolverBody {
    Vec3 dLin{0, 0, 0}, dAng{0, 0, 0}, push{0, 0, 0}, turn{0, 0, 0};
    Vec3 linVel{0, 0, 0}, angVel{0, 0, 0}, extForce{0, 0, 0}, extTorque{0, 0, 0};
    Vec3 invMass{0, 0, 0};
};

inline float Fma(float a, float b, float c) { return std::fma(a, b, c); }

// gResolveSingleConstraintRow{LowerLimit,Generic}_sse4_1_fma3 as compiled into the reference's exe (header of this file).
// The plane's side (solver body B: the fixed body, zero everywhere) contributes deltaVel2Dotn = 0 and receives zero updates.
inline void ResolveRow(SolverBody& a, SolverRow& c, bool withUpperLimit)
{
    float deltaImpulse = c.rhs - c.applied * c.cfm;
    const float dv1 = ((c.relposCrossN.x * a.dAng.x + c.relposCrossN.y * a.dAng.y) + c.relposCrossN.z * a.dAng.z) +
                      ((c.normal.x * a.dLin.x + c.normal.y * a.dLin.y) + c.normal.z * a.dLin.z);
    const float dv2 = 0.0f + 0.0f;
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
}

// gResolveSplitPenetrationImpulse_sse2 (exe VA 0x1401c9790: each dot product is (z + y) + x, no FMA anywhere)
inline void ResolveSplitPenetration(SolverBody& a, SolverRow& c)
{
    if (!c.rhsPenetration) return;
    auto dot3 = [](const Vec3& u, const Vec3& v) { return u.x * v.x + (u.y * v.y + u.z * v.z); };
    float deltaImpulse = c.rhsPenetration - c.appliedPush * c.cfm;
    const float dv1 = dot3(c.normal, a.push) + dot3(c.relposCrossN, a.turn);
    const float dv2 = 0.0f + 0.0f;
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
}

struct BodyState {
    Vec3 origin, linVel, angVel;
    Quat orn;   // what physics_ref.h's CurrentOrn() yields for the body
    Mat3 basis; // its matrix
};

// solveGroup for the island {body} with its ground manifold.  `force` = btRigidBody::m_totalForce after applyGravity.
// Returns true when the split impulse moved the body (origin / orientation were changed).
inline bool SolveBodyAgainstGround(BodyState& b, Manifold& m, const Shape& shape, float invMassScalar, const Vec3& invInertiaLocal,
                                   const Vec3& localInertia, float friction, const Vec3& force, float dt)
{
    constexpr int kIterations = 10;               // btContactSolverInfo::m_numIterations
    constexpr float kErp2 = 0.2f;                 // m_erp2
    constexpr float kSplitThreshold = -0.04f;     // m_splitImpulsePenetrationThreshold
    constexpr float kSplitTurnErp = 0.1f;         // m_splitImpulseTurnErp
    constexpr float kWarmstart = 0.85f;           // m_warmstartingFactor
    constexpr float kSor = 1.0f;                  // m_sor
    (void)shape;
    const Vec3 n = V(0.0f, 1.0f, 0.0f);
    const Mat3 invI = InvInertiaWorld(b.basis, invInertiaLocal);

    // convertBodies -> initSolverBody
    SolverBody sb;
    sb.invMass = V(invMassScalar, invMassScalar, invMassScalar); // invMass * linearFactor (1,1,1)
    sb.linVel = b.linVel;
    sb.angVel = b.angVel;
    sb.extForce = Scale(Scale(force, invMassScalar), dt);          // getTotalForce() * getInvMass() * timeStep
    sb.extTorque = Scale(MatVec(invI, V(0.0f, 0.0f, 0.0f)), dt);   // getTotalTorque() * invInertiaTensorWorld * timeStep: zero
    sb.extTorque = V(0.0f, 0.0f, 0.0f);
    sb.extTorque = Add(sb.extTorque, GyroscopicImpulse(localInertia, b.angVel, b.orn, dt));

    // convertContact
    SolverRow normalRow[4], frictionRow[4];
    const float invTimeStep = 1.0f / dt;
    const float combinedFriction = std::max(-10.0f, std::min(10.0f, friction * 1.0f)); // calculateCombinedFriction with the ground's 1.0
    for (int j = 0; j < m.n; ++j) {
        ContactPoint& cp = m.p[j];
        SolverRow& c = normalRow[j];
        c = SolverRow{};
        const Vec3 rel_pos1 = Sub(cp.worldA, b.origin);
        // getVelocityInLocalPointNoDelta
        const Vec3 vel1 = Add(Add(sb.linVel, sb.extForce), Cross(Add(sb.angVel, sb.extTorque), rel_pos1));
        const Vec3 vel = Sub(vel1, V(0.0f, 0.0f, 0.0f));
        const float rel_vel = Dot(n, vel);
        // setupContactConstraint
        const float relaxation = kSor;
        const Vec3 torqueAxis0 = Cross(rel_pos1, n);
        c.angularComp = MatVec(invI, torqueAxis0); // * angularFactor (1,1,1)
        {
            const Vec3 vec = Cross(c.angularComp, rel_pos1);
            const float denom0 = invMassScalar + Dot(n, vec);
            const float cfm0 = 0.0f * invTimeStep;
            c.jacDiagABInv = relaxation / (denom0 + 0.0f + cfm0);
        }
        c.normal = n;
        c.relposCrossN = torqueAxis0;
        const float penetration = cp.distance + 0.0f; // + m_linearSlop
        c.friction = combinedFriction;
        const float restitution = 0.0f; // combined restitution 0: restitutionCurve gives 0 or -0, clamped to 0
        // warm starting
        c.applied = cp.appliedImpulse * kWarmstart;
        {
            const Vec3 lin = V(c.normal.x * sb.invMass.x, c.normal.y * sb.invMass.y, c.normal.z * sb.invMass.z);
            sb.dLin = Add(sb.dLin, Scale(lin, c.applied));               // internalApplyImpulse: += linearComponent * impulse * linearFactor
            sb.dAng = Add(sb.dAng, Scale(c.angularComp, c.applied * 1.0f)); // += angularComponent * (impulse * angularFactor)
        }
        c.appliedPush = 0.0f;
        {
            const float vel1Dotn = Dot(c.normal, Add(sb.linVel, sb.extForce)) + Dot(c.relposCrossN, Add(sb.angVel, sb.extTorque));
            const float vel2Dotn = 0.0f + 0.0f;
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
            if (penetration > kSplitThreshold) { // m_splitImpulse is on
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
        // friction direction: the lateral relative velocity, or btPlaneSpace1's first tangent when it vanishes
        Vec3 dir = Sub(vel, Scale(n, rel_vel));
        const float lat_rel_vel = Dot(dir, dir);
        if (lat_rel_vel > bt::kEpsilon) {
            dir = Scale(dir, 1.0f / std::sqrt(lat_rel_vel));
        } else {
            dir = FallbackFrictionDir();
        }
        // setupFrictionConstraint
        SolverRow& f = frictionRow[j];
        f = SolverRow{};
        f.friction = combinedFriction;
        f.normal = dir;
        f.relposCrossN = Cross(rel_pos1, dir);
        f.angularComp = MatVec(invI, f.relposCrossN);
        {
            const Vec3 vec = Cross(f.angularComp, rel_pos1);
            const float denom0 = invMassScalar + Dot(dir, vec);
            f.jacDiagABInv = relaxation / (denom0 + 0.0f);
        }
        {
            const float vel1Dotn = Dot(f.normal, Add(sb.linVel, sb.extForce)) + Dot(f.relposCrossN, sb.angVel);
            const float vel2Dotn = 0.0f + 0.0f;
            const float rv = vel1Dotn + vel2Dotn;
            const float velocityError = 0.0f - rv;
            const float velocityImpulse = velocityError * f.jacDiagABInv;
            f.rhs = 0.0f + velocityImpulse;
            f.rhsPenetration = 0.0f;
            f.cfm = 0.0f;
            f.lower = -f.friction;
            f.upper = f.friction;
        }
        // setFrictionConstraintImpulse (warm starting)
        f.applied = cp.appliedImpulseLateral1 * kWarmstart;
        {
            const Vec3 lin = Scale(f.normal, invMassScalar); // m_contactNormal1 * rb0->getInvMass()
            sb.dLin = Add(sb.dLin, Scale(lin, f.applied));
            sb.dAng = Add(sb.dAng, Scale(f.angularComp, f.applied * 1.0f));
        }
    }

    // solveGroupCacheFriendlySplitImpulseIterations
    for (int it = 0; it < kIterations; ++it) {
        for (int j = 0; j < m.n; ++j) ResolveSplitPenetration(sb, normalRow[j]);
    }
    // velocity iterations: all contact rows, then all friction rows (no interleaving by default)
    for (int it = 0; it < kIterations; ++it) {
        for (int j = 0; j < m.n; ++j) ResolveRow(sb, normalRow[j], false);
        for (int j = 0; j < m.n; ++j) {
            const float totalImpulse = normalRow[j].applied;
            if (totalImpulse > 0.0f) {
                frictionRow[j].lower = -(frictionRow[j].friction * totalImpulse);
                frictionRow[j].upper = frictionRow[j].friction * totalImpulse;
                ResolveRow(sb, frictionRow[j], true);
            }
        }
    }
    // solveGroupCacheFriendlyFinish: impulses back into the manifold, velocities (and the pushed transform) into the body
    for (int j = 0; j < m.n; ++j) {
        m.p[j].appliedImpulse = normalRow[j].applied;
        m.p[j].appliedImpulseLateral1 = frictionRow[j].applied;
    }
    sb.linVel = Add(sb.linVel, sb.dLin); // writebackVelocityAndTransform
    sb.angVel = Add(sb.angVel, sb.dAng);
    bool moved = false;
    if (sb.push.x != 0.0f || sb.push.y != 0.0f || sb.push.z != 0.0f || sb.turn.x != 0.0f || sb.turn.y != 0.0f || sb.turn.z != 0.0f) {
        // btTransformUtil::integrateTransform(worldTransform, pushVelocity, turnVelocity * splitImpulseTurnErp, timeStep)
        b.origin = Add(b.origin, Scale(sb.push, dt));
        b.orn = bt::IntegrateOrientation(b.orn, Scale(sb.turn, kSplitTurnErp), dt);
        b.basis = bt::MatFromQuat(b.orn);
        moved = true;
    }
    b.linVel = Add(sb.linVel, sb.extForce);
    b.angVel = Add(sb.angVel, sb.extTorque);
    return moved;
}

} // namespace ct
} // namespace orc
