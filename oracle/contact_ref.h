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
//   * the rest of the path, executed symbolically by oracle/tools/check_solver_setup.py (expression trees of the COMPILED code
//     against this file's, as for integrateTransform): setupContactConstraint (VA 0x1401cbb10), setupFrictionConstraint
//     (0x1401cca80), convertContact's relative velocity and lateral direction (0x1401c58c0), computeGyroscopicImpulseImplicit_Body
//     (0x1401ae380) with btMatrix3x3::solve33, btPersistentManifold::refreshContactPoints (0x1401fed20) / sortCachedPoints
//     (0x1401ff840) / getCacheEntry (0x1401fe930), btManifoldResult::addContactPoint's local points (0x140204500),
//     btConvexPlaneCollisionAlgorithm::processCollision's single contact (0x14021ba50), the capsule's support vertex,
//     btRigidBody::updateInertiaTensor (0x1401b2b50) and the solver's write-back.  Bullet is built with MSVC /fp:fast in that
//     exe (its CMake default), so sums are NOT always left to right; what the restatement took over from the compiled code:
//       - denom0 of a row = (invMass + n.z vec.z) + (n.x vec.x + n.y vec.y)                       (InvMassPlusDot)
//       - the velocity dot products of a row's right-hand side are summed (x + z) + y              (DotXZY)
//       - refreshContactPoints: world point = (origin + l.y B[r][1]) + (l.x B[r][0] + l.z B[r][2]) (XformPoint; body B's x row
//         pairs the origin with the z product: XformPointB)
//       - friction rows are NOT warm-started: setFrictionConstraintImpulse of this Bullet zeroes m_appliedImpulse
//       - getLocalInertia() is 1 / m_invInertiaLocal per component, not the inertia the shape computed
//     and the sites that were already the compiled form: rel_pos, getVelocityInLocalPointNoDelta, rel_vel (x + y) + z, the
//     lateral direction and its normalisation by one reciprocal, btPlaneSpace1, restitutionCurve and its dead band, the
//     positional error -(pen erp) / dt, solve33, quatRotate, sortCachedPoints' areas, invXform, the support functions.
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

// ---- associations of the reference's compiled code (MSVC /fp:fast; oracle/tools/check_solver_setup.py)
// a row's velocity dot products (setupContactConstraint / setupFrictionConstraint): (x + z) + y
inline float DotXZY(const Vec3& a, const Vec3& b) { return (a.x * b.x + a.z * b.z) + a.y * b.y; }
// denom0 = rb0->getInvMass() + n.dot(vec) as compiled
inline float InvMassPlusDot(float invMass, const Vec3& n, const Vec3& vec) { return (invMass + n.z * vec.z) + (n.x * vec.x + n.y * vec.y); }
// refreshContactPoints: trA(localPointA), every row (origin + l.y B[r][1]) + (l.x B[r][0] + l.z B[r][2])
inline Vec3 XformPoint(const Mat3& b, const Vec3& o, const Vec3& l)
{
    return V((o.x + l.y * b.m[0][1]) + (l.x * b.m[0][0] + l.z * b.m[0][2]), (o.y + l.y * b.m[1][1]) + (l.x * b.m[1][0] + l.z * b.m[1][2]),
             (o.z + l.y * b.m[2][1]) + (l.x * b.m[2][0] + l.z * b.m[2][2]));
}
// refreshContactPoints: trB(localPointB): the x row pairs the origin with the z product
inline Vec3 XformPointB(const Mat3& b, const Vec3& o, const Vec3& l)
{
    return V((o.x + l.z * b.m[0][2]) + (l.x * b.m[0][0] + l.y * b.m[0][1]), (o.y + l.y * b.m[1][1]) + (l.x * b.m[1][0] + l.z * b.m[1][2]),
             (o.z + l.y * b.m[2][1]) + (l.x * b.m[2][0] + l.z * b.m[2][2]));
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
    // (mass / 12 in the source; the reference's compiled code — MSVC /fp:fast — multiplies by the constant 0x3daaaaab instead)
    const float m12 = mass * 0.0833333358168602f;
    return V(m12 * (ly * ly + lz * lz), m12 * (lx * lx + lz * lz), m12 * (lx * lx + ly * ly));
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
        c.worldA = XformPoint(basis, origin, c.localA);
        c.worldB = c.localB; // (the plane's transform is the identity: XformPointB gives the point itself)
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
// frame (BT_ENABLE_GYROSCOPIC_FORCE_IMPLICIT_BODY is a btRigidBody default flag).  idl = getLocalInertia() = 1 / m_invInertiaLocal
// per component (0 where that is 0).  J = Ib + (skew0 * Ib - skew1) * step as compiled: the products with the zeros of the
// diagonal Ib and of the skew matrices are folded away (they are exact), which leaves the entries below.
inline Vec3 GyroscopicImpulse(const Vec3& invInertiaLocal, const Vec3& omega1, const Quat& q, float step)
{
    const Vec3 idl = V(invInertiaLocal.x != 0.0f ? 1.0f / invInertiaLocal.x : 0.0f, invInertiaLocal.y != 0.0f ? 1.0f / invInertiaLocal.y : 0.0f,
                       invInertiaLocal.z != 0.0f ? 1.0f / invInertiaLocal.z : 0.0f);
    const Quat qinv{-q.x, -q.y, -q.z, q.w};
    Vec3 omegab = QuatRotate(qinv, omega1);
    const Vec3 ibo = V(idl.x * omegab.x, idl.y * omegab.y, idl.z * omegab.z); // Ib * omegab, Ib diagonal
    const Vec3 f = Scale(Cross(omegab, ibo), step);
    Mat3 J;
    J.m[0][0] = idl.x;
    J.m[0][1] = (idl.z * omegab.z - idl.y * omegab.z) * step;
    J.m[0][2] = (idl.z * omegab.y - idl.y * omegab.y) * step;
    J.m[1][0] = (idl.x * omegab.z - idl.z * omegab.z) * step;
    J.m[1][1] = idl.y;
    J.m[1][2] = (idl.x * omegab.x - idl.z * omegab.x) * step;
    J.m[2][0] = (idl.y * omegab.y - idl.x * omegab.y) * step;
    J.m[2][1] = (idl.y * omegab.x - idl.x * omegab.x) * step;
    J.m[2][2] = idl.z;
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

} // namespace ct
} // namespace orc
