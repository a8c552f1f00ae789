// bge_contact.hip — the static ground plane of the reference's physics world and Bullet's contact handling for it, one
// thread per body.
//
// The reference adds an infinite static plane at y = 0 (friction 1, restitution 0) to every world
// (src/physics/PhysicsSystem.cpp:149-166) and steps with Bullet's default narrowphase and sequential-impulse solver
// (:122-128, :863).  Bodies do not collide with each other on this path: every Dynamic body is an island of its own whose
// only manifold is the one with the plane, so the step is independent per body and maps onto one thread:
//   k_ground   performDiscreteCollisionDetection for (plane, body) — btConvexPlaneCollisionAlgorithm: one contact per step at
//              the shape's support vertex, kept in a 4-point persistent manifold (nearest-point replacement, area-based
//              eviction, refresh with removal beyond the breaking threshold) — and solveConstraints for the island {body}:
//              external force impulse, implicit gyroscopic impulse, contact and friction rows with warm starting,
//              10 split-impulse iterations, 10 velocity iterations, write-back of velocities, impulses and the pushed pose.
//              A body without a contact and without angular velocity is left to k_tick's plain update (the same arithmetic).
//   k_tick     then integrates the pose with the velocities found here (cinfo bit kCiSolved: gravity is already in them).
// The arithmetic — operation order, where Bullet's compiled row solvers fuse multiply-adds, which dot products add in
// which order — is oracle/contact_ref.h's (that header states what is pinned by the reference's exe and what is not);
// tests/test_gpu_parity.py compares every body bit for bit.  Heavy per-thread state (four contact rows and four friction
// rows) makes this a register-hungry kernel; it touches only bodies at the ground, so it is sized for correctness first.
// Further down: Dynamic boxes on the Static / Kinematic box colliders of the scene (k_obstacles, k_contact_boxes: one island per body, the
// rows in scratch memory) and Dynamic boxes against EACH OTHER ("islands": the pair cache with a manifold per pair, union-find over the
// pairs, a solver thread — or, for a big island, a workgroup walking Bullet's row order level by level — per island).
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include <algorithm>

#include "bge_boxbox_device.hpp"
#include "bge_device_math.hpp"
#include "bge_flatten.hpp"
#include "bge_kernels.hpp"

namespace bge {

using namespace dev;

namespace {

__device__ __forceinline__ F3 add3(const F3& a, const F3& b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 sub3(const F3& a, const F3& b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 scale3(const F3& a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(const F3& a, const F3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross3(const F3& a, const F3& b) { return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ F3 mat_vec(const M3& m, const F3& v)
{
    return F3{m.m[0][0] * v.x + m.m[0][1] * v.y + m.m[0][2] * v.z, m.m[1][0] * v.x + m.m[1][1] * v.y + m.m[1][2] * v.z,
              m.m[2][0] * v.x + m.m[2][1] * v.y + m.m[2][2] * v.z};
}
__device__ __forceinline__ F3 mat_t_vec(const M3& m, const F3& v)
{
    return F3{m.m[0][0] * v.x + m.m[1][0] * v.y + m.m[2][0] * v.z, m.m[0][1] * v.x + m.m[1][1] * v.y + m.m[2][1] * v.z,
              m.m[0][2] * v.x + m.m[1][2] * v.y + m.m[2][2] * v.z};
}

// ---- associations of the reference's compiled code (MSVC /fp:fast; oracle/contact_ref.h, oracle/tools/check_solver_setup.py)
__device__ __forceinline__ float dot_xzy(const F3& a, const F3& b) { return (a.x * b.x + a.z * b.z) + a.y * b.y; }
__device__ __forceinline__ float inv_mass_plus_dot(float invMass, const F3& n, const F3& vec) { return (invMass + n.z * vec.z) + (n.x * vec.x + n.y * vec.y); }
__device__ __forceinline__ F3 xform_point(const M3& b, const F3& o, const F3& l)
{
    return F3{(o.x + l.y * b.m[0][1]) + (l.x * b.m[0][0] + l.z * b.m[0][2]), (o.y + l.y * b.m[1][1]) + (l.x * b.m[1][0] + l.z * b.m[1][2]),
              (o.z + l.y * b.m[2][1]) + (l.x * b.m[2][0] + l.z * b.m[2][2])};
}
__device__ __forceinline__ F3 xform_point_b(const M3& b, const F3& o, const F3& l)
{
    return F3{(o.x + l.z * b.m[0][2]) + (l.x * b.m[0][0] + l.y * b.m[0][1]), (o.y + l.y * b.m[1][1]) + (l.x * b.m[1][0] + l.z * b.m[1][2]),
              (o.z + l.y * b.m[2][1]) + (l.x * b.m[2][0] + l.z * b.m[2][2])};
}

struct CtShape {
    bool capsule;
    F3 dims; // box: half extents with margin; capsule: (radius, half height, radius)
};

__device__ __forceinline__ F3 ct_local_inertia(const CtShape& s, float mass)
{
    if (s.capsule) {
        const float radius = s.dims.x;
        const float hx = radius, hy = radius + s.dims.y, hz = radius;
        const float lx = 2.0f * hx, ly = 2.0f * hy, lz = 2.0f * hz;
        const float x2 = lx * lx, y2 = ly * ly, z2 = lz * lz;
        const float scaledmass = mass * 0.08333333f;
        return F3{scaledmass * (y2 + z2), scaledmass * (x2 + z2), scaledmass * (x2 + y2)};
    }
    const float lx = 2.0f * s.dims.x, ly = 2.0f * s.dims.y, lz = 2.0f * s.dims.z;
    const float m12 = mass * 0.0833333358168602f; // (mass / 12 as the reference's compiled code has it: times 0x3daaaaab)
    return F3{m12 * (ly * ly + lz * lz), m12 * (lx * lx + lz * lz), m12 * (lx * lx + ly * ly)};
}
__device__ __forceinline__ F3 ct_inv_inertia_local(const F3& i)
{
    return F3{i.x != 0.0f ? 1.0f / i.x : 0.0f, i.y != 0.0f ? 1.0f / i.y : 0.0f, i.z != 0.0f ? 1.0f / i.z : 0.0f};
}
__device__ __forceinline__ M3 ct_inv_inertia_world(const M3& b, const F3& il)
{
    M3 s;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        s.m[r][0] = b.m[r][0] * il.x;
        s.m[r][1] = b.m[r][1] * il.y;
        s.m[r][2] = b.m[r][2] * il.z;
    }
    M3 o;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o.m[r][c] = s.m[r][0] * b.m[c][0] + s.m[r][1] * b.m[c][1] + s.m[r][2] * b.m[c][2];
    }
    return o;
}
__device__ __forceinline__ float ct_breaking_threshold(const CtShape& s)
{
    const float ex = s.dims.x;
    const float ey = s.capsule ? s.dims.x + s.dims.y : s.dims.y;
    const float ez = s.capsule ? s.dims.x : s.dims.z;
    const F3 mn = F3{0.0f - ex, 0.0f - ey, 0.0f - ez}, mx = F3{0.0f + ex, 0.0f + ey, 0.0f + ez};
    const F3 d = sub3(mx, mn);
    const float radius = __builtin_sqrtf(dot3(d, d)) * 0.5f;
    const F3 c = scale3(add3(mn, mx), 0.5f);
    const float disc = radius + __builtin_sqrtf(dot3(c, c));
    return disc * kBtContactBreakingThreshold;
}
__device__ __forceinline__ F3 ct_support_vertex(const CtShape& s, const F3& dir)
{
    if (!s.capsule) return F3{dir.x >= 0.0f ? s.dims.x : -s.dims.x, dir.y >= 0.0f ? s.dims.y : -s.dims.y, dir.z >= 0.0f ? s.dims.z : -s.dims.z};
    F3 vec = dir;
    const float lenSqr = dot3(vec, vec);
    if (lenSqr < 0.0001f) {
        vec = F3{1.0f, 0.0f, 0.0f};
    } else {
        const float rlen = 1.0f / __builtin_sqrtf(lenSqr);
        vec = scale3(vec, rlen);
    }
    F3 sup = F3{0.0f, 0.0f, 0.0f};
    float maxDot = -1.0e18f;
    {
        const F3 vtx = F3{0.0f, s.dims.y, 0.0f};
        const float d = dot3(vec, vtx);
        if (d > maxDot) {
            maxDot = d;
            sup = vtx;
        }
    }
    {
        const F3 vtx = F3{0.0f, -s.dims.y, 0.0f};
        const float d = dot3(vec, vtx);
        if (d > maxDot) {
            maxDot = d;
            sup = vtx;
        }
    }
    F3 vecnorm = dir;
    if (dot3(vecnorm, vecnorm) < kBtEpsilon * kBtEpsilon) vecnorm = F3{-1.0f, -1.0f, -1.0f};
    vecnorm = scale3(vecnorm, 1.0f / __builtin_sqrtf(dot3(vecnorm, vecnorm)));
    return add3(sup, scale3(vecnorm, s.dims.x));
}

struct CtPoint {
    F3 localA, localB;
    float appliedImpulse, appliedLateral;
    F3 worldA, worldB;
    float distance;
};
__device__ __forceinline__ CtPoint ct_empty_point()
{
    CtPoint p;
    p.localA = p.localB = p.worldA = p.worldB = F3{0.0f, 0.0f, 0.0f};
    p.appliedImpulse = p.appliedLateral = p.distance = 0.0f;
    return p;
}

// d = c ? s : d, field by field.  Written as `if (i == k) p[i] = s;` over the four points, the compiler turns the chain into a
// switch and sinks the stores behind a phi of POINTERS to the selected point's fields — which keeps all four points in scratch
// memory for the whole kernel (320 B per lane, every access a memory round trip).
__device__ __forceinline__ void ct_point_select(CtPoint& d, bool c, const CtPoint& s)
{
    d.localA = F3{c ? s.localA.x : d.localA.x, c ? s.localA.y : d.localA.y, c ? s.localA.z : d.localA.z};
    d.localB = F3{c ? s.localB.x : d.localB.x, c ? s.localB.y : d.localB.y, c ? s.localB.z : d.localB.z};
    d.worldA = F3{c ? s.worldA.x : d.worldA.x, c ? s.worldA.y : d.worldA.y, c ? s.worldA.z : d.worldA.z};
    d.worldB = F3{c ? s.worldB.x : d.worldB.x, c ? s.worldB.y : d.worldB.y, c ? s.worldB.z : d.worldB.z};
    d.appliedImpulse = c ? s.appliedImpulse : d.appliedImpulse;
    d.appliedLateral = c ? s.appliedLateral : d.appliedLateral;
    d.distance = c ? s.distance : d.distance;
}

__device__ __forceinline__ int ct_sort_cached_points(const CtPoint (&p)[4], const CtPoint& pt)
{
    int maxPenetrationIndex = -1;
    float maxPenetration = pt.distance;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (p[i].distance < maxPenetration) {
            maxPenetrationIndex = i;
            maxPenetration = p[i].distance;
        }
    }
    float res[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const F3 p0 = p[0].localA, p1 = p[1].localA, p2 = p[2].localA, p3 = p[3].localA;
    if (maxPenetrationIndex != 0) {
        const F3 c = cross3(sub3(pt.localA, p1), sub3(p3, p2));
        res[0] = dot3(c, c);
    }
    if (maxPenetrationIndex != 1) {
        const F3 c = cross3(sub3(pt.localA, p0), sub3(p3, p2));
        res[1] = dot3(c, c);
    }
    if (maxPenetrationIndex != 2) {
        const F3 c = cross3(sub3(pt.localA, p0), sub3(p3, p1));
        res[2] = dot3(c, c);
    }
    if (maxPenetrationIndex != 3) {
        const F3 c = cross3(sub3(pt.localA, p0), sub3(p2, p1));
        res[3] = dot3(c, c);
    }
    int maxIndex = -1;
    float maxVal = -1.0e18f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = __builtin_fabsf(res[i]);
        if (a > maxVal) {
            maxIndex = i;
            maxVal = a;
        }
    }
    return maxIndex;
}

// btConvexPlaneCollisionAlgorithm::processCollision against y = 0 (oracle/contact_ref.h CollideWithGround)
__device__ __forceinline__ void ct_collide(CtPoint (&p)[4], int& n, const CtShape& shape, float breaking, const F3& origin, const M3& basis)
{
    const F3 dirLocal = F3{-basis.m[1][0], -basis.m[1][1], -basis.m[1][2]};
    const F3 vtx = ct_support_vertex(shape, dirLocal);
    const F3 vtxInPlane = add3(mat_vec(basis, vtx), origin);
    const float distance = vtxInPlane.y;
    if (distance < breaking) {
        const F3 pointInWorld = F3{vtxInPlane.x, vtxInPlane.y - distance, vtxInPlane.z};
        if (!(distance > breaking)) {
            CtPoint np = ct_empty_point();
            const F3 pointA = F3{pointInWorld.x, pointInWorld.y + distance, pointInWorld.z};
            np.localA = mat_t_vec(basis, sub3(pointA, origin));
            np.localB = pointInWorld;
            np.worldA = pointA;
            np.worldB = pointInWorld;
            np.distance = distance;
            float shortest = breaking * breaking;
            int nearest = -1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < n) {
                    const F3 diffA = sub3(p[i].localA, np.localA);
                    const float d2 = dot3(diffA, diffA);
                    if (d2 < shortest) {
                        shortest = d2;
                        nearest = i;
                    }
                }
            }
            int insert;
            if (nearest >= 0) {
                insert = nearest;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i == nearest) {
                        np.appliedImpulse = p[i].appliedImpulse;
                        np.appliedLateral = p[i].appliedLateral;
                    }
                }
            } else {
                insert = n;
                if (insert == 4) {
                    insert = ct_sort_cached_points(p, np);
                } else {
                    n++;
                }
                if (insert < 0) insert = 0;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) ct_point_select(p[i], i == insert, np);
        }
    }
    // refreshContactPoints
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        if (i < n) {
            p[i].worldA = xform_point(basis, origin, p[i].localA);
            p[i].worldB = p[i].localB;
            p[i].distance = dot3(sub3(p[i].worldA, p[i].worldB), F3{0.0f, 1.0f, 0.0f});
        }
    }
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        if (i < n) {
            bool remove = !(p[i].distance <= breaking);
            if (!remove) {
                const F3 projectedPoint = sub3(p[i].worldA, scale3(F3{0.0f, 1.0f, 0.0f}, p[i].distance));
                const F3 projectedDifference = sub3(p[i].worldB, projectedPoint);
                const float distance2d = dot3(projectedDifference, projectedDifference);
                remove = distance2d > breaking * breaking;
            }
            {
                // removeContactPoint: the last point takes the removed one's place (selects, not branches: see ct_point_select)
                const int last = n - 1;
                CtPoint moved = ct_empty_point();
#pragma unroll
                for (int k = 0; k < 4; ++k) ct_point_select(moved, k == last, p[k]);
                ct_point_select(p[i], remove && i != last, moved);
                const CtPoint empty = ct_empty_point();
#pragma unroll
                for (int k = 0; k < 4; ++k) ct_point_select(p[k], remove && k == last, empty);
                if (remove) n--;
            }
        }
    }
}

__device__ __forceinline__ Q4 ct_quat_times_vec(const Q4& q, const F3& w)
{
    return Q4{q.w * w.x + q.y * w.z - q.z * w.y, q.w * w.y + q.z * w.x - q.x * w.z, q.w * w.z + q.x * w.y - q.y * w.x,
              -q.x * w.x - q.y * w.y - q.z * w.z};
}
__device__ __forceinline__ Q4 ct_quat_mul(const Q4& a, const Q4& b)
{
    Q4 r;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    return r;
}
__device__ __forceinline__ F3 ct_quat_rotate(const Q4& rotation, const F3& v)
{
    const Q4 q = ct_quat_times_vec(rotation, v);
    const Q4 inv{-rotation.x, -rotation.y, -rotation.z, rotation.w};
    const Q4 r = ct_quat_mul(q, inv);
    return F3{r.x, r.y, r.z};
}
__device__ __forceinline__ F3 ct_solve33(const M3& J, const F3& b)
{
    const F3 col1 = F3{J.m[0][0], J.m[1][0], J.m[2][0]};
    const F3 col2 = F3{J.m[0][1], J.m[1][1], J.m[2][1]};
    const F3 col3 = F3{J.m[0][2], J.m[1][2], J.m[2][2]};
    float det = dot3(col1, cross3(col2, col3));
    if (__builtin_fabsf(det) > kBtEpsilon) det = 1.0f / det;
    return F3{det * dot3(b, cross3(col2, col3)), det * dot3(col1, cross3(b, col3)), det * dot3(col1, cross3(col2, b))};
}
// computeGyroscopicImpulseImplicit_Body: idl = getLocalInertia() = 1 / m_invInertiaLocal; J with the exact zero products folded
// away, as compiled (oracle/contact_ref.h GyroscopicImpulse)
__device__ __forceinline__ F3 ct_gyroscopic_impulse(const F3& invInertiaLocal, const F3& omega1, const Q4& q, float step)
{
    const F3 idl = F3{invInertiaLocal.x != 0.0f ? 1.0f / invInertiaLocal.x : 0.0f, invInertiaLocal.y != 0.0f ? 1.0f / invInertiaLocal.y : 0.0f,
                      invInertiaLocal.z != 0.0f ? 1.0f / invInertiaLocal.z : 0.0f};
    const Q4 qinv{-q.x, -q.y, -q.z, q.w};
    F3 omegab = ct_quat_rotate(qinv, omega1);
    const F3 ibo = F3{idl.x * omegab.x, idl.y * omegab.y, idl.z * omegab.z};
    const F3 f = scale3(cross3(omegab, ibo), step);
    M3 J;
    J.m[0][0] = idl.x;
    J.m[0][1] = (idl.z * omegab.z - idl.y * omegab.z) * step;
    J.m[0][2] = (idl.z * omegab.y - idl.y * omegab.y) * step;
    J.m[1][0] = (idl.x * omegab.z - idl.z * omegab.z) * step;
    J.m[1][1] = idl.y;
    J.m[1][2] = (idl.x * omegab.x - idl.z * omegab.x) * step;
    J.m[2][0] = (idl.y * omegab.y - idl.x * omegab.y) * step;
    J.m[2][1] = (idl.y * omegab.x - idl.x * omegab.x) * step;
    J.m[2][2] = idl.z;
    const F3 omega_div = ct_solve33(J, f);
    omegab = sub3(omegab, omega_div);
    const F3 omega2 = ct_quat_rotate(q, omegab);
    return sub3(omega2, omega1);
}

struct CtRow {
    F3 normal, relposCrossN, angularComp;
    float jacDiagABInv, rhs, rhsPenetration, cfm, lower, upper, friction, applied, appliedPush;
};
struct CtBody {
    F3 dLin, dAng, push, turn, linVel, angVel, extForce, extTorque, invMass;
};

// PLANE: the row's normal is the constant (0, 1, 0) of the ground plane and the body's inverse mass is finite.  Then
//   (0 * dLin.x + 1 * dLin.y) + 0 * dLin.z  ==  dLin.y   and   fma(0 * invMass, deltaImpulse, dLin.x)  ==  dLin.x   (z alike)
// bit for bit, PROVIDED no component of dLin is -0 (and none is inf / NaN) — and none ever is: dLin starts at +0, every update is
// a sum or an fma whose addend is dLin itself, and in round-to-nearest such a result is -0 only when the addend already was.
// With that the dot product's four operations and the two dead updates are left out: ten of a row's 28 instructions.
template <bool PLANE = false>
__device__ __forceinline__ void ct_resolve_row(CtBody& a, CtRow& c, bool withUpperLimit)
{
    float deltaImpulse = c.rhs - c.applied * c.cfm;
    const float lin = PLANE ? a.dLin.y : ((c.normal.x * a.dLin.x + c.normal.y * a.dLin.y) + c.normal.z * a.dLin.z);
    const float dv1 = ((c.relposCrossN.x * a.dAng.x + c.relposCrossN.y * a.dAng.y) + c.relposCrossN.z * a.dAng.z) + lin;
    deltaImpulse = __builtin_fmaf(-dv1, c.jacDiagABInv, deltaImpulse);
    // (the other body's fnmadd, fma(-(0 + 0), jacDiagABInv, deltaImpulse), adds -0 — jacDiagABInv is positive — and changes nothing)
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
    if (PLANE) {
        a.dLin.y = __builtin_fmaf(a.invMass.y, deltaImpulse, a.dLin.y); // (1 * invMass is invMass)
    } else {
        a.dLin = F3{__builtin_fmaf(c.normal.x * a.invMass.x, deltaImpulse, a.dLin.x), __builtin_fmaf(c.normal.y * a.invMass.y, deltaImpulse, a.dLin.y),
                    __builtin_fmaf(c.normal.z * a.invMass.z, deltaImpulse, a.dLin.z)};
    }
    a.dAng = F3{__builtin_fmaf(c.angularComp.x, deltaImpulse, a.dAng.x), __builtin_fmaf(c.angularComp.y, deltaImpulse, a.dAng.y),
                __builtin_fmaf(c.angularComp.z, deltaImpulse, a.dAng.z)};
}

__device__ __forceinline__ void ct_resolve_split(CtBody& a, CtRow& c)
{
    if (!c.rhsPenetration) return;
    float deltaImpulse = c.rhsPenetration - c.appliedPush * c.cfm;
    const float dv1 = (c.normal.x * a.push.x + (c.normal.y * a.push.y + c.normal.z * a.push.z)) +
                      (c.relposCrossN.x * a.turn.x + (c.relposCrossN.y * a.turn.y + c.relposCrossN.z * a.turn.z));
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
    const F3 lin = F3{c.normal.x * a.invMass.x, c.normal.y * a.invMass.y, c.normal.z * a.invMass.z};
    a.push = add3(a.push, scale3(lin, deltaImpulse));
    a.turn = add3(a.turn, scale3(c.angularComp, deltaImpulse));
}

__device__ __forceinline__ CtRow ct_zero_row()
{
    CtRow c;
    c.normal = c.relposCrossN = c.angularComp = F3{0.0f, 0.0f, 0.0f};
    c.jacDiagABInv = c.rhs = c.rhsPenetration = c.cfm = c.lower = c.upper = c.friction = c.applied = c.appliedPush = 0.0f;
    return c;
}

// solveGroup for the island {body} against the plane alone: oracle/boxbox_ref.h SolveBody with no box manifold
// (inlined into its one caller: as a call its reference arguments — pose, velocities, the four points — lived in scratch memory.
//  1 M resting bodies: 0.426 -> 0.355 ms per tick; with ct_point_select 0.234 and no scratch at all)
#ifndef BGE_CT_SOLVE_INLINE
#define BGE_CT_SOLVE_INLINE __forceinline__
#endif
__device__ BGE_CT_SOLVE_INLINE bool ct_solve(F3& origin, F3& linVel, F3& angVel, Q4& orn, M3& basis, CtPoint (&p)[4], int n, float invMassScalar,
                                      const F3& invInertiaLocal, float friction, const F3& force, float dt)
{
    constexpr int kIterations = 10;
    constexpr float kErp2 = 0.2f, kSplitThreshold = -0.04f, kSplitTurnErp = 0.1f, kWarmstart = 0.85f, kSor = 1.0f;
    const F3 nrm = F3{0.0f, 1.0f, 0.0f};
    const M3 invI = ct_inv_inertia_world(basis, invInertiaLocal);
    CtBody sb;
    sb.dLin = sb.dAng = sb.push = sb.turn = F3{0.0f, 0.0f, 0.0f};
    sb.invMass = F3{invMassScalar, invMassScalar, invMassScalar};
    sb.linVel = linVel;
    sb.angVel = angVel;
    sb.extForce = scale3(scale3(force, invMassScalar), dt);
    sb.extTorque = F3{0.0f, 0.0f, 0.0f};
    sb.extTorque = add3(sb.extTorque, ct_gyroscopic_impulse(invInertiaLocal, angVel, orn, dt));

    CtRow normalRow[4], frictionRow[4];
    const float invTimeStep = 1.0f / dt;
    const float combinedFriction = fmaxf(-10.0f, fminf(10.0f, friction * 1.0f));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        normalRow[j] = ct_zero_row();
        frictionRow[j] = ct_zero_row();
        // what is the same for every row is set whether or not the row exists (rows j >= n are never looked at): set under
        // `j < n` these fields were select(j < n, constant, 0) — eight live registers more per contact in the solver's loops
        normalRow[j].normal = nrm;
        normalRow[j].cfm = 0.0f;
        normalRow[j].lower = 0.0f;
        normalRow[j].upper = 1e10f;
        normalRow[j].friction = combinedFriction;
        frictionRow[j].friction = combinedFriction;
        frictionRow[j].rhsPenetration = 0.0f;
        frictionRow[j].cfm = 0.0f;
        if (j < n) {
            CtRow& c = normalRow[j];
            const F3 rel_pos1 = sub3(p[j].worldA, origin);
            const F3 vel1 = add3(add3(sb.linVel, sb.extForce), cross3(add3(sb.angVel, sb.extTorque), rel_pos1));
            const F3 vel = sub3(vel1, F3{0.0f, 0.0f, 0.0f});
            const float rel_vel = dot3(nrm, vel);
            const float relaxation = kSor;
            const F3 torqueAxis0 = cross3(rel_pos1, nrm);
            c.angularComp = mat_vec(invI, torqueAxis0);
            {
                const F3 vec = cross3(c.angularComp, rel_pos1);
                const float denom0 = inv_mass_plus_dot(invMassScalar, nrm, vec);
                const float cfm0 = 0.0f * invTimeStep;
                c.jacDiagABInv = relaxation / (denom0 + 0.0f + cfm0);
            }
            c.normal = nrm;
            c.relposCrossN = torqueAxis0;
            const float penetration = p[j].distance + 0.0f;
            c.friction = combinedFriction;
            const float restitution = 0.0f;
            c.applied = p[j].appliedImpulse * kWarmstart;
            {
                const F3 lin = F3{c.normal.x * sb.invMass.x, c.normal.y * sb.invMass.y, c.normal.z * sb.invMass.z};
                sb.dLin = add3(sb.dLin, scale3(lin, c.applied));
                sb.dAng = add3(sb.dAng, scale3(c.angularComp, c.applied * 1.0f));
            }
            c.appliedPush = 0.0f;
            {
                const float vel1Dotn = dot_xzy(c.normal, add3(sb.linVel, sb.extForce)) + dot_xzy(c.relposCrossN, add3(sb.angVel, sb.extTorque));
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
            F3 dir = sub3(vel, scale3(nrm, rel_vel));
            const float lat_rel_vel = dot3(dir, dir);
            if (lat_rel_vel > kBtEpsilon) {
                dir = scale3(dir, 1.0f / __builtin_sqrtf(lat_rel_vel));
            } else {
                dir = F3{-1.0f, 0.0f, 0.0f}; // btPlaneSpace1((0,1,0)), first tangent
            }
            CtRow& f = frictionRow[j];
            f.friction = combinedFriction;
            f.normal = dir;
            f.relposCrossN = cross3(rel_pos1, dir);
            f.angularComp = mat_vec(invI, f.relposCrossN);
            {
                const F3 vec = cross3(f.angularComp, rel_pos1);
                const float denom0 = inv_mass_plus_dot(invMassScalar, dir, vec);
                f.jacDiagABInv = relaxation / (denom0 + 0.0f);
            }
            {
                const float vel1Dotn = dot_xzy(f.normal, add3(sb.linVel, sb.extForce)) + dot_xzy(f.relposCrossN, sb.angVel);
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
            f.applied = 0.0f; // setFrictionConstraintImpulse of the reference's Bullet zeroes it: friction rows are not warm-started
        }
    }
    // solveGroupCacheFriendlySplitImpulseIterations.  A row without a penetration impulse returns at once (ct_resolve_split), so a
    // WAVE none of whose bodies has one skips the ten iterations: a resting body's penetration stays above the -0.04 threshold,
    // and its 1,140 predicated instructions were a quarter of the kernel
    bool any_split = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) any_split = any_split || (j < n && normalRow[j].rhsPenetration != 0.0f);
    if (__builtin_amdgcn_ballot_w64(any_split) != 0ull) {
#pragma unroll 1
        for (int it = 0; it < kIterations; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < n) ct_resolve_split(sb, normalRow[j]);
            }
        }
    }
#pragma unroll 1
    for (int it = 0; it < kIterations; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < n) ct_resolve_row<true>(sb, normalRow[j], false);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < n) {
                const float totalImpulse = normalRow[j].applied;
                if (totalImpulse > 0.0f) {
                    frictionRow[j].lower = -(frictionRow[j].friction * totalImpulse);
                    frictionRow[j].upper = frictionRow[j].friction * totalImpulse;
                    ct_resolve_row(sb, frictionRow[j], true);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (j < n) {
            p[j].appliedImpulse = normalRow[j].applied;
            p[j].appliedLateral = frictionRow[j].applied;
        }
    }
    sb.linVel = add3(sb.linVel, sb.dLin);
    sb.angVel = add3(sb.angVel, sb.dAng);
    bool moved = false;
    if (sb.push.x != 0.0f || sb.push.y != 0.0f || sb.push.z != 0.0f || sb.turn.x != 0.0f || sb.turn.y != 0.0f || sb.turn.z != 0.0f) {
        origin = add3(origin, scale3(sb.push, dt));
        orn = bt_integrate_orientation(orn, scale3(sb.turn, kSplitTurnErp), dt);
        basis = bt_mat_from_quat(orn);
        moved = true;
    }
    linVel = add3(sb.linVel, sb.extForce);
    angVel = add3(sb.angVel, sb.extTorque);
    return moved;
}

// One body against the plane: collide, refresh the cached manifold, solve.  Every test of k_ground_select is repeated here (they
// are cheap beside what follows), so the function is correct for any slot.
template <bool BASIS>
__device__ void ground_body(const WorldView& w, const GroundParams& g, uint32_t slot)
{
    const uint32_t f0 = w.flags[slot];
    if ((f0 & kTypeMask) != 2u) return; // Dynamic bodies only (with a Transform, or orphaned): nothing else responds to a contact
    const uint32_t ci0 = w.cinfo[slot];
    if (!(ci0 & kCiGroundMask)) return; // the body's mask excludes the ground's group (StaticFilter)
    bool collide_only = false;
    if (f0 & kDrowsy) {
        // asleep: not collided (both objects inactive: btCollisionDispatcher::needsCollision), not solved.  Falling asleep at this
        // step's island build (WANTS_DEACTIVATION): isActive() is still true during performDiscreteCollisionDetection, which comes
        // first — the pair is collided once more, its manifold refreshed, and nothing is solved
        const uint32_t dz = w.deact[slot];
        if (dz == kDeactSleeping) return;
        collide_only = dz == kDeactWants;
    }
    const uint32_t cls = f0 >> kMassShift;
    float inv_mass;
    F3 force;
    if (cls != kMassClassArray) {
        const float4 gf = w.grav_palette[cls];
        inv_mass = gf.w;
        force = F3{gf.x, gf.y, gf.z};
    } else {
        inv_mass = w.inv_mass[slot];
        force = F3{g.gx / inv_mass, g.gy / inv_mass, g.gz / inv_mass};
    }
    if (inv_mass == 0.0f) return;
    const float4 cs = w.cshape[slot];
    CtShape shape;
    shape.capsule = (ci0 & kCiCapsule) != 0;
    shape.dims = F3{cs.x, cs.y, cs.z};
    int n = static_cast<int>((ci0 >> kCiCountShift) & 7u);
    const bool spin = (f0 & kSpin) != 0;
    F3 pos = ld3(w.pos, slot);
    const float breaking = ct_breaking_threshold(shape);
    if (n == 0 && !spin) {
        // cheap reject: no vertex of the shape can be within the breaking threshold of the plane
        const float reach = (__builtin_fabsf(cs.x) + __builtin_fabsf(cs.y) + __builtin_fabsf(cs.z)) * 1.01f + 0.01f;
        if (pos.y - reach > breaking) return;
    }
    Q4 q = ld4(w.quat, slot);
    M3 basis = bt_mat_from_quat(q);
    CtPoint p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p[i] = ct_empty_point();
        if (i < n) {
            const float4 a = reinterpret_cast<const float4*>(w.manifold)[8ull * slot + 2 * i];
            const float4 b = reinterpret_cast<const float4*>(w.manifold)[8ull * slot + 2 * i + 1];
            p[i].localA = F3{a.x, a.y, a.z};
            p[i].appliedImpulse = a.w;
            // localB.y is exactly 0 (the point is the projection onto y = 0): its slot carries the point's distance as the last
            // refresh left it, which sortCachedPoints reads before this step's refresh
            p[i].localB = F3{b.x, 0.0f, b.z};
            p[i].distance = b.y;
            p[i].appliedLateral = b.w;
        }
    }
    ct_collide(p, n, shape, breaking, pos, basis);
    uint32_t ci = (ci0 & ~(7u << kCiCountShift)) | (static_cast<uint32_t>(n) << kCiCountShift);
    if (collide_only) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < n) {
                reinterpret_cast<float4*>(w.manifold)[8ull * slot + 2 * i] = make_float4(p[i].localA.x, p[i].localA.y, p[i].localA.z, p[i].appliedImpulse);
                reinterpret_cast<float4*>(w.manifold)[8ull * slot + 2 * i + 1] = make_float4(p[i].localB.x, p[i].distance, p[i].localB.z, p[i].appliedLateral);
            }
        }
        if (ci != ci0) w.cinfo[slot] = ci;
        return; // k_tick puts it to sleep
    }
    if (n == 0 && !spin) {
        if (ci != ci0) w.cinfo[slot] = ci;
        return; // k_tick's plain update
    }
    F3 v = ld3(w.vel, slot);
    F3 av = spin ? ld3(w.angvel, slot) : F3{0.0f, 0.0f, 0.0f};
    if (g.want_aabb) {
        // the AABB Bullet feeds its broadphase is taken BEFORE the solver runs (predictUnconstraintMotion / updateAabbs):
        // k_tick, which runs after this kernel, would see the solved velocities — so it is written here (same arithmetic)
        const F3 he = ld3(w.half_extent, slot);
        float mn[3], mx[3];
        bt_aabb_of_pose(pos, basis, he, mn, mx);
        const F3 pp{pos.x + v.x * g.dt, pos.y + v.y * g.dt, pos.z + v.z * g.dt};
        float mn2[3], mx2[3];
        const bool turn = BASIS || spin;
        if (turn) {
            const M3 r2 = bt_mat_from_quat(bt_integrate_orientation(BASIS ? bt_quat_from_mat(basis) : q, av, g.dt));
            bt_aabb_of_pose(pp, r2, he, mn2, mx2);
        } else {
            bt_aabb_of_pose(pp, basis, he, mn2, mx2);
        }
        float* bb = w.aabb + 6ull * slot;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            bb[a] = mn2[a] < mn[a] ? mn2[a] : mn[a];
            bb[3 + a] = mx2[a] > mx[a] ? mx2[a] : mx[a];
        }
    }
    const float mass = w.cmass[slot];
    const F3 localInertia = ct_local_inertia(shape, mass);
    const F3 invInertiaLocal = ct_inv_inertia_local(localInertia);
    Q4 orn = BASIS ? bt_quat_from_mat(basis) : q;
    const bool moved = ct_solve(pos, v, av, orn, basis, p, n, inv_mass, invInertiaLocal, w.cfriction[slot], force, g.dt);
    st3(w.vel, slot, v);
    st3(w.angvel, slot, av);
    if (moved) {
        st3(w.pos, slot, pos);
        st4(w.quat, slot, orn);
        ci |= kCiMoved;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i < n) {
            reinterpret_cast<float4*>(w.manifold)[8ull * slot + 2 * i] = make_float4(p[i].localA.x, p[i].localA.y, p[i].localA.z, p[i].appliedImpulse);
            reinterpret_cast<float4*>(w.manifold)[8ull * slot + 2 * i + 1] = make_float4(p[i].localB.x, p[i].distance, p[i].localB.z, p[i].appliedLateral);
        }
    }
    w.cinfo[slot] = ci | kCiSolved;
    const bool spin_now = av.x != 0.0f || av.y != 0.0f || av.z != 0.0f;
    const uint32_t f = spin_now ? (f0 | kSpin) : (f0 & ~kSpin);
    if (f != f0) w.flags[slot] = f;
}

// ---- the obstacles' grid (GroundParams::obstacle_grid)
__device__ __forceinline__ int obs_cell(float x, float mn, float per_unit, int n)
{
    // monotone in x, clamped: two intervals that overlap map to index ranges that overlap, whatever the rounding
    const float c = (x - mn) * per_unit;
    int i = c > 0.0f ? (c < static_cast<float>(n) ? static_cast<int>(c) : n - 1) : 0;
    return i < n ? i : n - 1;
}

// One workgroup builds the whole index: bounds, counts per cell (LDS), scan, fill.  A few thousand obstacles are microseconds.
__global__ void __launch_bounds__(1024) k_obstacle_grid(GroundParams g)
{
    __shared__ uint32_t s_cnt[kObstacleGridAxis * kObstacleGridAxis];
    __shared__ uint32_t s_scan[1024];
    __shared__ float s_red[4][16];
    __shared__ uint32_t s_wide, s_bad;
    const uint32_t tid = threadIdx.x, K = g.n_obstacles;
    uint32_t* hdr = g.obstacle_grid;
    float mnx = INFINITY, mnz = INFINITY, mxx = -INFINITY, mxz = -INFINITY;
    for (uint32_t k = tid; k < K; k += 1024u) {
        const ObstacleRec& o = g.obstacles[k];
        if (!o.live || !(o.aabb[0] <= o.aabb[3]) || !(o.aabb[2] <= o.aabb[5])) continue;
        mnx = fminf(mnx, o.aabb[0]);
        mxx = fmaxf(mxx, o.aabb[3]);
        mnz = fminf(mnz, o.aabb[2]);
        mxz = fmaxf(mxz, o.aabb[5]);
    }
    for (int d = 32; d > 0; d >>= 1) {
        mnx = fminf(mnx, __shfl_down(mnx, d));
        mnz = fminf(mnz, __shfl_down(mnz, d));
        mxx = fmaxf(mxx, __shfl_down(mxx, d));
        mxz = fmaxf(mxz, __shfl_down(mxz, d));
    }
    if ((tid & 63u) == 0u) {
        s_red[0][tid >> 6] = mnx;
        s_red[1][tid >> 6] = mnz;
        s_red[2][tid >> 6] = mxx;
        s_red[3][tid >> 6] = mxz;
    }
    if (tid == 0) s_wide = s_bad = 0u;
    for (uint32_t c = tid; c < kObstacleGridAxis * kObstacleGridAxis; c += 1024u) s_cnt[c] = 0u;
    __syncthreads();
    mnx = mnz = INFINITY;
    mxx = mxz = -INFINITY;
    for (int k = 0; k < 16; ++k) {
        mnx = fminf(mnx, s_red[0][k]);
        mnz = fminf(mnz, s_red[1][k]);
        mxx = fmaxf(mxx, s_red[2][k]);
        mxz = fmaxf(mxz, s_red[3][k]);
    }
    int n = 1;
    while (n < static_cast<int>(kObstacleGridAxis) && static_cast<uint32_t>(n * n) < K) n *= 2;
    const bool any = mnx <= mxx && mnz <= mxz && mxx - mnx < INFINITY && mxz - mnz < INFINITY;
    const float ux = any && mxx > mnx ? static_cast<float>(n) / (mxx - mnx) : 0.0f, uz = any && mxz > mnz ? static_cast<float>(n) / (mxz - mnz) : 0.0f;
    // pass 1: counts (an obstacle that covers more than 64 cells goes on the wide list instead)
    for (uint32_t k = tid; k < K; k += 1024u) {
        const ObstacleRec& o = g.obstacles[k];
        if (!any || !o.live || !(o.aabb[0] <= o.aabb[3]) || !(o.aabb[2] <= o.aabb[5])) continue;
        const int x0 = obs_cell(o.aabb[0], mnx, ux, n), x1 = obs_cell(o.aabb[3], mnx, ux, n), z0 = obs_cell(o.aabb[2], mnz, uz, n), z1 = obs_cell(o.aabb[5], mnz, uz, n);
        if ((x1 - x0 + 1) * (z1 - z0 + 1) > 64) {
            const uint32_t at = atomicAdd(&s_wide, 1u);
            if (at < kObstacleGridWide) hdr[8 + at] = k;
            else s_bad = 1u;
            continue;
        }
        for (int z = z0; z <= z1; ++z) {
            for (int x = x0; x <= x1; ++x) atomicAdd(&s_cnt[z * n + x], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the n * n counts: four cells per thread, then the 1024 partial sums
    const uint32_t cells = static_cast<uint32_t>(n * n);
    uint32_t mine[4], sum = 0;
    for (int j = 0; j < 4; ++j) {
        const uint32_t c = tid * 4u + j;
        mine[j] = c < cells ? s_cnt[c] : 0u;
        sum += mine[j];
    }
    s_scan[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t v = tid >= d ? s_scan[tid - d] : 0u;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_scan[tid] - sum;
    const uint32_t total = s_scan[1023];
    uint32_t* start = hdr + kObstacleGridStart;
    for (int j = 0; j < 4; ++j) {
        const uint32_t c = tid * 4u + j;
        if (c < cells) {
            start[c] = run;
            s_cnt[c] = run; // (from here on: the cell's write cursor)
        }
        run += mine[j];
    }
    if (tid == 0) start[cells] = total;
    __syncthreads();
    const bool fits = total <= g.obstacle_grid_cap && !s_bad;
    if (fits) {
        uint32_t* items = hdr + kObstacleGridItems;
        for (uint32_t k = tid; k < K; k += 1024u) {
            const ObstacleRec& o = g.obstacles[k];
            if (!any || !o.live || !(o.aabb[0] <= o.aabb[3]) || !(o.aabb[2] <= o.aabb[5])) continue;
            const int x0 = obs_cell(o.aabb[0], mnx, ux, n), x1 = obs_cell(o.aabb[3], mnx, ux, n), z0 = obs_cell(o.aabb[2], mnz, uz, n), z1 = obs_cell(o.aabb[5], mnz, uz, n);
            if ((x1 - x0 + 1) * (z1 - z0 + 1) > 64) continue;
            for (int z = z0; z <= z1; ++z) {
                for (int x = x0; x <= x1; ++x) items[atomicAdd(&s_cnt[z * n + x], 1u)] = k;
            }
        }
    }
    if (tid == 0) {
        hdr[0] = fits ? 1u : 0u;
        hdr[1] = static_cast<uint32_t>(n);
        hdr[2] = s_wide < kObstacleGridWide ? s_wide : kObstacleGridWide;
        hdr[4] = __float_as_uint(mnx);
        hdr[5] = __float_as_uint(mnz);
        hdr[6] = __float_as_uint(ux);
        hdr[7] = __float_as_uint(uz);
    }
}

// fn(k) for every obstacle number whose fed AABB may overlap the box [x0, x1] x [z0, z1] in x and z — possibly more than once and in
// no particular order; through the grid when it is valid and the box covers few cells, otherwise all of them.  fn returns true to stop.
template <class Fn>
__device__ __forceinline__ void for_each_obstacle_near(const GroundParams& g, float x0, float x1, float z0, float z1, Fn fn)
{
    const uint32_t* hdr = g.obstacle_grid;
    if (hdr && hdr[0]) {
        const int n = static_cast<int>(hdr[1]);
        const float mnx = __uint_as_float(hdr[4]), mnz = __uint_as_float(hdr[5]), ux = __uint_as_float(hdr[6]), uz = __uint_as_float(hdr[7]);
        const int cx0 = obs_cell(x0, mnx, ux, n), cx1 = obs_cell(x1, mnx, ux, n), cz0 = obs_cell(z0, mnz, uz, n), cz1 = obs_cell(z1, mnz, uz, n);
        if (x0 <= x1 && z0 <= z1 && (cx1 - cx0 + 1) * (cz1 - cz0 + 1) <= 64) {
            const uint32_t n_wide = hdr[2];
            for (uint32_t j = 0; j < n_wide; ++j) {
                if (fn(hdr[8 + j])) return;
            }
            const uint32_t* start = hdr + kObstacleGridStart;
            const uint32_t* items = hdr + kObstacleGridItems;
            for (int z = cz0; z <= cz1; ++z) {
                for (int x = cx0; x <= cx1; ++x) {
                    const uint32_t b = start[z * n + x], e = start[z * n + x + 1];
                    for (uint32_t at = b; at < e; ++at) {
                        if (fn(items[at])) return;
                    }
                }
            }
            return;
        }
    }
    for (uint32_t k = 0; k < g.n_obstacles; ++k) {
        if (fn(k)) return;
    }
}

// ---- two launches per sub-step
// ground_body needs 246 VGPRs (two waves per SIMD) — and most bodies of a scene need none of it: they sleep, or are nowhere
// near the plane.  As ONE kernel over all slots (the first version) even those paid for the solver's occupancy: two waves per
// SIMD cannot keep enough loads in flight, and the tests sat behind five dependent round trips (flags -> contact word ->
// palette -> shape -> position).  1 M bodies, per tick on top of the 24.5 us tick: asleep +38 us, airborne +22 us.
//   k_ground_select  256 threads, a handful of registers: the teleport rule of PhysicsSystem::Update for dirty bodies (what
//                    k_pose_only did as a third launch), then the tests in two batches of loads — flags + contact word +
//                    deactivation record; shape + position for those still in — and the slots that need the solver appended
//                    to a list (one atomic per wave)
//   k_ground         a resident-sized grid walks the list: full waves of bodies that are all in contact
// The list's order depends on the atomics, the results do not: a body touches nothing but its own records.
template <bool BASIS>
__global__ void __launch_bounds__(256) k_ground_select(WorldView w, GroundParams g)
{
    const uint64_t slot64 = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    const bool in_range = slot64 < g.n_slots;
    const uint32_t slot = in_range ? static_cast<uint32_t>(slot64) : 0u;
    uint32_t f = w.flags[slot];
    const uint32_t ci0 = w.cinfo[slot];
    uint32_t dz = w.deact[slot]; // (read whether or not kDrowsy says it is meaningful: one round trip instead of two)
    asm volatile("" : "+v"(f), "+v"(dz));
    bool need = in_range;
    const uint32_t type = f & kTypeMask;
    if (need && g.repose && (f & kValid) && type != 0 && (f & (kTDirty | kBDirty))) {
        // EnsureRigidBody / SyncKinematicBodiesToPhysics before stepSimulation (PhysicsSystem.cpp:952-989): pose from the Transform,
        // zero velocities — k_pose_only's re-pose, for the body types it applies to; the tick kernel that follows (no_repose) marks
        // Dynamic transforms dirty and clears kBDirty as always
        const uint32_t f_in = f;
        const Q4 q = bt_quat_from_transform_euler(ld3(w.euler, slot));
        st4(w.quat, slot, q);
        f &= ~kSettled;
        const F3 zero{0.0f, 0.0f, 0.0f};
        if (type == 2u) st3(w.vel, slot, zero);
        if (f & kSpin) {
            st3(w.angvel, slot, zero);
            f &= ~kSpin;
        }
        if (type == 2u) st3(w.euler, slot, bt_transform_euler_from_mat(bt_mat_from_quat(q)));
        if (f != f_in) w.flags[slot] = f;
    }
    // (a body that wants to sleep is still collided this step — only a sleeping one is skipped)
    // (a body of an island of several bodies that stays awake is collided and solved by the island kernels, which ran before)
    const bool awake = in_range && type == 2u && !((f & kDrowsy) && dz == kDeactSleeping) && !(ci0 & kCiIsland);
    // Static / Kinematic box colliders on: a Dynamic BOX that holds manifolds with boxes, or whose reach (conservative: the L1 norm
    // of its half extents bounds its AABB at any orientation, plus this step's motion, plus Bullet's 0.02) touches an obstacle's
    // fed AABB, goes to k_contact_boxes — which decides the pairs exactly and handles the plane for that body too
    bool boxes = false;
    if (g.n_obstacles != 0u || (ci0 & kCiBoxes)) {
        if (awake && !(ci0 & kCiCapsule)) {
            boxes = (ci0 & kCiBoxes) != 0;
            if (!boxes) {
                const float4 cs = w.cshape[slot];
                const F3 pos = ld3(w.pos, slot);
                const F3 v = ld3(w.vel, slot);
                const float reach = (__builtin_fabsf(cs.x) + __builtin_fabsf(cs.y) + __builtin_fabsf(cs.z)) * 1.01f + 0.05f;
                const float rx = reach + __builtin_fabsf(v.x) * g.dt * 1.01f, ry = reach + __builtin_fabsf(v.y) * g.dt * 1.01f,
                            rz = reach + __builtin_fabsf(v.z) * g.dt * 1.01f;
                for_each_obstacle_near(g, pos.x - rx, pos.x + rx, pos.z - rz, pos.z + rz, [&](uint32_t k) {
                    const float* bb = g.obstacles[k].aabb;
                    boxes = pos.x - rx <= bb[3] && pos.x + rx >= bb[0] && pos.y - ry <= bb[4] && pos.y + ry >= bb[1] && pos.z - rz <= bb[5] &&
                            pos.z + rz >= bb[2];
                    return boxes;
                });
            }
        }
    }
    need = awake && !boxes && g.plane != 0u && (ci0 & kCiGroundMask) != 0;
    if (__any(need)) {
        const uint32_t n = (ci0 >> kCiCountShift) & 7u;
        if (need && n == 0u && !(f & kSpin)) {
            // cheap reject, as in ground_body: no vertex of the shape can be within the breaking threshold of the plane
            const float4 cs = w.cshape[slot];
            const F3 pos = ld3(w.pos, slot);
            CtShape shape;
            shape.capsule = (ci0 & kCiCapsule) != 0;
            shape.dims = F3{cs.x, cs.y, cs.z};
            const float reach = (__builtin_fabsf(cs.x) + __builtin_fabsf(cs.y) + __builtin_fabsf(cs.z)) * 1.01f + 0.01f;
            if (pos.y - reach > ct_breaking_threshold(shape)) need = false;
        }
    }
    {
        const unsigned long long mb = __ballot(boxes);
        if (mb != 0) { // (rare: one list, one atomic per wave)
            const uint32_t lane = threadIdx.x & 63u;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&g.box_count[0], static_cast<uint32_t>(__popcll(mb)));
            base = __shfl(base, 0, 64);
            if (boxes) g.box_list[base + static_cast<uint32_t>(__popcll(mb & ((1ull << lane) - 1ull)))] = slot;
        }
    }
    const unsigned long long m = __ballot(need);
    if (m == 0) return;
    // one atomic per wave, on the counter of this workgroup's shard: a single counter word takes ~10^8 atomics a second, and
    // 15,625 waves of a million resting bodies queued on it for 130 us.  Workgroup b appends to shard b % kGroundShards, whose
    // segment holds every slot those workgroups could ever send: it cannot overflow.
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t shard = blockIdx.x % kGroundShards;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&g.list_count[16u * shard], static_cast<uint32_t>(__popcll(m)));
    base = __shfl(base, 0, 64);
    const uint32_t at = base + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
    if (need && at < g.shard_cap) g.list[static_cast<uint64_t>(shard) * g.shard_cap + at] = slot; // (the bound cannot bite while the counts are reset: a guard, not a path)
}

template <bool BASIS>
#ifndef BGE_GROUND_MIN_BLOCKS
#define BGE_GROUND_MIN_BLOCKS 2 /* waves per SIMD the register budget is set for: 2 -> 246 VGPRs, no scratch; 4 -> 128 VGPRs and spills (measured slower) */
#endif
__global__ void __launch_bounds__(128, BGE_GROUND_MIN_BLOCKS) k_ground(WorldView w, GroundParams g)
{
    // workgroup b works on shard b % kGroundShards, together with the other gridDim.x / kGroundShards workgroups of that shard
    const uint32_t shard = blockIdx.x % kGroundShards;
    const uint32_t n_list = min(g.list_count[16u * shard], static_cast<uint32_t>(g.shard_cap));
    const uint32_t* list = g.list + static_cast<uint64_t>(shard) * g.shard_cap;
    const uint32_t step = (gridDim.x / kGroundShards) * blockDim.x;
#ifndef BGE_GROUND_EMPTY /* timing experiment: the launch without the solver (and so without scratch) */
    for (uint32_t i = (blockIdx.x / kGroundShards) * blockDim.x + threadIdx.x; i < n_list; i += step) {
        const uint32_t slot = list[i];
        if (slot < g.n_slots) ground_body<BASIS>(w, g, slot);
    }
#else
    if (n_list == 0xffffffffu) w.cinfo[list[step]] = 0;
#endif
    // the workgroup that draws the shard's last ticket empties its list for the next sub-step's k_ground_select: by then every
    // workgroup of the shard has read the count (it did so before it drew its own ticket).  (One ticket word for all 1024
    // workgroups made an EMPTY launch take 13.6 us: a thousand atomics on one address.)
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(&g.list_count[16u * shard + 1u], 1u) == gridDim.x / kGroundShards - 1u) {
            g.list_count[16u * shard] = 0;
            g.list_count[16u * shard + 1u] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 3: Dynamic boxes on the Static / Kinematic BOX colliders of the scene (Bullet's btBoxBoxCollisionAlgorithm; the reference's
// demo.json "Ground" is one).  oracle/boxbox_ref.h + physics_ref.h (CollideWithBoxes, ct::SolveBody) are the specification; the
// choices Bullet's history-dependence forces are stated there (the Dynamic body is body A; pairs = fed AABBs overlap + filter; at
// most kBoxManifolds manifolds, lowest entity ids).  Capsules against boxes (GJK) are not built.
//   k_obstacles       one thread per Static / Kinematic box body (a compact list the host keeps, ascending entity): pose as Bullet
//                     holds it at this sub-step (a dirty body's from its Transform), fed AABB, material -> ObstacleRec
//   k_ground_select   routes a Dynamic box that holds box manifolds, or whose reach touches an obstacle's AABB, to box_list
//   k_contact_boxes   one thread per listed body: exact pairs, box-box detector into the body's persistent manifolds (rows of
//                     bmanifold, kept in global memory), the plane manifold as k_ground has it, then ONE solver for the island
//                     {body}: plane rows, then every box manifold's rows in ascending entity — with rows in scratch memory and
//                     loops, not unrolled registers: this kernel serves the handful of bodies that rest on a static box, and is
//                     sized for correctness (k_ground stays the fast path for everything that touches only the plane).
__global__ void __launch_bounds__(64) k_obstacles(WorldView w, GroundParams g)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= g.n_obstacles) return;
    const uint32_t slot = g.obstacle_slots[k];
    ObstacleRec r;
    const uint32_t f = w.flags[slot];
    const uint32_t type = f & kTypeMask;
    const uint32_t ci = w.cinfo[slot];
    r.live = (type == 1u || type == 3u) && !(ci & kCiCapsule) ? 1u : 0u;
    const F3 pos = ld3(w.pos, slot);
    // SyncKinematicBodiesToPhysics runs before the step: a dirty body is posed from its Transform (k_ground_select / k_tick store that
    // quaternion; this kernel runs before them)
    const bool repose = g.repose && (f & kValid) && (f & (kTDirty | kBDirty));
    const Q4 q = repose ? bt_quat_from_transform_euler(ld3(w.euler, slot)) : ld4(w.quat, slot);
    const M3 basis = bt_mat_from_quat(q);
    const float4 cs = w.cshape[slot];
    CtShape shape;
    shape.capsule = false;
    shape.dims = F3{cs.x, cs.y, cs.z};
    float mn[3], mx[3];
    bt_aabb_of_pose(pos, basis, ld3(w.half_extent, slot), mn, mx);
    r.origin[0] = pos.x; r.origin[1] = pos.y; r.origin[2] = pos.z;
    r.half[0] = cs.x; r.half[1] = cs.y; r.half[2] = cs.z;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) r.basis[3 * i + j] = basis.m[i][j];
        r.aabb[i] = mn[i];
        r.aabb[3 + i] = mx[i];
    }
    r.friction = w.cfriction[slot];
    r.restitution = w.crestitution[slot];
    r.breaking = ct_breaking_threshold(shape);
    r.entity = g.entity_of_slot[slot];
    r.group = w.group[slot];
    r.mask = w.mask[slot];
    r.generation = g.obstacle_gen[k];
    r.pad[0] = r.pad[1] = r.pad[2] = 0u;
    g.obstacles[k] = r;
}

// a box manifold's points live in global memory: 12 floats each (localA, localB, normalWorldOnB, distance, appliedImpulse, lateral)
__device__ __forceinline__ F3 bp_get3(const float* p, int at) { return F3{p[at], p[at + 1], p[at + 2]}; }
__device__ __forceinline__ void bp_put3(float* p, int at, const F3& v)
{
    p[at] = v.x;
    p[at + 1] = v.y;
    p[at + 2] = v.z;
}

// btPersistentManifold::sortCachedPoints on a full row (oracle/boxbox_ref.h SortCachedBoxPoints)
__device__ int bp_sort_cached_points(const float* pts, const F3& newLocalA, float newDistance)
{
    int maxPenetrationIndex = -1;
    float maxPenetration = newDistance;
    for (int i = 0; i < 4; ++i) {
        const float d = pts[12 * i + 9];
        if (d < maxPenetration) {
            maxPenetrationIndex = i;
            maxPenetration = d;
        }
    }
    float res[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const F3 p0 = bp_get3(pts, 0), p1 = bp_get3(pts, 12), p2 = bp_get3(pts, 24), p3 = bp_get3(pts, 36);
    if (maxPenetrationIndex != 0) {
        const F3 c = cross3(sub3(newLocalA, p1), sub3(p3, p2));
        res[0] = dot3(c, c);
    }
    if (maxPenetrationIndex != 1) {
        const F3 c = cross3(sub3(newLocalA, p0), sub3(p3, p2));
        res[1] = dot3(c, c);
    }
    if (maxPenetrationIndex != 2) {
        const F3 c = cross3(sub3(newLocalA, p0), sub3(p3, p1));
        res[2] = dot3(c, c);
    }
    if (maxPenetrationIndex != 3) {
        const F3 c = cross3(sub3(newLocalA, p0), sub3(p2, p1));
        res[3] = dot3(c, c);
    }
    int maxIndex = -1;
    float maxVal = -1.0e18f;
    for (int i = 0; i < 4; ++i) {
        const float a = __builtin_fabsf(res[i]);
        if (a > maxVal) {
            maxIndex = i;
            maxVal = a;
        }
    }
    return maxIndex;
}

// btBoxBoxCollisionAlgorithm::processCollision, body0 = the Dynamic box (oracle/boxbox_ref.h CollideBoxBox); returns the point count
__device__ int bp_collide(float* pts, int n, float breaking, const F3& originA, const M3& basisA, const F3& halfA, const ObstacleRec& o)
{
    const F3 originB = F3{o.origin[0], o.origin[1], o.origin[2]};
    M3 basisB;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) basisB.m[i][j] = o.basis[3 * i + j];
    }
    boxbox::Out out;
    boxbox::box_box(originA, basisA, halfA, originB, basisB, F3{o.half[0], o.half[1], o.half[2]}, out);
    for (int k = 0; k < out.n; ++k) {
        const float depth = out.depth[k];
        if (depth > breaking) continue;
        const F3 normalOnB = out.normalOnB;
        const F3 pointInWorld = out.point[k];
        const F3 pointA = add3(pointInWorld, scale3(normalOnB, depth));
        const F3 localA = mat_t_vec(basisA, sub3(pointA, originA));
        const F3 localB = mat_t_vec(basisB, sub3(pointInWorld, originB));
        float shortest = breaking * breaking;
        int nearest = -1;
        for (int i = 0; i < n; ++i) {
            const F3 diffA = sub3(bp_get3(pts, 12 * i), localA);
            const float d2 = dot3(diffA, diffA);
            if (d2 < shortest) {
                shortest = d2;
                nearest = i;
            }
        }
        float applied = 0.0f, lateral = 0.0f;
        int insert;
        if (nearest >= 0) {
            insert = nearest;
            applied = pts[12 * nearest + 10];
            lateral = pts[12 * nearest + 11];
        } else {
            insert = n;
            if (insert == 4) {
                insert = bp_sort_cached_points(pts, localA, depth);
            } else {
                n++;
            }
            if (insert < 0) insert = 0;
        }
        float* d = pts + 12 * insert;
        bp_put3(d, 0, localA);
        bp_put3(d, 3, localB);
        bp_put3(d, 6, normalOnB);
        d[9] = depth;
        d[10] = applied;
        d[11] = lateral;
    }
    // refreshContactPoints(body0 transform, body1 transform)
    for (int i = n - 1; i >= 0; --i) {
        float* c = pts + 12 * i;
        const F3 worldA = xform_point(basisA, originA, bp_get3(c, 0));
        const F3 worldB = xform_point_b(basisB, originB, bp_get3(c, 3));
        c[9] = dot3(sub3(worldA, worldB), bp_get3(c, 6));
    }
    for (int i = n - 1; i >= 0; --i) {
        float* c = pts + 12 * i;
        const float distance = c[9];
        bool remove = !(distance <= breaking);
        if (!remove) {
            const F3 nB = bp_get3(c, 6);
            const F3 worldA = xform_point(basisA, originA, bp_get3(c, 0));
            const F3 worldB = xform_point_b(basisB, originB, bp_get3(c, 3));
            const F3 projectedPoint = sub3(worldA, scale3(nB, distance));
            const F3 projectedDifference = sub3(worldB, projectedPoint);
            const float distance2d = dot3(projectedDifference, projectedDifference);
            remove = distance2d > breaking * breaking;
        }
        if (remove) {
            const int last = n - 1;
            if (i != last) {
                for (int k = 0; k < 12; ++k) c[k] = pts[12 * last + k];
            }
            for (int k = 0; k < 12; ++k) pts[12 * last + k] = 0.0f;
            n--;
        }
    }
    return n;
}

// btPlaneSpace1, first tangent
__device__ __forceinline__ F3 ct_plane_space1(const F3& n)
{
    if (__builtin_fabsf(n.z) > 0.7071067811865475244008443621048490f) {
        const float a = n.y * n.y + n.z * n.z;
        const float k = 1.0f / __builtin_sqrtf(a);
        return F3{0.0f, -n.z * k, n.y * k};
    }
    const float a = n.x * n.x + n.y * n.y;
    const float k = 1.0f / __builtin_sqrtf(a);
    return F3{-n.y * k, n.x * k, 0.0f};
}

constexpr int kMaxContactRows = 4 * (1 + static_cast<int>(kBoxManifolds));

// One contact's rows (setupContactConstraint + the friction row of convertContact), appended to the island's pools and warm
// started: oracle/boxbox_ref.h SolveBody's loop body, one operation after the other
__device__ void ct_add_contact(CtBody& sb, CtRow* normalRow, CtRow* frictionRow, int j, const F3& origin, const F3& bodyLinVel, const F3& bodyAngVel,
                               const M3& invI, float invMassScalar, float invTimeStep, const F3& worldA, const F3& n, float distance,
                               float friction, float combinedRestitution, float appliedIn, float lateralIn)
{
    constexpr float kErp2 = 0.2f, kSplitThreshold = -0.04f, kWarmstart = 0.85f, kSor = 1.0f, kRestitutionVelocityThreshold = 0.2f;
    CtRow c = ct_zero_row();
    const F3 rel_pos1 = sub3(worldA, origin);
    const F3 vel1 = add3(add3(sb.linVel, sb.extForce), cross3(add3(sb.angVel, sb.extTorque), rel_pos1));
    const F3 vel = sub3(vel1, F3{0.0f, 0.0f, 0.0f});
    const float rel_vel = dot3(n, vel);
    const float relaxation = kSor;
    const F3 torqueAxis0 = cross3(rel_pos1, n);
    c.angularComp = mat_vec(invI, torqueAxis0);
    {
        const F3 vec = cross3(c.angularComp, rel_pos1);
        const float denom0 = inv_mass_plus_dot(invMassScalar, n, vec);
        const float cfm0 = 0.0f * invTimeStep;
        c.jacDiagABInv = relaxation / (denom0 + 0.0f + cfm0);
    }
    c.normal = n;
    c.relposCrossN = torqueAxis0;
    const float penetration = distance + 0.0f;
    c.friction = friction;
    float restitution = 0.0f;
    if (combinedRestitution != 0.0f) {
        const F3 rbVel = add3(bodyLinVel, cross3(bodyAngVel, rel_pos1));
        const float rbRelVel = dot3(n, sub3(rbVel, F3{0.0f, 0.0f, 0.0f}));
        restitution = __builtin_fabsf(rbRelVel) < kRestitutionVelocityThreshold ? 0.0f : combinedRestitution * -rbRelVel;
        if (restitution <= 0.0f) restitution = 0.0f;
    }
    c.applied = appliedIn * kWarmstart;
    {
        const F3 lin = F3{c.normal.x * sb.invMass.x, c.normal.y * sb.invMass.y, c.normal.z * sb.invMass.z};
        sb.dLin = add3(sb.dLin, scale3(lin, c.applied));
        sb.dAng = add3(sb.dAng, scale3(c.angularComp, c.applied * 1.0f));
    }
    c.appliedPush = 0.0f;
    {
        const float vel1Dotn = dot_xzy(c.normal, add3(sb.linVel, sb.extForce)) + dot_xzy(c.relposCrossN, add3(sb.angVel, sb.extTorque));
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
    F3 dir = sub3(vel, scale3(n, rel_vel));
    const float lat_rel_vel = dot3(dir, dir);
    if (lat_rel_vel > kBtEpsilon) {
        dir = scale3(dir, 1.0f / __builtin_sqrtf(lat_rel_vel));
    } else {
        dir = ct_plane_space1(n);
    }
    CtRow fr = ct_zero_row();
    fr.friction = friction;
    fr.normal = dir;
    fr.relposCrossN = cross3(rel_pos1, dir);
    fr.angularComp = mat_vec(invI, fr.relposCrossN);
    {
        const F3 vec = cross3(fr.angularComp, rel_pos1);
        const float denom0 = inv_mass_plus_dot(invMassScalar, dir, vec);
        fr.jacDiagABInv = relaxation / (denom0 + 0.0f);
    }
    {
        const float vel1Dotn = dot_xzy(fr.normal, add3(sb.linVel, sb.extForce)) + dot_xzy(fr.relposCrossN, sb.angVel);
        const float vel2Dotn = 0.0f + 0.0f;
        const float rv = vel1Dotn + vel2Dotn;
        const float velocityError = 0.0f - rv;
        const float velocityImpulse = velocityError * fr.jacDiagABInv;
        fr.rhs = 0.0f + velocityImpulse;
        fr.rhsPenetration = 0.0f;
        fr.cfm = 0.0f;
        fr.lower = -fr.friction;
        fr.upper = fr.friction;
    }
    fr.applied = 0.0f; // (not warm-started: see ct_solve)
    normalRow[j] = c;
    frictionRow[j] = fr;
}

// `island`: the body belongs to an island of several bodies — its own pairs (plane, Static / Kinematic boxes) are collided here, the
// island's solver thread does the rest (k_island_solve)
template <bool BASIS>
__device__ void contact_body(const WorldView& w, const GroundParams& g, uint32_t slot, bool island = false)
{
    const uint32_t f0 = w.flags[slot];
    if ((f0 & kTypeMask) != 2u) return;
    const uint32_t ci0 = w.cinfo[slot];
    if (ci0 & kCiCapsule) return; // (never routed here)
    bool collide_only = island;
    if (f0 & kDrowsy) {
        const uint32_t dz = w.deact[slot];
        if (dz == kDeactSleeping) return;
        collide_only = island || dz == kDeactWants;
    }
    const uint32_t cls = f0 >> kMassShift;
    float inv_mass;
    F3 force;
    if (cls != kMassClassArray) {
        const float4 gf = w.grav_palette[cls];
        inv_mass = gf.w;
        force = F3{gf.x, gf.y, gf.z};
    } else {
        inv_mass = w.inv_mass[slot];
        force = F3{g.gx / inv_mass, g.gy / inv_mass, g.gz / inv_mass};
    }
    if (inv_mass == 0.0f) return;
    const float4 cs = w.cshape[slot];
    CtShape shape;
    shape.capsule = false;
    shape.dims = F3{cs.x, cs.y, cs.z};
    int n = static_cast<int>((ci0 >> kCiCountShift) & 7u);
    const bool spin = (f0 & kSpin) != 0;
    F3 pos = ld3(w.pos, slot);
    const float breaking = ct_breaking_threshold(shape);
    Q4 q = ld4(w.quat, slot);
    M3 basis = bt_mat_from_quat(q);
    F3 v = ld3(w.vel, slot);
    F3 av = spin ? ld3(w.angvel, slot) : F3{0.0f, 0.0f, 0.0f};
    // the AABB Bullet feeds its broadphase (predictUnconstraintMotion / updateAabbs: pose and velocity as the sub-step starts)
    float fed_mn[3], fed_mx[3];
    {
        const F3 he = ld3(w.half_extent, slot);
        float mn[3], mx[3], mn2[3], mx2[3];
        bt_aabb_of_pose(pos, basis, he, mn, mx);
        const F3 pp{pos.x + v.x * g.dt, pos.y + v.y * g.dt, pos.z + v.z * g.dt};
        if (BASIS || spin) {
            const M3 r2 = bt_mat_from_quat(bt_integrate_orientation(BASIS ? bt_quat_from_mat(basis) : q, av, g.dt));
            bt_aabb_of_pose(pp, r2, he, mn2, mx2);
        } else {
            bt_aabb_of_pose(pp, basis, he, mn2, mx2);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            fed_mn[a] = mn2[a] < mn[a] ? mn2[a] : mn[a];
            fed_mx[a] = mx2[a] > mx[a] ? mx2[a] : mx[a];
        }
    }
    // ---- the plane (k_ground's ground_body, for this body)
    const bool plane_ok = g.plane != 0u && (ci0 & kCiGroundMask) != 0;
    CtPoint p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p[i] = ct_empty_point();
        if (plane_ok && i < n) {
            const float4 a = reinterpret_cast<const float4*>(w.manifold)[8ull * slot + 2 * i];
            const float4 b = reinterpret_cast<const float4*>(w.manifold)[8ull * slot + 2 * i + 1];
            p[i].localA = F3{a.x, a.y, a.z};
            p[i].appliedImpulse = a.w;
            p[i].localB = F3{b.x, 0.0f, b.z};
            p[i].distance = b.y;
            p[i].appliedLateral = b.w;
        }
    }
    if (plane_ok) ct_collide(p, n, shape, breaking, pos, basis);
    // ---- the boxes: exact pairs (fed AABBs overlap non-strictly, filter both ways), the kBoxManifolds lowest entities
    uint32_t* rows = w.bmanifold + static_cast<uint64_t>(slot) * (kBoxManifolds * kBoxManifoldWords);
    const bool rows_live = (ci0 & kCiBoxes) != 0;
    const uint32_t my_entity = g.entity_of_slot[slot];
    const uint32_t grp = w.group[slot], msk = w.mask[slot];
    uint32_t accepted[kBoxManifolds];
    int row_of[kBoxManifolds];
    int n_acc = 0;
    // (the obstacle numbers ascend with the entity ids: the kBoxManifolds LOWEST partners are kept, in ascending order, whatever order
    //  the candidates come in and however often)
    for_each_obstacle_near(g, fed_mn[0], fed_mx[0], fed_mn[2], fed_mx[2], [&](uint32_t k) {
        const ObstacleRec& o = g.obstacles[k];
        if (!o.live || o.entity == my_entity) return false;
        if ((grp & o.mask) == 0u || (o.group & msk) == 0u) return false;
        const bool overlap = fed_mn[0] <= o.aabb[3] && fed_mx[0] >= o.aabb[0] && fed_mn[1] <= o.aabb[4] && fed_mx[1] >= o.aabb[1] &&
                             fed_mn[2] <= o.aabb[5] && fed_mx[2] >= o.aabb[2];
        if (!overlap) return false;
        int at = 0;
        while (at < n_acc && accepted[at] < k) ++at;
        if (at < n_acc && accepted[at] == k) return false;                       // seen in another cell
        if (at >= static_cast<int>(kBoxManifolds)) return false;                 // four lower ones are known already
        const int last = n_acc < static_cast<int>(kBoxManifolds) ? n_acc : static_cast<int>(kBoxManifolds) - 1;
        for (int j = last; j > at; --j) accepted[j] = accepted[j - 1];
        accepted[at] = k;
        if (n_acc < static_cast<int>(kBoxManifolds)) n_acc++;
        return false;
    });
    for (int a = 0; a < n_acc; ++a) row_of[a] = -1;
    // a manifold lives as long as its pair: rows whose box is no longer a partner (or was re-created) are freed
    uint32_t row_used = 0;
    for (uint32_t e = 0; e < kBoxManifolds; ++e) {
        uint32_t* hdr = rows + e * kBoxManifoldWords;
        bool keep = false;
        if (rows_live && hdr[0] != kBoxNone) {
            for (int a = 0; a < n_acc; ++a) {
                const ObstacleRec& o = g.obstacles[accepted[a]];
                if (o.entity == hdr[0] && o.generation == hdr[2]) {
                    row_of[a] = static_cast<int>(e);
                    keep = true;
                }
            }
        }
        if (keep) {
            row_used |= 1u << e;
        } else if (!rows_live || hdr[0] != kBoxNone) {
            hdr[0] = kBoxNone;
            hdr[1] = 0u;
        }
    }
    bool touching = plane_ok && n > 0;
    for (int a = 0; a < n_acc; ++a) {
        const ObstacleRec& o = g.obstacles[accepted[a]];
        if (row_of[a] < 0) {
            uint32_t e = 0;
            while (row_used & (1u << e)) ++e; // (n_acc <= kBoxManifolds: there is a free row)
            row_of[a] = static_cast<int>(e);
            row_used |= 1u << e;
            uint32_t* hdr = rows + e * kBoxManifoldWords;
            hdr[0] = o.entity;
            hdr[1] = 0u;
            hdr[2] = o.generation;
            hdr[3] = 0u;
        }
        uint32_t* hdr = rows + static_cast<uint32_t>(row_of[a]) * kBoxManifoldWords;
        const float pair_breaking = fminf(breaking, o.breaking); // btCollisionDispatcher::getNewManifold
        const int np = bp_collide(reinterpret_cast<float*>(hdr + 4), static_cast<int>(hdr[1]), pair_breaking, pos, basis, shape.dims, o);
        hdr[1] = static_cast<uint32_t>(np);
        touching = touching || np > 0;
    }
    uint32_t ci = (ci0 & ~((7u << kCiCountShift) | kCiBoxes)) | (static_cast<uint32_t>(n) << kCiCountShift) | (n_acc > 0 ? kCiBoxes : 0u);
    auto store_plane = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (plane_ok && i < n) {
                reinterpret_cast<float4*>(w.manifold)[8ull * slot + 2 * i] = make_float4(p[i].localA.x, p[i].localA.y, p[i].localA.z, p[i].appliedImpulse);
                reinterpret_cast<float4*>(w.manifold)[8ull * slot + 2 * i + 1] = make_float4(p[i].localB.x, p[i].distance, p[i].localB.z, p[i].appliedLateral);
            }
        }
    };
    if (collide_only || !((plane_ok || n_acc > 0) && (touching || spin))) {
        store_plane();
        if (ci != ci0) w.cinfo[slot] = ci;
        return; // k_tick puts it to sleep / takes the plain update
    }
    if (g.want_aabb) {
        float* bb = w.aabb + 6ull * slot;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            bb[a] = fed_mn[a];
            bb[3 + a] = fed_mx[a];
        }
    }
    // ---- solveGroup for the island {body} (oracle/boxbox_ref.h SolveBody)
    constexpr int kIterations = 10;
    constexpr float kSplitTurnErp = 0.1f;
    const float mass = w.cmass[slot];
    const F3 localInertia = ct_local_inertia(shape, mass);
    const F3 invInertiaLocal = ct_inv_inertia_local(localInertia);
    Q4 orn = BASIS ? bt_quat_from_mat(basis) : q;
    const M3 invI = ct_inv_inertia_world(basis, invInertiaLocal);
    const float bodyFriction = w.cfriction[slot], bodyRestitution = w.crestitution[slot];
    CtBody sb;
    sb.dLin = sb.dAng = sb.push = sb.turn = F3{0.0f, 0.0f, 0.0f};
    sb.invMass = F3{inv_mass, inv_mass, inv_mass};
    sb.linVel = v;
    sb.angVel = av;
    sb.extForce = scale3(scale3(force, inv_mass), g.dt);
    sb.extTorque = F3{0.0f, 0.0f, 0.0f};
    sb.extTorque = add3(sb.extTorque, ct_gyroscopic_impulse(invInertiaLocal, av, orn, g.dt));
    CtRow normalRow[kMaxContactRows], frictionRow[kMaxContactRows];
    const float invTimeStep = 1.0f / g.dt;
    int n_rows = 0;
    if (plane_ok) {
        const float combinedFriction = fmaxf(-10.0f, fminf(10.0f, bodyFriction * 1.0f));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < n) {
                ct_add_contact(sb, normalRow, frictionRow, n_rows, pos, v, av, invI, inv_mass, invTimeStep, p[j].worldA, F3{0.0f, 1.0f, 0.0f}, p[j].distance,
                               combinedFriction, 0.0f, p[j].appliedImpulse, p[j].appliedLateral);
                n_rows++;
            }
        }
    }
    for (int a = 0; a < n_acc; ++a) { // (accepted is in ascending entity: the island's manifold order)
        const ObstacleRec& o = g.obstacles[accepted[a]];
        const uint32_t* hdr = rows + static_cast<uint32_t>(row_of[a]) * kBoxManifoldWords;
        const float* pts = reinterpret_cast<const float*>(hdr + 4);
        const float combinedFriction = fmaxf(-10.0f, fminf(10.0f, bodyFriction * o.friction)); // btManifoldResult::calculateCombinedFriction
        const float combinedRestitution = bodyRestitution * o.restitution;
        const int np = static_cast<int>(hdr[1]);
        for (int j = 0; j < np; ++j) {
            const float* c = pts + 12 * j;
            const F3 worldA = xform_point(basis, pos, bp_get3(c, 0)); // what refreshContactPoints left in m_positionWorldOnA
            ct_add_contact(sb, normalRow, frictionRow, n_rows, pos, v, av, invI, inv_mass, invTimeStep, worldA, bp_get3(c, 6), c[9], combinedFriction,
                           combinedRestitution, c[10], c[11]);
            n_rows++;
        }
    }
#pragma unroll 1
    for (int it = 0; it < kIterations; ++it) {
#pragma unroll 1
        for (int j = 0; j < n_rows; ++j) ct_resolve_split(sb, normalRow[j]);
    }
#pragma unroll 1
    for (int it = 0; it < kIterations; ++it) {
#pragma unroll 1
        for (int j = 0; j < n_rows; ++j) ct_resolve_row(sb, normalRow[j], false);
#pragma unroll 1
        for (int j = 0; j < n_rows; ++j) {
            const float totalImpulse = normalRow[j].applied;
            if (totalImpulse > 0.0f) {
                frictionRow[j].lower = -(frictionRow[j].friction * totalImpulse);
                frictionRow[j].upper = frictionRow[j].friction * totalImpulse;
                ct_resolve_row(sb, frictionRow[j], true);
            }
        }
    }
    {
        int j = 0;
        if (plane_ok) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i < n) {
                    p[i].appliedImpulse = normalRow[j].applied;
                    p[i].appliedLateral = frictionRow[j].applied;
                    j++;
                }
            }
        }
        for (int a = 0; a < n_acc; ++a) {
            uint32_t* hdr = rows + static_cast<uint32_t>(row_of[a]) * kBoxManifoldWords;
            float* pts = reinterpret_cast<float*>(hdr + 4);
            const int np = static_cast<int>(hdr[1]);
            for (int i = 0; i < np; ++i) {
                pts[12 * i + 10] = normalRow[j].applied;
                pts[12 * i + 11] = frictionRow[j].applied;
                j++;
            }
        }
    }
    sb.linVel = add3(sb.linVel, sb.dLin);
    sb.angVel = add3(sb.angVel, sb.dAng);
    bool moved = false;
    if (sb.push.x != 0.0f || sb.push.y != 0.0f || sb.push.z != 0.0f || sb.turn.x != 0.0f || sb.turn.y != 0.0f || sb.turn.z != 0.0f) {
        pos = add3(pos, scale3(sb.push, g.dt));
        orn = bt_integrate_orientation(orn, scale3(sb.turn, kSplitTurnErp), g.dt);
        moved = true;
    }
    v = add3(sb.linVel, sb.extForce);
    av = add3(sb.angVel, sb.extTorque);
    st3(w.vel, slot, v);
    st3(w.angvel, slot, av);
    if (moved) {
        st3(w.pos, slot, pos);
        st4(w.quat, slot, orn);
        ci |= kCiMoved;
    }
    store_plane();
    w.cinfo[slot] = ci | kCiSolved;
    const bool spin_now = av.x != 0.0f || av.y != 0.0f || av.z != 0.0f;
    const uint32_t f = spin_now ? (f0 | kSpin) : (f0 & ~kSpin);
    if (f != f0) w.flags[slot] = f;
}

template <bool BASIS>
__global__ void __launch_bounds__(64) k_contact_boxes(WorldView w, GroundParams g)
{
    const uint32_t n_list = g.box_count[0];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_list; i += gridDim.x * blockDim.x) {
        const uint32_t slot = g.box_list[i];
        if (slot < g.n_slots) contact_body<BASIS>(w, g, slot);
    }
    // the workgroup that finishes last empties the list for the next sub-step's k_ground_select (every workgroup has read the count)
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(&g.box_count[1], 1u) == gridDim.x - 1u) {
            g.box_count[0] = 0;
            g.box_count[1] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Dynamic boxes against each other: the pair cache, simulation islands, one solver thread per island (oracle/island_ref.h and
// oracle/physics_ref.h CollideDynamicPairs / StepIsland, operation for operation).  Sub-step order:
//   k_island_begin      teleport rule (what k_ground_select does otherwise), the AABB Bullet feeds its broadphase for every body,
//                       per-slot scratch reset
//   (Broadphase::run on those AABBs; the host reads the number of pairs back)
//   k_island_pair_keys  the pairs of two Dynamic boxes as keys lower entity << 32 | higher entity    (sorted by hipcub)
//   k_island_carry      a pair's manifold from last sub-step's sorted pair list (binary search), or a fresh one
//   k_island_narrow     btBoxBoxDetector + the persistent manifold for every pair with an active body
//   k_island_union / k_island_members   union-find over the pairs (findUnions unites every pair of the cache); the bodies that are
//                       in a pair, keyed root slot << 32 | entity, and whether their island holds an ACTIVE_TAG body   (sorted by hipcub)
//   k_island_flags      bodies of islands that stay awake get kCiIsland; k_island_own collides their own pairs (plane, obstacles)
//   k_island_solve      one thread per island (iteration state in LDS where it fits; islands of 5 .. 16 bodies in a second launch);
//   k_island_solve_big  a workgroup per island of more than IslandParams::big_points contact points, Bullet's row order kept by levels
// then k_ground_select / k_ground / k_contact_boxes for the one-body islands and k_tick for everybody, as always.
struct IslBody {
    F3 dLin, dAng, push, turn, linVel, angVel, extForce, extTorque;
    float invMass;
    float invI[9];
    F3 origin;
    uint32_t slot, woken, pad;
};
static_assert(sizeof(IslBody) == kIslBodyBytes, "IslandParams::solver_bodies");
struct IslRow {
    F3 normal, relposCrossN, angularComp, relpos2CrossN, angularCompB;
    float jacDiagABInv, rhs, rhsPenetration, cfm, lower, upper, friction, applied, appliedPush;
    uint32_t a, b;       // positions in the sorted body list; b = kNone: the fixed solver body
    float* out;          // the manifold point's appliedImpulse (contact rows only)
    uint32_t lateral_at; // ... and how many floats behind it appliedImpulseLateral1 is
    float invMassA, invMassB; // the two bodies' inverse masses (B's 0 without a second body): a resolve out of LDS state then needs no load
                              // of its own — one issued behind the next row's would have to wait for that one first (loads return in order)
    uint32_t pad;
};
static_assert(sizeof(IslRow) == kIslRowBytes, "IslandParams::rows");

__device__ __forceinline__ uint32_t isl_find(uint32_t* parent, uint32_t s)
{
    while (true) {
        const uint32_t p = __hip_atomic_load(&parent[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == s) return s;
        s = p;
    }
}

template <bool BASIS>
__global__ void __launch_bounds__(256) k_island_begin(WorldView w, GroundParams g, IslandParams ip)
{
    const uint64_t slot64 = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (slot64 >= ip.n_slots) return;
    const uint32_t slot = static_cast<uint32_t>(slot64);
    ip.parent[slot] = slot;
    ip.member[slot] = 0u;
    ip.active[slot] = 0u;
    ip.index_of_slot[slot] = kNone;
    uint32_t f = w.flags[slot];
    const uint32_t type = f & kTypeMask;
    if (type == 0u) return;
    if (ip.repose && (f & kValid) && (f & (kTDirty | kBDirty))) {
        // (k_ground_select's re-pose, word for word: launch_ground is told not to do it again)
        const uint32_t f_in = f;
        const Q4 q = bt_quat_from_transform_euler(ld3(w.euler, slot));
        st4(w.quat, slot, q);
        f &= ~kSettled;
        const F3 zero{0.0f, 0.0f, 0.0f};
        if (type == 2u) st3(w.vel, slot, zero);
        if (f & kSpin) {
            st3(w.angvel, slot, zero);
            f &= ~kSpin;
        }
        if (type == 2u) st3(w.euler, slot, bt_transform_euler_from_mat(bt_mat_from_quat(q)));
        if (f != f_in) w.flags[slot] = f;
    }
    const uint32_t ci0 = w.cinfo[slot];
    uint32_t ci = ci0 & ~kCiIsland;
    if (ip.repose) {
        // applyGravity, once per stepSimulation call: a body that sleeps now gets none until the call ends, whatever wakes it later
        const bool sleeping = type == 2u && (f & kDrowsy) && w.deact[slot] == kDeactSleeping;
        ci = sleeping ? (ci | kCiNoGravity) : (ci & ~kCiNoGravity);
    }
    if (ci != ci0) w.cinfo[slot] = ci;
    // predictUnconstraintMotion / updateAabbs: the box of the pose united with the box of the predicted pose (k_tick's AABB block)
    const F3 pos = ld3(w.pos, slot);
    const Q4 q = ld4(w.quat, slot);
    const M3 basis = bt_mat_from_quat(q);
    const F3 he = ld3(w.half_extent, slot);
    float mn[3], mx[3];
    bt_aabb_of_pose(pos, basis, he, mn, mx);
    if (type == 2u) {
        const bool spin = (f & kSpin) != 0;
        const F3 v = ld3(w.vel, slot);
        const F3 av = spin ? ld3(w.angvel, slot) : F3{0.0f, 0.0f, 0.0f};
        const F3 pp{pos.x + v.x * g.dt, pos.y + v.y * g.dt, pos.z + v.z * g.dt};
        float mn2[3], mx2[3];
        if (BASIS || spin) {
            const M3 r2 = bt_mat_from_quat(bt_integrate_orientation(BASIS ? bt_quat_from_mat(basis) : q, av, g.dt));
            bt_aabb_of_pose(pp, r2, he, mn2, mx2);
        } else {
            bt_aabb_of_pose(pp, basis, he, mn2, mx2);
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            mn[a] = mn2[a] < mn[a] ? mn2[a] : mn[a];
            mx[a] = mx2[a] > mx[a] ? mx2[a] : mx[a];
        }
    }
    float* bb = w.aabb + 6ull * slot;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        bb[a] = mn[a];
        bb[3 + a] = mx[a];
    }
}

__device__ __forceinline__ bool isl_dynamic_box(const WorldView& w, uint32_t slot)
{
    return (w.flags[slot] & kTypeMask) == 2u && !(w.cinfo[slot] & kCiCapsule);
}

__global__ void __launch_bounds__(256) k_island_pair_keys(WorldView w, IslandParams ip)
{
    // workgroup b walks slice b % shards, interleaved with the other workgroups of that slice
    const uint32_t shard = blockIdx.x % ip.bp_shards, part = blockIdx.x / ip.bp_shards, parts = gridDim.x / ip.bp_shards;
    const unsigned long long found = ip.bp_counts[8u * shard];
    if (found > ip.bp_shard_cap && threadIdx.x == 0 && part == 0) atomicOr(&ip.counts[3], 2u); // the broadphase dropped pairs
    const uint32_t n = static_cast<uint32_t>(found < ip.bp_shard_cap ? found : ip.bp_shard_cap);
    const uint2* slice = ip.bp_stage + static_cast<uint64_t>(shard) * ip.bp_shard_cap;
    for (uint32_t i = part * blockDim.x + threadIdx.x; i < n; i += parts * blockDim.x) {
        uint2 pr = slice[i];
        uint32_t ea, eb;
        if (ip.bp_ids_are_entities) {
            ea = pr.x;
            eb = pr.y;
            pr.x = ip.slot_of_entity[ea];
            pr.y = ip.slot_of_entity[eb];
        }
        if (pr.x >= ip.n_slots || pr.y >= ip.n_slots) continue;
        if (!isl_dynamic_box(w, pr.x) || !isl_dynamic_box(w, pr.y)) continue;
        if (!ip.bp_ids_are_entities) {
            ea = ip.entity_of_slot[pr.x];
            eb = ip.entity_of_slot[pr.y];
        }
        const uint64_t key = ea < eb ? (static_cast<uint64_t>(ea) << 32) | eb : (static_cast<uint64_t>(eb) << 32) | ea;
        const uint32_t at = atomicAdd(&ip.counts[0], 1u);
        if (at < ip.pair_cap) ip.keys_raw[at] = key;
    }
}

__global__ void __launch_bounds__(256) k_island_carry(IslandParams ip)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ip.n_pairs) return;
    const uint64_t key = ip.keys[i];
    const uint32_t ga = ip.gen_of_entity[static_cast<uint32_t>(key >> 32)], gb = ip.gen_of_entity[static_cast<uint32_t>(key)];
    uint32_t lo = 0, hi = ip.n_prev;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ip.prev_keys[mid] < key) lo = mid + 1;
        else hi = mid;
    }
    uint32_t* m = ip.man + static_cast<uint64_t>(i) * kBoxManifoldWords;
    const uint32_t* old = ip.prev_man + static_cast<uint64_t>(lo) * kBoxManifoldWords;
    if (lo < ip.n_prev && ip.prev_keys[lo] == key && old[1] == ga && old[2] == gb) {
        for (uint32_t k = 0; k < kBoxManifoldWords; ++k) m[k] = old[k];
    } else {
        m[0] = 0u;
        m[1] = ga;
        m[2] = gb;
        for (uint32_t k = 3; k < kBoxManifoldWords; ++k) m[k] = 0u;
    }
}

__device__ __forceinline__ bool isl_sleeping(const WorldView& w, uint32_t slot)
{
    return (w.flags[slot] & kDrowsy) && w.deact[slot] == kDeactSleeping;
}

__global__ void __launch_bounds__(64) k_island_narrow(WorldView w, IslandParams ip)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ip.n_pairs) return;
    const uint64_t key = ip.keys[i];
    const uint32_t sa = ip.slot_of_entity[static_cast<uint32_t>(key >> 32)], sb = ip.slot_of_entity[static_cast<uint32_t>(key)];
    // btCollisionDispatcher::needsCollision: not when neither body is active (WANTS_DEACTIVATION counts as active)
    if (isl_sleeping(w, sa) && isl_sleeping(w, sb)) return;
    const float4 ca = w.cshape[sa], cb = w.cshape[sb];
    CtShape shape_a, shape_b;
    shape_a.capsule = shape_b.capsule = false;
    shape_a.dims = F3{ca.x, ca.y, ca.z};
    shape_b.dims = F3{cb.x, cb.y, cb.z};
    const F3 pos_a = ld3(w.pos, sa), pos_b = ld3(w.pos, sb);
    const M3 basis_a = bt_mat_from_quat(ld4(w.quat, sa)), basis_b = bt_mat_from_quat(ld4(w.quat, sb));
    ObstacleRec o;
    o.origin[0] = pos_b.x; o.origin[1] = pos_b.y; o.origin[2] = pos_b.z;
    o.half[0] = cb.x; o.half[1] = cb.y; o.half[2] = cb.z;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) o.basis[3 * r + c] = basis_b.m[r][c];
    }
    uint32_t* m = ip.man + static_cast<uint64_t>(i) * kBoxManifoldWords;
    const float breaking = fminf(ct_breaking_threshold(shape_a), ct_breaking_threshold(shape_b)); // btCollisionDispatcher::getNewManifold
    m[0] = static_cast<uint32_t>(bp_collide(reinterpret_cast<float*>(m + 4), static_cast<int>(m[0]), breaking, pos_a, basis_a, shape_a.dims, o));
}

// lock-free union by index: the larger root goes under the smaller one, so an island's root is its lowest slot
__global__ void __launch_bounds__(256) k_island_union(IslandParams ip)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ip.n_pairs) return;
    const uint64_t key = ip.keys[i];
    uint32_t a = ip.slot_of_entity[static_cast<uint32_t>(key >> 32)], b = ip.slot_of_entity[static_cast<uint32_t>(key)];
    while (true) {
        a = isl_find(ip.parent, a);
        b = isl_find(ip.parent, b);
        if (a == b) break;
        if (a < b) {
            const uint32_t t = a;
            a = b;
            b = t;
        }
        if (atomicCAS(&ip.parent[a], a, b) == a) break;
    }
}

__device__ __forceinline__ void isl_list_body(const WorldView& w, const IslandParams& ip, uint32_t s)
{
    if (atomicExch(&ip.member[s], 1u) != 0u) return;
    const uint32_t root = isl_find(ip.parent, s);
    const uint32_t at = atomicAdd(&ip.counts[1], 1u);
    if (at < ip.body_cap) {
        ip.body_keys_raw[at] = (static_cast<uint64_t>(root) << 32) | ip.entity_of_slot[s];
        ip.body_slot_raw[at] = s;
    }
    // buildIslands: "all sleeping" unless a body is ACTIVE_TAG (or DISABLE_DEACTIVATION: such a world keeps no records at all)
    const uint32_t f = w.flags[s];
    const uint32_t dz = (f & kDrowsy) ? w.deact[s] : 0u;
    if (dz != kDeactSleeping && dz != kDeactWants) atomicOr(&ip.active[root], 1u);
}

__global__ void __launch_bounds__(256) k_island_members(WorldView w, IslandParams ip)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ip.n_pairs) return;
    const uint64_t key = ip.keys[i];
    isl_list_body(w, ip, ip.slot_of_entity[static_cast<uint32_t>(key >> 32)]);
    isl_list_body(w, ip, ip.slot_of_entity[static_cast<uint32_t>(key)]);
}

// a body that slept when this stepSimulation call applied gravity, was woken since and is in no pair any more: an island of its own
// on this path (its gravity is off until the call ends, which only the island solver knows how to do)
__global__ void __launch_bounds__(256) k_island_orphans(WorldView w, IslandParams ip)
{
    const uint64_t slot64 = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (slot64 >= ip.n_slots) return;
    const uint32_t s = static_cast<uint32_t>(slot64);
    if ((w.flags[s] & kTypeMask) != 2u || !(w.cinfo[s] & kCiNoGravity)) return;
    if (isl_sleeping(w, s)) return;
    isl_list_body(w, ip, s);
}

__global__ void __launch_bounds__(256) k_island_flags(WorldView w, IslandParams ip)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ip.n_bodies) return;
    const uint32_t s = ip.body_slot[i];
    ip.index_of_slot[s] = i;
    if (ip.active[static_cast<uint32_t>(ip.body_keys[i] >> 32)]) w.cinfo[s] |= kCiIsland;
}

template <bool BASIS>
__global__ void __launch_bounds__(64) k_island_own(WorldView w, GroundParams g, IslandParams ip)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ip.n_bodies) return;
    const uint32_t s = ip.body_slot[i];
    if (w.cinfo[s] & kCiIsland) contact_body<BASIS>(w, g, s, true);
}

__device__ __forceinline__ float isl_dpps(const F3& u, const F3& v) { return (u.x * v.x + u.y * v.y) + u.z * v.z; }
__device__ __forceinline__ float isl_dot3s(const F3& u, const F3& v) { return u.x * v.x + (u.y * v.y + u.z * v.z); }
__device__ __forceinline__ F3 neg3(const F3& a) { return F3{-a.x, -a.y, -a.z}; }

// oracle/island_ref.h isl::ResolveRow2
__device__ void isl_resolve_row(IslBody* sb, IslRow& c, bool withUpperLimit)
{
    IslBody& a = sb[c.a];
    const bool two = c.b != kNone;
    float deltaImpulse = c.rhs - c.applied * c.cfm;
    const float dv1 = isl_dpps(c.relposCrossN, a.dAng) + isl_dpps(c.normal, a.dLin);
    const float dv2 = two ? isl_dpps(neg3(c.normal), sb[c.b].dLin) + isl_dpps(c.relpos2CrossN, sb[c.b].dAng) : 0.0f + 0.0f;
    deltaImpulse = __builtin_fmaf(-dv1, c.jacDiagABInv, deltaImpulse);
    deltaImpulse = __builtin_fmaf(-dv2, c.jacDiagABInv, deltaImpulse);
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
    a.dLin = F3{__builtin_fmaf(c.normal.x * a.invMass, deltaImpulse, a.dLin.x), __builtin_fmaf(c.normal.y * a.invMass, deltaImpulse, a.dLin.y),
                __builtin_fmaf(c.normal.z * a.invMass, deltaImpulse, a.dLin.z)};
    a.dAng = F3{__builtin_fmaf(c.angularComp.x, deltaImpulse, a.dAng.x), __builtin_fmaf(c.angularComp.y, deltaImpulse, a.dAng.y),
                __builtin_fmaf(c.angularComp.z, deltaImpulse, a.dAng.z)};
    if (two) {
        IslBody& b = sb[c.b];
        b.dLin = F3{__builtin_fmaf(-c.normal.x * b.invMass, deltaImpulse, b.dLin.x), __builtin_fmaf(-c.normal.y * b.invMass, deltaImpulse, b.dLin.y),
                    __builtin_fmaf(-c.normal.z * b.invMass, deltaImpulse, b.dLin.z)};
        b.dAng = F3{__builtin_fmaf(c.angularCompB.x, deltaImpulse, b.dAng.x), __builtin_fmaf(c.angularCompB.y, deltaImpulse, b.dAng.y),
                    __builtin_fmaf(c.angularCompB.z, deltaImpulse, b.dAng.z)};
    }
}

// oracle/island_ref.h isl::ResolveSplitPenetration2
__device__ void isl_resolve_split(IslBody* sb, IslRow& c)
{
    if (!c.rhsPenetration) return;
    IslBody& a = sb[c.a];
    const bool two = c.b != kNone;
    float deltaImpulse = c.rhsPenetration - c.appliedPush * c.cfm;
    const float dv1 = isl_dot3s(c.normal, a.push) + isl_dot3s(c.relposCrossN, a.turn);
    const float dv2 = two ? isl_dot3s(neg3(c.normal), sb[c.b].push) + isl_dot3s(c.relpos2CrossN, sb[c.b].turn) : 0.0f + 0.0f;
    deltaImpulse = deltaImpulse - dv1 * c.jacDiagABInv;
    deltaImpulse = deltaImpulse - dv2 * c.jacDiagABInv;
    const float sum = c.appliedPush + deltaImpulse;
    if (sum < c.lower) {
        deltaImpulse = c.lower - c.appliedPush;
        c.appliedPush = c.lower;
    } else {
        c.appliedPush = sum;
    }
    const F3 lin = F3{c.normal.x * a.invMass, c.normal.y * a.invMass, c.normal.z * a.invMass};
    a.push = add3(a.push, scale3(lin, deltaImpulse));
    a.turn = add3(a.turn, scale3(c.angularComp, deltaImpulse));
    if (two) {
        IslBody& b = sb[c.b];
        const F3 lin2 = F3{-c.normal.x * b.invMass, -c.normal.y * b.invMass, -c.normal.z * b.invMass};
        b.push = add3(b.push, scale3(lin2, deltaImpulse));
        b.turn = add3(b.turn, scale3(c.angularCompB, deltaImpulse));
    }
}

__device__ __forceinline__ M3 isl_inv_i(const IslBody& b)
{
    M3 m;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) m.m[r][c] = b.invI[3 * r + c];
    }
    return m;
}

// One contact's two rows, warm started: oracle/island_ref.h SolveIsland's loop body (ct_add_contact with a second body)
template <bool WARM = true>
__device__ void isl_add_contact(IslBody* sb, IslRow& c, IslRow& fr, uint32_t ia, uint32_t ib, float invTimeStep, const F3& worldA, const F3& worldB,
                                const F3& n, float distance, float friction, float combinedRestitution, float* out, uint32_t lateral_at)
{
    constexpr float kErp2 = 0.2f, kSplitThreshold = -0.04f, kWarmstart = 0.85f, kSor = 1.0f, kRestitutionVelocityThreshold = 0.2f;
    IslBody& A = sb[ia];
    const bool two = ib != kNone;
    const F3 zero{0.0f, 0.0f, 0.0f};
    c.a = fr.a = ia;
    c.b = fr.b = ib;
    c.out = out;
    c.lateral_at = lateral_at;
    fr.out = nullptr;
    fr.lateral_at = 0u;
    c.invMassA = fr.invMassA = A.invMass;
    c.invMassB = fr.invMassB = two ? sb[ib].invMass : 0.0f;
    c.pad = fr.pad = 0u;
    const M3 invIA = isl_inv_i(A);
    const F3 rel_pos1 = sub3(worldA, A.origin);
    const F3 vel1 = add3(add3(A.linVel, A.extForce), cross3(add3(A.angVel, A.extTorque), rel_pos1));
    F3 rel_pos2 = zero, vel2 = zero;
    if (two) {
        const IslBody& B = sb[ib];
        rel_pos2 = sub3(worldB, B.origin);
        vel2 = add3(add3(B.linVel, B.extForce), cross3(add3(B.angVel, B.extTorque), rel_pos2));
    }
    const F3 vel = sub3(vel1, vel2);
    const float rel_vel = dot3(n, vel);
    const float relaxation = kSor;
    const F3 torqueAxis0 = cross3(rel_pos1, n);
    c.angularComp = mat_vec(invIA, torqueAxis0);
    F3 torqueAxis1 = zero;
    c.angularCompB = zero;
    {
        const F3 vec = cross3(c.angularComp, rel_pos1);
        const float denom0 = inv_mass_plus_dot(A.invMass, n, vec);
        float denom1 = 0.0f;
        if (two) {
            const IslBody& B = sb[ib];
            torqueAxis1 = cross3(n, rel_pos2);
            c.angularCompB = mat_vec(isl_inv_i(B), torqueAxis1);
            denom1 = inv_mass_plus_dot(B.invMass, n, cross3(rel_pos2, c.angularCompB));
        }
        const float cfm0 = 0.0f * invTimeStep;
        c.jacDiagABInv = relaxation / (denom0 + denom1 + cfm0);
    }
    c.normal = n;
    c.relposCrossN = torqueAxis0;
    c.relpos2CrossN = torqueAxis1;
    const float penetration = distance + 0.0f;
    c.friction = friction;
    float restitution = 0.0f;
    if (combinedRestitution != 0.0f) {
        const F3 rbVel1 = add3(A.linVel, cross3(A.angVel, rel_pos1));
        F3 rbVel2 = zero;
        if (two) rbVel2 = add3(sb[ib].linVel, cross3(sb[ib].angVel, rel_pos2));
        const float rbRelVel = dot3(n, sub3(rbVel1, rbVel2));
        restitution = __builtin_fabsf(rbRelVel) < kRestitutionVelocityThreshold ? 0.0f : combinedRestitution * -rbRelVel;
        if (restitution <= 0.0f) restitution = 0.0f;
    }
    c.applied = *out * kWarmstart;
    if (WARM) { // (k_island_solve_big applies the warm start level by level: isl_warm_start)
        const F3 lin = F3{c.normal.x * A.invMass, c.normal.y * A.invMass, c.normal.z * A.invMass};
        A.dLin = add3(A.dLin, scale3(lin, c.applied));
        A.dAng = add3(A.dAng, scale3(c.angularComp, c.applied * 1.0f));
        if (two) {
            IslBody& B = sb[ib];
            const F3 linB = F3{B.invMass * n.x, B.invMass * n.y, B.invMass * n.z};
            B.dLin = sub3(B.dLin, scale3(linB, c.applied));
            B.dAng = add3(B.dAng, scale3(c.angularCompB, c.applied * 1.0f));
        }
    }
    c.appliedPush = 0.0f;
    {
        const float vel1Dotn = dot_xzy(c.normal, add3(A.linVel, A.extForce)) + dot_xzy(c.relposCrossN, add3(A.angVel, A.extTorque));
        float vel2Dotn = 0.0f + 0.0f;
        if (two) {
            const IslBody& B = sb[ib];
            const F3 l = add3(B.linVel, B.extForce);
            vel2Dotn = dot_xzy(c.relpos2CrossN, add3(B.angVel, B.extTorque)) + ((-(l.x * n.x) - l.z * n.z) - l.y * n.y);
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
    F3 dir = sub3(vel, scale3(n, rel_vel));
    const float lat_rel_vel = dot3(dir, dir);
    if (lat_rel_vel > kBtEpsilon) {
        dir = scale3(dir, 1.0f / __builtin_sqrtf(lat_rel_vel));
    } else {
        dir = ct_plane_space1(n);
    }
    fr.friction = friction;
    fr.normal = dir;
    fr.relposCrossN = cross3(rel_pos1, dir);
    fr.angularComp = mat_vec(invIA, fr.relposCrossN);
    fr.relpos2CrossN = zero;
    fr.angularCompB = zero;
    {
        const F3 vec = cross3(fr.angularComp, rel_pos1);
        const float denom0 = inv_mass_plus_dot(A.invMass, dir, vec);
        float denom1 = 0.0f;
        if (two) {
            const IslBody& B = sb[ib];
            fr.relpos2CrossN = cross3(dir, rel_pos2);
            fr.angularCompB = mat_vec(isl_inv_i(B), fr.relpos2CrossN);
            denom1 = inv_mass_plus_dot(B.invMass, dir, cross3(rel_pos2, fr.angularCompB));
        }
        fr.jacDiagABInv = relaxation / (denom0 + denom1);
    }
    {
        const float vel1Dotn = dot_xzy(fr.normal, add3(A.linVel, A.extForce)) + dot_xzy(fr.relposCrossN, A.angVel);
        float rv;
        if (two) {
            const IslBody& B = sb[ib];
            const F3 l = add3(B.linVel, B.extForce);
            rv = dot_xzy(fr.relpos2CrossN, B.angVel) + ((vel1Dotn - l.z * dir.z) + (-(l.x * dir.x) - l.y * dir.y));
        } else {
            const float vel2Dotn = 0.0f + 0.0f;
            rv = vel1Dotn + vel2Dotn;
        }
        const float velocityError = 0.0f - rv;
        const float velocityImpulse = velocityError * fr.jacDiagABInv;
        fr.rhs = 0.0f + velocityImpulse;
        fr.rhsPenetration = 0.0f;
        fr.cfm = 0.0f;
        fr.lower = -fr.friction;
        fr.upper = fr.friction;
    }
    fr.applied = 0.0f;
    fr.appliedPush = 0.0f;
}

// ---- the iterations of a small island out of LDS.  In global memory every row update is a store that the next row's load has to wait
//      for (a body's delta velocities, a row's applied impulse): ~7 us per row, 1.9 ms for an island of two boxes on eight points.
//      What the iterations CHANGE — four vectors per body, three scalars per contact point — lives in a lane-private LDS column
//      (word k of lane l at [64 k + l]: no bank conflicts); what they only read stays in the rows in global memory.
constexpr uint32_t kIslLdsBodies = 4, kIslLdsPoints = 16, kIslLdsWords = kIslLdsBodies * 12u + kIslLdsPoints * 3u;
constexpr uint32_t kIslMidBodies = 16; // k_island_solve<.., true>: that many bodies' delta velocities in a lane's LDS column (49 KB per workgroup)
template <uint32_t STRIDE>
struct IslLocalT {
    float* p;       // this lane's column (STRIDE 64), or the workgroup's block (STRIDE 1: k_island_solve_big)
    uint32_t first; // the island's first body in the sorted list
    __device__ __forceinline__ F3 get(uint32_t body, uint32_t field) const
    {
        const float* q = p + ((body - first) * 12u + field * 3u) * STRIDE;
        return F3{q[0], q[STRIDE], q[2u * STRIDE]};
    }
    __device__ __forceinline__ void set(uint32_t body, uint32_t field, const F3& v) const
    {
        float* q = p + ((body - first) * 12u + field * 3u) * STRIDE;
        q[0] = v.x;
        q[STRIDE] = v.y;
        q[2u * STRIDE] = v.z;
    }
    // k: 0 the contact row's applied impulse, 1 its applied push impulse, 2 the friction row's applied impulse
    __device__ __forceinline__ float& row(uint32_t r, uint32_t k) const { return p[(kIslLdsBodies * 12u + r * 3u + k) * STRIDE]; }
};
using IslLocal = IslLocalT<64u>;

// isl_resolve_row on that state (fields 0 dLin, 1 dAng)
template <class Local>
__device__ __forceinline__ void isl_resolve_row_lds(const Local& L, const IslBody* sb, const IslRow& c, float& applied, float lower, float upper, bool withUpperLimit)
{
    const bool two = c.b != kNone;
    const float invMassA = c.invMassA;
    F3 aLin = L.get(c.a, 0), aAng = L.get(c.a, 1);
    float deltaImpulse = c.rhs - applied * c.cfm;
    const float dv1 = isl_dpps(c.relposCrossN, aAng) + isl_dpps(c.normal, aLin);
    F3 bLin{0.0f, 0.0f, 0.0f}, bAng{0.0f, 0.0f, 0.0f};
    float invMassB = 0.0f;
    if (two) {
        bLin = L.get(c.b, 0);
        bAng = L.get(c.b, 1);
        invMassB = c.invMassB;
    }
    const float dv2 = two ? isl_dpps(neg3(c.normal), bLin) + isl_dpps(c.relpos2CrossN, bAng) : 0.0f + 0.0f;
    deltaImpulse = __builtin_fmaf(-dv1, c.jacDiagABInv, deltaImpulse);
    deltaImpulse = __builtin_fmaf(-dv2, c.jacDiagABInv, deltaImpulse);
    const float sum = applied + deltaImpulse;
    if (lower < sum) {
        if (withUpperLimit && !(sum < upper)) {
            deltaImpulse = upper - applied;
            applied = upper;
        } else {
            applied = sum;
        }
    } else {
        deltaImpulse = lower - applied;
        applied = lower;
    }
    L.set(c.a, 0, F3{__builtin_fmaf(c.normal.x * invMassA, deltaImpulse, aLin.x), __builtin_fmaf(c.normal.y * invMassA, deltaImpulse, aLin.y),
                     __builtin_fmaf(c.normal.z * invMassA, deltaImpulse, aLin.z)});
    L.set(c.a, 1, F3{__builtin_fmaf(c.angularComp.x, deltaImpulse, aAng.x), __builtin_fmaf(c.angularComp.y, deltaImpulse, aAng.y),
                     __builtin_fmaf(c.angularComp.z, deltaImpulse, aAng.z)});
    if (two) {
        L.set(c.b, 0, F3{__builtin_fmaf(-c.normal.x * invMassB, deltaImpulse, bLin.x), __builtin_fmaf(-c.normal.y * invMassB, deltaImpulse, bLin.y),
                         __builtin_fmaf(-c.normal.z * invMassB, deltaImpulse, bLin.z)});
        L.set(c.b, 1, F3{__builtin_fmaf(c.angularCompB.x, deltaImpulse, bAng.x), __builtin_fmaf(c.angularCompB.y, deltaImpulse, bAng.y),
                         __builtin_fmaf(c.angularCompB.z, deltaImpulse, bAng.z)});
    }
}

// isl_resolve_split on that state (fields 2 push, 3 turn)
template <class Local>
__device__ __forceinline__ void isl_resolve_split_lds(const Local& L, const IslBody* sb, const IslRow& c, float& appliedPush)
{
    if (!c.rhsPenetration) return;
    const bool two = c.b != kNone;
    const float invMassA = c.invMassA;
    const F3 aPush = L.get(c.a, 2), aTurn = L.get(c.a, 3);
    float deltaImpulse = c.rhsPenetration - appliedPush * c.cfm;
    const float dv1 = isl_dot3s(c.normal, aPush) + isl_dot3s(c.relposCrossN, aTurn);
    F3 bPush{0.0f, 0.0f, 0.0f}, bTurn{0.0f, 0.0f, 0.0f};
    float invMassB = 0.0f;
    if (two) {
        bPush = L.get(c.b, 2);
        bTurn = L.get(c.b, 3);
        invMassB = c.invMassB;
    }
    const float dv2 = two ? isl_dot3s(neg3(c.normal), bPush) + isl_dot3s(c.relpos2CrossN, bTurn) : 0.0f + 0.0f;
    deltaImpulse = deltaImpulse - dv1 * c.jacDiagABInv;
    deltaImpulse = deltaImpulse - dv2 * c.jacDiagABInv;
    const float sum = appliedPush + deltaImpulse;
    if (sum < c.lower) {
        deltaImpulse = c.lower - appliedPush;
        appliedPush = c.lower;
    } else {
        appliedPush = sum;
    }
    const F3 lin = F3{c.normal.x * invMassA, c.normal.y * invMassA, c.normal.z * invMassA};
    L.set(c.a, 2, add3(aPush, scale3(lin, deltaImpulse)));
    L.set(c.a, 3, add3(aTurn, scale3(c.angularComp, deltaImpulse)));
    if (two) {
        const F3 lin2 = F3{-c.normal.x * invMassB, -c.normal.y * invMassB, -c.normal.z * invMassB};
        L.set(c.b, 2, add3(bPush, scale3(lin2, deltaImpulse)));
        L.set(c.b, 3, add3(bTurn, scale3(c.angularCompB, deltaImpulse)));
    }
}

// the number of the obstacle that is entity `entity` (the list ascends), or kNone
__device__ __forceinline__ uint32_t isl_obstacle_of(const GroundParams& g, uint32_t entity)
{
    uint32_t lo = 0, hi = g.n_obstacles;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (g.obstacles[mid].entity < entity) lo = mid + 1;
        else hi = mid;
    }
    return lo < g.n_obstacles && g.obstacles[lo].entity == entity ? lo : kNone;
}

// ---- the pieces of an island's solve, per body (k_island_solve: one thread walks them; k_island_solve_big: a workgroup shares them out)
// convertBodies for body i of the sorted list (and, for a body woken just now, the pairs with obstacles that ended while it slept);
// returns the contact points of its own manifolds (plane, obstacles)
template <bool BASIS>
__device__ uint32_t isl_prepare_body(const WorldView& w, const GroundParams& g, const IslandParams& ip, IslBody* sb, uint32_t i)
{
    uint32_t own = 0;
        const uint32_t slot = ip.body_slot[i];
        const uint32_t f0 = w.flags[slot];
        const uint32_t ci = w.cinfo[slot];
        const bool woken = (f0 & kDrowsy) && w.deact[slot] == kDeactSleeping;
        const bool no_gravity = (ci & kCiNoGravity) != 0; // (asleep when this call applied gravity: k_island_begin)
        const uint32_t cls = f0 >> kMassShift;
        float inv_mass;
        F3 force;
        if (cls != kMassClassArray) {
            const float4 gf = w.grav_palette[cls];
            inv_mass = gf.w;
            force = F3{gf.x, gf.y, gf.z};
        } else {
            inv_mass = w.inv_mass[slot];
            force = F3{g.gx / inv_mass, g.gy / inv_mass, g.gz / inv_mass};
        }
        if (no_gravity) force = F3{0.0f, 0.0f, 0.0f};
        const float4 cs = w.cshape[slot];
        CtShape shape;
        shape.capsule = false;
        shape.dims = F3{cs.x, cs.y, cs.z};
        const F3 invInertiaLocal = ct_inv_inertia_local(ct_local_inertia(shape, w.cmass[slot]));
        const bool spin = (f0 & kSpin) != 0;
        const Q4 q = ld4(w.quat, slot);
        const M3 basis = bt_mat_from_quat(q);
        const Q4 orn = BASIS ? bt_quat_from_mat(basis) : q;
        const M3 invI = ct_inv_inertia_world(basis, invInertiaLocal);
        IslBody b;
        b.dLin = b.dAng = b.push = b.turn = F3{0.0f, 0.0f, 0.0f};
        b.linVel = ld3(w.vel, slot);
        b.angVel = spin ? ld3(w.angvel, slot) : F3{0.0f, 0.0f, 0.0f};
        b.invMass = inv_mass;
        b.extForce = scale3(scale3(force, inv_mass), g.dt);
        b.extTorque = F3{0.0f, 0.0f, 0.0f};
        b.extTorque = add3(b.extTorque, ct_gyroscopic_impulse(invInertiaLocal, b.angVel, orn, g.dt));
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) b.invI[3 * r + c] = invI.m[r][c];
        }
        b.origin = ld3(w.pos, slot);
        b.slot = slot;
        b.woken = woken ? 1u : 0u;
        b.pad = 0u;
        sb[i] = b;
        if (g.plane != 0u && (ci & kCiGroundMask)) own += (ci >> kCiCountShift) & 7u;
        if (ci & kCiBoxes) {
            uint32_t* rows = w.bmanifold + static_cast<uint64_t>(slot) * (kBoxManifolds * kBoxManifoldWords);
            for (uint32_t e = 0; e < kBoxManifolds; ++e) {
                uint32_t* hdr = rows + e * kBoxManifoldWords;
                if (hdr[0] == kBoxNone) continue;
                if (woken) {
                    // not collided this sub-step: its manifolds are what its last collision left, minus the pairs that ended while it
                    // slept (oracle/physics_ref.h StepIsland) — obstacle gone or re-created, filter, fed AABBs apart
                    const uint32_t k = isl_obstacle_of(g, hdr[0]);
                    bool keep = k != kNone;
                    if (keep) {
                        const ObstacleRec& o = g.obstacles[k];
                        const float* bb = w.aabb + 6ull * slot;
                        keep = o.live && o.generation == hdr[2] && (w.group[slot] & o.mask) != 0u && (o.group & w.mask[slot]) != 0u && bb[0] <= o.aabb[3] &&
                               bb[3] >= o.aabb[0] && bb[1] <= o.aabb[4] && bb[4] >= o.aabb[1] && bb[2] <= o.aabb[5] && bb[5] >= o.aabb[2];
                    }
                    if (!keep) {
                        hdr[0] = kBoxNone;
                        hdr[1] = 0u;
                        continue;
                    }
                }
                own += hdr[1];
            }
        }
    return own;
}

// first pair of the sorted pair list whose lower entity is `entity`
__device__ __forceinline__ uint32_t isl_first_pair_of(const IslandParams& ip, uint32_t entity)
{
    const uint64_t owner = static_cast<uint64_t>(entity) << 32;
    uint32_t lo = 0, hi = ip.n_pairs;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ip.keys[mid] < owner) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// contact points of the pairs the body owns (it is their lower entity)
__device__ uint32_t isl_pair_points(const IslandParams& ip, uint32_t slot)
{
    const uint32_t entity = ip.entity_of_slot[slot];
    uint32_t n = 0;
    for (uint32_t k = isl_first_pair_of(ip, entity); k < ip.n_pairs && static_cast<uint32_t>(ip.keys[k] >> 32) == entity; ++k) {
        n += ip.man[static_cast<uint64_t>(k) * kBoxManifoldWords];
    }
    return n;
}

// convertContacts for body i: its plane manifold, its manifolds with obstacles (ascending entity), its pairs with Dynamic boxes of higher
// entity (ascending) — rows j, j + 1, ... of the island; returns the row after its last
template <bool WARM>
__device__ uint32_t isl_build_body_rows(const WorldView& w, const GroundParams& g, const IslandParams& ip, IslBody* sb, uint32_t i, float invTimeStep,
                                        IslRow* normalRow, IslRow* frictionRow, uint32_t j)
{
        const uint32_t slot = sb[i].slot;
        const uint32_t ci = w.cinfo[slot];
        const M3 basis = bt_mat_from_quat(ld4(w.quat, slot));
        const F3 pos = sb[i].origin;
        const float bodyFriction = w.cfriction[slot], bodyRestitution = w.crestitution ? w.crestitution[slot] : 0.0f;
        if (g.plane != 0u && (ci & kCiGroundMask)) {
            const uint32_t n = (ci >> kCiCountShift) & 7u;
            const float combinedFriction = fmaxf(-10.0f, fminf(10.0f, bodyFriction * 1.0f));
            float* mp = w.manifold + 32ull * slot;
            for (uint32_t k = 0; k < n; ++k) {
                const F3 worldA = xform_point(basis, pos, F3{mp[8 * k], mp[8 * k + 1], mp[8 * k + 2]});
                isl_add_contact<WARM>(sb, normalRow[j], frictionRow[j], i, kNone, invTimeStep, worldA, F3{0.0f, 0.0f, 0.0f}, F3{0.0f, 1.0f, 0.0f}, mp[8 * k + 5],
                                combinedFriction, 0.0f, mp + 8 * k + 3, 4u);
                j++;
            }
        }
        if (ci & kCiBoxes) {
            uint32_t* rows = w.bmanifold + static_cast<uint64_t>(slot) * (kBoxManifolds * kBoxManifoldWords);
            uint32_t done = 0;
            for (uint32_t pass = 0; pass < kBoxManifolds; ++pass) { // (the rows are in no particular order: lowest entity first)
                uint32_t best = kBoxManifolds;
                for (uint32_t e = 0; e < kBoxManifolds; ++e) {
                    if ((done & (1u << e)) || rows[e * kBoxManifoldWords] == kBoxNone) continue;
                    if (best == kBoxManifolds || rows[e * kBoxManifoldWords] < rows[best * kBoxManifoldWords]) best = e;
                }
                if (best == kBoxManifolds) break;
                done |= 1u << best;
                uint32_t* hdr = rows + best * kBoxManifoldWords;
                const uint32_t at = isl_obstacle_of(g, hdr[0]); // (by its entity: the list may have been rebuilt since the body was last collided)
                // (at == kNone cannot happen — a collided body's partners are in the list, a woken body's were checked — and is reported, with
                //  the rows still built so that the sweeps stay inside the island)
                if (at == kNone) atomicOr(&ip.counts[3], 4u);
                const float combinedFriction = fmaxf(-10.0f, fminf(10.0f, bodyFriction * (at == kNone ? 0.0f : g.obstacles[at].friction)));
                const float combinedRestitution = bodyRestitution * (at == kNone ? 0.0f : g.obstacles[at].restitution);
                float* pts = reinterpret_cast<float*>(hdr + 4);
                for (uint32_t k = 0; k < hdr[1]; ++k) {
                    float* c = pts + 12 * k;
                    const F3 worldA = xform_point(basis, pos, bp_get3(c, 0));
                    isl_add_contact<WARM>(sb, normalRow[j], frictionRow[j], i, kNone, invTimeStep, worldA, F3{0.0f, 0.0f, 0.0f}, bp_get3(c, 6), c[9], combinedFriction,
                                    combinedRestitution, c + 10, 1u);
                    j++;
                }
            }
        }
        const uint64_t owner = static_cast<uint64_t>(ip.entity_of_slot[slot]) << 32;
        uint32_t lo = 0, hi = ip.n_pairs;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (ip.keys[mid] < owner) lo = mid + 1;
            else hi = mid;
        }
        for (uint32_t k = lo; k < ip.n_pairs && (ip.keys[k] >> 32) == (owner >> 32); ++k) {
            uint32_t* m = ip.man + static_cast<uint64_t>(k) * kBoxManifoldWords;
            const uint32_t other_slot = ip.slot_of_entity[static_cast<uint32_t>(ip.keys[k])];
            const uint32_t ib = ip.index_of_slot[other_slot];
            const M3 basis_b = bt_mat_from_quat(ld4(w.quat, other_slot));
            const float combinedFriction = fmaxf(-10.0f, fminf(10.0f, bodyFriction * w.cfriction[other_slot]));
            const float combinedRestitution = bodyRestitution * (w.crestitution ? w.crestitution[other_slot] : 0.0f);
            float* pts = reinterpret_cast<float*>(m + 4);
            for (uint32_t q = 0; q < m[0]; ++q) {
                float* c = pts + 12 * q;
                const F3 worldA = xform_point(basis, pos, bp_get3(c, 0));
                const F3 worldB = xform_point_b(basis_b, sb[ib].origin, bp_get3(c, 3));
                isl_add_contact<WARM>(sb, normalRow[j], frictionRow[j], i, ib, invTimeStep, worldA, worldB, bp_get3(c, 6), c[9], combinedFriction, combinedRestitution,
                                c + 10, 1u);
                j++;
            }
        }
    return j;
}

// solveGroupCacheFriendlyFinish for body i
template <bool BASIS>
__device__ void isl_finish_body(const WorldView& w, const GroundParams& g, IslBody* sb, uint32_t i)
{
    constexpr float kSplitTurnErp = 0.1f;
        IslBody& s = sb[i];
        const uint32_t slot = s.slot;
        s.linVel = add3(s.linVel, s.dLin);
        s.angVel = add3(s.angVel, s.dAng);
        uint32_t ci = w.cinfo[slot] | kCiSolved;
        if (s.push.x != 0.0f || s.push.y != 0.0f || s.push.z != 0.0f || s.turn.x != 0.0f || s.turn.y != 0.0f || s.turn.z != 0.0f) {
            const Q4 q = ld4(w.quat, slot);
            const Q4 orn = BASIS ? bt_quat_from_mat(bt_mat_from_quat(q)) : q;
            st3(w.pos, slot, add3(s.origin, scale3(s.push, g.dt)));
            st4(w.quat, slot, bt_integrate_orientation(orn, scale3(s.turn, kSplitTurnErp), g.dt));
            ci |= kCiMoved;
        }
        const F3 v = add3(s.linVel, s.extForce), av = add3(s.angVel, s.extTorque);
        st3(w.vel, slot, v);
        st3(w.angvel, slot, av);
        uint32_t f0 = w.flags[slot];
        uint32_t f = (av.x != 0.0f || av.y != 0.0f || av.z != 0.0f) ? (f0 | kSpin) : (f0 & ~kSpin);
        if (s.woken) {
            // buildIslands: a sleeping body of an island that has an active body -> WANTS_DEACTIVATION, timer 0
            w.deact[slot] = kDeactWants;
            f |= kDrowsy;
        }
        if (f != f0) w.flags[slot] = f;
        w.cinfo[slot] = ci;
}

// MID = false: the grid walks the sorted body list, an island's first body solves it — or hands it on: to the mid list (5 .. kIslMidBodies bodies:
// k_island_solve<.., true>, launched next, keeps their bodies' delta velocities in LDS) or to the big list (k_island_solve_big).
template <bool BASIS, bool MID>
__global__ void __launch_bounds__(64) k_island_solve(WorldView w, GroundParams g, IslandParams ip)
{
    __shared__ float s_isl[(MID ? kIslMidBodies * 12u : kIslLdsWords) * 64u];
    uint32_t first, end;
    if (MID) {
        const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
        if (t >= ip.counts[7]) return;
        first = ip.mid_list[2u * t];
        end = ip.mid_list[2u * t + 1u];
    } else {
        first = blockIdx.x * blockDim.x + threadIdx.x;
        if (first >= ip.n_bodies) return;
        const uint32_t root = static_cast<uint32_t>(ip.body_keys[first] >> 32);
        if (first > 0 && static_cast<uint32_t>(ip.body_keys[first - 1] >> 32) == root) return; // not the island's first body
        if (!ip.active[root]) return; // "all sleeping": k_tick turns its WANTS_DEACTIVATION bodies to ISLAND_SLEEPING, the others sleep already
        {
            // the island's bodies are the run of keys with this root: its end by bisection (a big island's head counted 1,828 keys one by
            // one here — 0.46 ms)
            uint32_t lo = first + 1, hi = ip.n_bodies;
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (static_cast<uint32_t>(ip.body_keys[mid] >> 32) <= root) lo = mid + 1;
                else hi = mid;
            }
            end = lo;
        }
        if (end - first > ip.big_points && end - first > kIslLdsBodies) { // (that many bodies: not worth counting)
            const uint32_t at = atomicAdd(&ip.counts[4], 1u);
            ip.big_list[2u * at] = first;
            ip.big_list[2u * at + 1u] = end;
            return;
        }
    }
    IslBody* sb = static_cast<IslBody*>(ip.solver_bodies);
    const int kIterations = static_cast<int>(ip.iterations);
    const float invTimeStep = 1.0f / g.dt;
    // ---- convertBodies, and how many rows the island needs
    uint32_t n_points = 0;
    for (uint32_t i = first; i < end; ++i) n_points += isl_prepare_body<BASIS>(w, g, ip, sb, i);
    // the pairs owned by the island's bodies (keys ascend with the owner's entity, so they are one run per body)
    for (uint32_t i = first; i < end; ++i) n_points += isl_pair_points(ip, ip.body_slot[i]);
    const bool small = end - first <= kIslLdsBodies && n_points <= kIslLdsPoints;
    if (!MID && !small) { // (whoever takes it prepares its bodies again: the same values)
        if (n_points > ip.big_points) {
            const uint32_t at = atomicAdd(&ip.counts[4], 1u);
            ip.big_list[2u * at] = first;
            ip.big_list[2u * at + 1u] = end;
            return;
        }
        if (end - first <= kIslMidBodies) {
            const uint32_t at = atomicAdd(&ip.counts[7], 1u);
            ip.mid_list[2u * at] = first;
            ip.mid_list[2u * at + 1u] = end;
            return;
        }
    }
    IslRow* rows_base = nullptr;
    if (n_points) {
        const uint32_t at = atomicAdd(&ip.counts[2], 2u * n_points);
        if (at + 2u * n_points > ip.row_cap) {
            atomicOr(&ip.counts[3], 1u); // (cannot happen: the host sizes the pool for every point the manifolds can hold)
            return;
        }
        rows_base = static_cast<IslRow*>(ip.rows) + at;
    }
    IslRow* normalRow = rows_base;
    IslRow* frictionRow = rows_base + n_points;
    // ---- convertContacts: body by body (ascending entity) its plane manifold, its manifolds with obstacles (ascending entity), its pairs
    //      with Dynamic boxes of higher entity (ascending)
    uint32_t j = 0;
    for (uint32_t i = first; i < end; ++i) j = isl_build_body_rows<true>(w, g, ip, sb, i, invTimeStep, normalRow, frictionRow, j);
    // ---- solveGroupCacheFriendlySplitImpulseIterations, solveGroupCacheFriendlyIterations
    if (!MID && small) {
        const IslLocal L{s_isl + (threadIdx.x & 63u), first};
        for (uint32_t i = first; i < end; ++i) {
            L.set(i, 0, sb[i].dLin); // (the warm start)
            L.set(i, 1, sb[i].dAng);
            L.set(i, 2, F3{0.0f, 0.0f, 0.0f});
            L.set(i, 3, F3{0.0f, 0.0f, 0.0f});
        }
        for (uint32_t r = 0; r < n_points; ++r) {
            L.row(r, 0) = normalRow[r].applied;
            L.row(r, 1) = 0.0f;
            L.row(r, 2) = 0.0f;
        }
        // (a row's constants are requested one row ahead: the load of row r + 1 is in flight while row r is resolved — what a sweep
        //  waits for is then the LDS round trip of the bodies it shares with the row before, not a global load per row)
        for (int it = 0; it < kIterations; ++it) {
            bool any = false;
            for (uint32_t r = 0; r < n_points; ++r) any = any || normalRow[r].rhsPenetration != 0.0f;
            if (!any) break; // (no row takes the split impulse: every sweep would return at its first test)
            IslRow cur = normalRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = normalRow[r + 1 < n_points ? r + 1 : r];
                isl_resolve_split_lds(L, sb, cur, L.row(r, 1));
                cur = nxt;
            }
        }
        for (int it = 0; it < kIterations; ++it) {
            if (n_points == 0) break;
            IslRow cur = normalRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = normalRow[r + 1 < n_points ? r + 1 : r];
                isl_resolve_row_lds(L, sb, cur, L.row(r, 0), cur.lower, cur.upper, false);
                cur = nxt;
            }
            cur = frictionRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = frictionRow[r + 1 < n_points ? r + 1 : r];
                const float totalImpulse = L.row(r, 0);
                if (totalImpulse > 0.0f) {
                    const float friction = cur.friction;
                    isl_resolve_row_lds(L, sb, cur, L.row(r, 2), -(friction * totalImpulse), friction * totalImpulse, true);
                }
                cur = nxt;
            }
        }
        for (uint32_t i = first; i < end; ++i) {
            sb[i].dLin = L.get(i, 0);
            sb[i].dAng = L.get(i, 1);
            sb[i].push = L.get(i, 2);
            sb[i].turn = L.get(i, 3);
        }
        for (uint32_t r = 0; r < n_points; ++r) {
            normalRow[r].applied = L.row(r, 0);
            frictionRow[r].applied = L.row(r, 2);
        }
    } else if (MID) {
        // (5 .. kIslMidBodies bodies, up to IslandParams::big_points contact points — a tower, a small pile: the bodies' delta velocities
        //  in LDS, a row's own scalars in the row; rows one ahead, a resolved row writes back the one word that changed)
        const IslLocal L{s_isl + (threadIdx.x & 63u), first};
        for (uint32_t i = first; i < end; ++i) {
            L.set(i, 0, sb[i].dLin);
            L.set(i, 1, sb[i].dAng);
            L.set(i, 2, F3{0.0f, 0.0f, 0.0f});
            L.set(i, 3, F3{0.0f, 0.0f, 0.0f});
        }
        for (int it = 0; it < kIterations; ++it) {
            bool any = false;
            for (uint32_t r = 0; r < n_points; ++r) any = any || normalRow[r].rhsPenetration != 0.0f;
            if (!any) break;
            IslRow cur = normalRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = normalRow[r + 1 < n_points ? r + 1 : r];
                if (cur.rhsPenetration) {
                    isl_resolve_split_lds(L, sb, cur, cur.appliedPush);
                    normalRow[r].appliedPush = cur.appliedPush;
                }
                cur = nxt;
            }
        }
        for (int it = 0; it < kIterations; ++it) {
            if (n_points == 0) break;
            IslRow cur = normalRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = normalRow[r + 1 < n_points ? r + 1 : r];
                isl_resolve_row_lds(L, sb, cur, cur.applied, cur.lower, cur.upper, false);
                normalRow[r].applied = cur.applied;
                frictionRow[r].appliedPush = cur.applied; // (a friction row has no push impulse: the word carries its contact row's impulse to
                                                          //  the friction sweep inside the row — no load of its own behind the next row's)
                cur = nxt;
            }
            cur = frictionRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = frictionRow[r + 1 < n_points ? r + 1 : r];
                const float totalImpulse = cur.appliedPush;
                if (totalImpulse > 0.0f) {
                    const float friction = cur.friction;
                    isl_resolve_row_lds(L, sb, cur, cur.applied, -(friction * totalImpulse), friction * totalImpulse, true);
                    frictionRow[r].applied = cur.applied;
                }
                cur = nxt;
            }
        }
        for (uint32_t i = first; i < end; ++i) {
            sb[i].dLin = L.get(i, 0);
            sb[i].dAng = L.get(i, 1);
            sb[i].push = L.get(i, 2);
            sb[i].turn = L.get(i, 3);
        }
    } else {
        // (more than kIslMidBodies bodies on at most IslandParams::big_points contact points — rare —: everything in global memory, still
        //  one thread; rows one ahead, a resolved row writes back the one word that changed)
        for (int it = 0; it < kIterations; ++it) {
            bool any = false;
            for (uint32_t r = 0; r < n_points; ++r) any = any || normalRow[r].rhsPenetration != 0.0f;
            if (!any) break;
            IslRow cur = normalRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = normalRow[r + 1 < n_points ? r + 1 : r];
                if (cur.rhsPenetration) {
                    isl_resolve_split(sb, cur);
                    normalRow[r].appliedPush = cur.appliedPush;
                }
                cur = nxt;
            }
        }
        for (int it = 0; it < kIterations; ++it) {
            if (n_points == 0) break;
            IslRow cur = normalRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = normalRow[r + 1 < n_points ? r + 1 : r];
                isl_resolve_row(sb, cur, false);
                normalRow[r].applied = cur.applied;
                cur = nxt;
            }
            cur = frictionRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = frictionRow[r + 1 < n_points ? r + 1 : r];
                const float totalImpulse = normalRow[r].applied;
                if (totalImpulse > 0.0f) {
                    cur.lower = -(cur.friction * totalImpulse);
                    cur.upper = cur.friction * totalImpulse;
                    isl_resolve_row(sb, cur, true);
                    frictionRow[r].applied = cur.applied;
                }
                cur = nxt;
            }
        }
    }
    // ---- solveGroupCacheFriendlyFinish
    for (uint32_t r = 0; r < n_points; ++r) {
        normalRow[r].out[0] = normalRow[r].applied;
        normalRow[r].out[normalRow[r].lateral_at] = frictionRow[r].applied;
    }
    for (uint32_t i = first; i < end; ++i) isl_finish_body<BASIS>(w, g, sb, i);
}

// the warm start of one contact row (the block isl_add_contact<true> runs in place)
__device__ __forceinline__ void isl_warm_start(IslBody* sb, const IslRow& c)
{
    IslBody& A = sb[c.a];
    const F3 n = c.normal;
    const F3 lin = F3{c.normal.x * A.invMass, c.normal.y * A.invMass, c.normal.z * A.invMass};
    A.dLin = add3(A.dLin, scale3(lin, c.applied));
    A.dAng = add3(A.dAng, scale3(c.angularComp, c.applied * 1.0f));
    if (c.b != kNone) {
        IslBody& B = sb[c.b];
        const F3 linB = F3{B.invMass * n.x, B.invMass * n.y, B.invMass * n.z};
        B.dLin = sub3(B.dLin, scale3(linB, c.applied));
        B.dAng = add3(B.dAng, scale3(c.angularCompB, c.applied * 1.0f));
    }
}

template <class Local>
__device__ __forceinline__ void isl_warm_start_lds(const Local& L, const IslBody* sb, const IslRow& c)
{
    const float invMassA = c.invMassA;
    const F3 n = c.normal;
    const F3 lin = F3{c.normal.x * invMassA, c.normal.y * invMassA, c.normal.z * invMassA};
    L.set(c.a, 0, add3(L.get(c.a, 0), scale3(lin, c.applied)));
    L.set(c.a, 1, add3(L.get(c.a, 1), scale3(c.angularComp, c.applied * 1.0f)));
    if (c.b != kNone) {
        const float invMassB = c.invMassB;
        const F3 linB = F3{invMassB * n.x, invMassB * n.y, invMassB * n.z};
        L.set(c.b, 0, sub3(L.get(c.b, 0), scale3(linB, c.applied)));
        L.set(c.b, 1, add3(L.get(c.b, 1), scale3(c.angularCompB, c.applied * 1.0f)));
    }
}

// Exclusive scan of a[0 .. n) in place by the workgroup (256 threads, contiguous chunks); returns the total.  Ends with a barrier.
__device__ uint32_t isl_wg_scan(uint32_t* a, uint32_t n, uint32_t stride, uint32_t* s_part, uint32_t* s_total)
{
    const uint32_t tid = threadIdx.x, chunk = (n + 255u) / 256u;
    const uint32_t lo = tid * chunk < n ? tid * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
    uint32_t sum = 0;
    for (uint32_t k = lo; k < hi; ++k) sum += a[static_cast<uint64_t>(k) * stride];
    s_part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < 256u; ++k) {
            const uint32_t v = s_part[k];
            s_part[k] = run;
            run += v;
        }
        *s_total = run;
    }
    __syncthreads();
    uint32_t run = s_part[tid];
    for (uint32_t k = lo; k < hi; ++k) {
        const uint32_t v = a[static_cast<uint64_t>(k) * stride];
        a[static_cast<uint64_t>(k) * stride] = run;
        run += v;
    }
    __syncthreads();
    return *s_total;
}

// ---- an island too big for one thread: a workgroup of 256 and Bullet's row order kept by LEVELS.  Gauss-Seidel is sequential in the
//      rows that share a body, and only in those: row r gets level 1 + max(level of the last earlier row of body A, of body B); rows of one
//      level touch pairwise different bodies and commute exactly, rows of a lower level come first as they do in the sequence.  So every
//      sweep — the warm start, ten split-impulse sweeps, ten sweeps of contact rows and of friction rows — walks the levels with a barrier
//      between them and the rows of a level side by side: the same operations on the same operands as the one-thread walk, bit for bit.
//      (A heap of 2,000 boxes: ~1,300 rows in ~30 levels.)  Bodies, rows and the level lists live in global memory; workgroups take
//      islands off the list k_island_solve left (ticket).
constexpr uint32_t kIslBigLdsBytes = 144u * 1024u, kIslBigLastLds = kIslBigLdsBytes / 4u, kIslBigLdsBodies = kIslBigLdsBytes / 48u;
template <bool BASIS>
__global__ void __launch_bounds__(256) k_island_solve_big(WorldView w, GroundParams g, IslandParams ip)
{
    __shared__ uint32_t s_part[256];
    // kIslBigLdsBytes of LDS, twice: while the levels are computed it holds the level of every body's last row, during the sweeps the
    // bodies' delta velocities (dLin, dAng, push, turn: 48 bytes a body) — after a level's barrier a row then waits for an LDS round trip,
    // not for the stores of the level before to reach L2 and come back
    extern __shared__ float s_dyn[];
    uint32_t* s_last = reinterpret_cast<uint32_t*>(s_dyn);
    __shared__ uint32_t s_ticket, s_total, s_depth, s_rows_at, s_ints_at, s_fail, s_any;
    IslBody* sb = static_cast<IslBody*>(ip.solver_bodies);
    const uint32_t tid = threadIdx.x;
    const int kIterations = static_cast<int>(ip.iterations);
    const float invTimeStep = 1.0f / g.dt;
    for (;;) {
        __syncthreads();
        if (tid == 0) s_ticket = atomicAdd(&ip.counts[5], 1u);
        __syncthreads();
        const uint32_t t = s_ticket;
        if (t >= ip.counts[4]) return;
        const uint32_t first = ip.big_list[2u * t], end = ip.big_list[2u * t + 1u], nb = end - first;
        // convertBodies; the rows of every body
        for (uint32_t i = first + tid; i < end; i += 256u) {
            ip.body_words[2u * i] = isl_prepare_body<BASIS>(w, g, ip, sb, i) + isl_pair_points(ip, ip.body_slot[i]);
        }
        __syncthreads();
        const uint32_t P = isl_wg_scan(ip.body_words + 2ull * first, nb, 2u, s_part, &s_total);
        if (P == 0) { // (bodies in each other's AABBs, nothing touches: gravity and the gyroscopic term only)
            for (uint32_t i = first + tid; i < end; i += 256u) isl_finish_body<BASIS>(w, g, sb, i);
            continue;
        }
        if (tid == 0) {
            s_fail = 0u;
            s_rows_at = atomicAdd(&ip.counts[2], 2u * P);
            s_ints_at = atomicAdd(&ip.counts[6], 4u * P + 8u);
            if (s_rows_at + 2u * P > ip.row_cap || s_ints_at + 4u * P + 8u > ip.int_cap) {
                atomicOr(&ip.counts[3], 1u); // (cannot happen: both pools hold every point the manifolds can hold)
                s_fail = 1u;
            }
        }
        __syncthreads();
        if (s_fail) continue;
        IslRow* normalRow = static_cast<IslRow*>(ip.rows) + s_rows_at;
        IslRow* frictionRow = normalRow + P;
        uint32_t* level = ip.ints + s_ints_at;  // [P] level of row r (1 ..)
        uint32_t* order = level + P;            // [P] rows in level order
        uint32_t* start = order + P;            // [depth + 2] first entry of level l in `order`
        uint32_t* cursor = start + P + 4u;      // [depth + 2]
        // convertContacts without the warm start, body by body
        for (uint32_t i = first + tid; i < end; i += 256u) {
            isl_build_body_rows<false>(w, g, ip, sb, i, invTimeStep, normalRow, frictionRow, ip.body_words[2u * i]);
        }
        // the levels: one walk over the rows in their order (integers only).  Where it fits, the walk runs out of LDS: the two body
        // numbers of every row are fetched by all threads first (a walk that waits for a global load per row took 2 of this kernel's
        // 2.9 ms on a 2,000-box heap)
        const bool walk_in_lds = 12ull * P + 4ull * nb <= kIslBigLdsBytes;
        const bool last_in_lds = nb <= kIslBigLastLds;
        uint32_t* l_ab = s_last + nb;          // [P][2] (walk_in_lds)
        uint32_t* l_level = l_ab + 2ull * P;   // [P]
        if (tid == 0) s_any = 0u;
        for (uint32_t k = tid; k < nb; k += 256u) {
            if (last_in_lds) s_last[k] = 0u;
            else ip.body_words[2u * (first + k) + 1u] = 0u;
        }
        __syncthreads();
        {
            uint32_t any = 0u;
            for (uint32_t r = tid; r < P; r += 256u) {
                any |= normalRow[r].rhsPenetration != 0.0f ? 1u : 0u;
                if (walk_in_lds) {
                    const uint32_t b = normalRow[r].b;
                    l_ab[2u * r] = normalRow[r].a - first;
                    l_ab[2u * r + 1u] = b == kNone ? kNone : b - first;
                }
            }
            if (any) atomicOr(&s_any, 1u);
        }
        __syncthreads();
        if (tid == 0) {
            uint32_t depth = 0;
            if (walk_in_lds) {
                for (uint32_t r = 0; r < P; ++r) {
                    const uint32_t a = l_ab[2u * r], b = l_ab[2u * r + 1u];
                    uint32_t l = s_last[a];
                    if (b != kNone) {
                        const uint32_t lb = s_last[b];
                        l = lb > l ? lb : l;
                    }
                    l += 1u;
                    s_last[a] = l;
                    if (b != kNone) s_last[b] = l;
                    l_level[r] = l;
                    depth = l > depth ? l : depth;
                }
            } else {
                for (uint32_t r = 0; r < P; ++r) {
                    const uint32_t a = normalRow[r].a - first, b = normalRow[r].b;
                    uint32_t l = last_in_lds ? s_last[a] : ip.body_words[2u * (first + a) + 1u];
                    if (b != kNone) {
                        const uint32_t lb = last_in_lds ? s_last[b - first] : ip.body_words[2u * b + 1u];
                        l = lb > l ? lb : l;
                    }
                    l += 1u;
                    if (last_in_lds) {
                        s_last[a] = l;
                        if (b != kNone) s_last[b - first] = l;
                    } else {
                        ip.body_words[2u * (first + a) + 1u] = l;
                        if (b != kNone) ip.body_words[2u * b + 1u] = l;
                    }
                    level[r] = l;
                    depth = l > depth ? l : depth;
                }
            }
            s_depth = depth;
        }
        __syncthreads();
        if (walk_in_lds) {
            for (uint32_t r = tid; r < P; r += 256u) level[r] = l_level[r];
        }
        const uint32_t depth = s_depth;
        for (uint32_t k = tid; k < depth + 2u; k += 256u) start[k] = 0u;
        __syncthreads();
        for (uint32_t r = tid; r < P; r += 256u) atomicAdd(&start[level[r]], 1u);
        __syncthreads();
        isl_wg_scan(start, depth + 2u, 1u, s_part, &s_total); // start[l] = rows of the levels below l; start[depth + 1] = P
        for (uint32_t k = tid; k < depth + 2u; k += 256u) cursor[k] = start[k];
        __syncthreads();
        for (uint32_t r = tid; r < P; r += 256u) order[atomicAdd(&cursor[level[r]], 1u)] = r;
        __syncthreads();
        // One sweep over the rows of `arr` in level order: fn(row copy, row number) for every row, a barrier after every level.  A thread's
        // first row of the NEXT level is requested before this level's barrier (the row's constants never change, and what does change in
        // it — its applied impulse — is only ever written by this same thread, which has the same place in every sweep): after the barrier a
        // row waits for its bodies only.
        auto sweep = [&](IslRow* arr, auto&& fn) {
            uint32_t kn = start[1] + tid, rn = 0;
            bool hv = kn < start[2];
            IslRow nx{};
            if (hv) {
                rn = order[kn];
                nx = arr[rn];
            }
            for (uint32_t l = 1; l <= depth; ++l) {
                IslRow cur = nx;
                const uint32_t r = rn;
                const bool have = hv;
                hv = false;
                if (l < depth) {
                    kn = start[l + 1u] + tid;
                    hv = kn < start[l + 2u];
                    if (hv) {
                        rn = order[kn];
                        nx = arr[rn];
                    }
                }
                if (have) fn(cur, r);
                for (uint32_t k = start[l] + tid + 256u; k < start[l + 1u]; k += 256u) {
                    const uint32_t r2 = order[k];
                    IslRow c2 = arr[r2];
                    fn(c2, r2);
                }
                __syncthreads();
            }
        };
        const bool bodies_in_lds = nb <= kIslBigLdsBodies;
        const IslLocalT<1u> L{s_dyn, first};
        if (bodies_in_lds) { // (the levels are computed: their LDS now holds the bodies' delta velocities)
            for (uint32_t k = tid; k < nb * 12u; k += 256u) s_dyn[k] = 0.0f;
            __syncthreads();
        }
        // the warm start, in the rows' order
        if (bodies_in_lds) sweep(normalRow, [&](IslRow& c, uint32_t) { isl_warm_start_lds(L, sb, c); });
        else sweep(normalRow, [&](IslRow& c, uint32_t) { isl_warm_start(sb, c); });
        // solveGroupCacheFriendlySplitImpulseIterations
        if (s_any) {
            for (int it = 0; it < kIterations; ++it) {
                sweep(normalRow, [&](IslRow& c, uint32_t r) {
                    if (c.rhsPenetration) {
                        if (bodies_in_lds) isl_resolve_split_lds(L, sb, c, c.appliedPush);
                        else isl_resolve_split(sb, c);
                        normalRow[r].appliedPush = c.appliedPush;
                    }
                });
            }
        }
        // solveGroupCacheFriendlyIterations: all contact rows, then all friction rows
        for (int it = 0; it < kIterations; ++it) {
            sweep(normalRow, [&](IslRow& c, uint32_t r) {
                if (bodies_in_lds) isl_resolve_row_lds(L, sb, c, c.applied, c.lower, c.upper, false);
                else isl_resolve_row(sb, c, false);
                normalRow[r].applied = c.applied;
                frictionRow[r].appliedPush = c.applied; // (carries the contact row's impulse to the friction sweep inside the friction row)
            });
            sweep(frictionRow, [&](IslRow& c, uint32_t r) {
                const float totalImpulse = c.appliedPush;
                if (totalImpulse > 0.0f) {
                    c.lower = -(c.friction * totalImpulse);
                    c.upper = c.friction * totalImpulse;
                    if (bodies_in_lds) isl_resolve_row_lds(L, sb, c, c.applied, c.lower, c.upper, true);
                    else isl_resolve_row(sb, c, true);
                    frictionRow[r].applied = c.applied;
                }
            });
        }
        if (bodies_in_lds) {
            for (uint32_t i = first + tid; i < end; i += 256u) {
                sb[i].dLin = L.get(i, 0);
                sb[i].dAng = L.get(i, 1);
                sb[i].push = L.get(i, 2);
                sb[i].turn = L.get(i, 3);
            }
            // (each thread finishes the bodies it has just written: no barrier needed before isl_finish_body below — same i, same thread)
        }
        // solveGroupCacheFriendlyFinish
        for (uint32_t r = tid; r < P; r += 256u) {
            normalRow[r].out[0] = normalRow[r].applied;
            normalRow[r].out[normalRow[r].lateral_at] = frictionRow[r].applied;
        }
        for (uint32_t i = first + tid; i < end; i += 256u) isl_finish_body<BASIS>(w, g, sb, i);
    }
}

} // namespace

hipError_t launch_ground(hipStream_t stream, const WorldView& w, const GroundParams& g, bool bullet_basis)
{
    if (g.n_slots == 0) return hipSuccess;
    const dim3 sgrid(static_cast<uint32_t>((g.n_slots + 255) / 256)), sblock(256);
    // what is resident: 2 x BGE_GROUND_MIN_BLOCKS workgroups of 128 threads on each of the 256 CUs (a multiple of the shard count)
    const dim3 grid(512u * BGE_GROUND_MIN_BLOCKS), block(128);
    const bool boxes = g.box_list != nullptr; // Static / Kinematic box colliders are on
    if (!g.obstacles_ready) {
        if (boxes && g.n_obstacles) hipLaunchKernelGGL(k_obstacles, dim3((g.n_obstacles + 63u) / 64u), dim3(64), 0, stream, w, g);
        if (boxes && g.obstacle_grid) hipLaunchKernelGGL(k_obstacle_grid, dim3(1), dim3(1024), 0, stream, g);
    }
    if (bullet_basis) {
        hipLaunchKernelGGL(k_ground_select<true>, sgrid, sblock, 0, stream, w, g);
        if (g.plane) hipLaunchKernelGGL(k_ground<true>, grid, block, 0, stream, w, g);
        if (boxes) hipLaunchKernelGGL(k_contact_boxes<true>, dim3(256), dim3(64), 0, stream, w, g);
    } else {
        hipLaunchKernelGGL(k_ground_select<false>, sgrid, sblock, 0, stream, w, g);
        if (g.plane) hipLaunchKernelGGL(k_ground<false>, grid, block, 0, stream, w, g);
        if (boxes) hipLaunchKernelGGL(k_contact_boxes<false>, dim3(256), dim3(64), 0, stream, w, g);
    }
    return hipGetLastError();
}

hipError_t launch_obstacles(hipStream_t stream, const WorldView& w, const GroundParams& g)
{
    if (g.box_list && g.n_obstacles) hipLaunchKernelGGL(k_obstacles, dim3((g.n_obstacles + 63u) / 64u), dim3(64), 0, stream, w, g);
    if (g.box_list && g.obstacle_grid) hipLaunchKernelGGL(k_obstacle_grid, dim3(1), dim3(1024), 0, stream, g);
    return hipGetLastError();
}

hipError_t launch_island_begin(hipStream_t stream, const WorldView& w, const GroundParams& g, const IslandParams& ip, bool bullet_basis)
{
    if (ip.n_slots == 0) return hipSuccess;
    const dim3 grid(static_cast<uint32_t>((ip.n_slots + 255) / 256)), block(256);
    if (bullet_basis) hipLaunchKernelGGL(k_island_begin<true>, grid, block, 0, stream, w, g, ip);
    else hipLaunchKernelGGL(k_island_begin<false>, grid, block, 0, stream, w, g, ip);
    return hipGetLastError();
}

hipError_t launch_island_pair_keys(hipStream_t stream, const WorldView& w, const IslandParams& ip)
{
    hipLaunchKernelGGL(k_island_pair_keys, dim3(ip.bp_shards * 16u), dim3(256), 0, stream, w, ip);
    return hipGetLastError();
}

// (tmp == nullptr: only the size of the temporary storage is returned)
hipError_t island_sort_keys(hipStream_t stream, void* tmp, size_t& tmp_bytes, const uint64_t* in, uint64_t* out, uint32_t n)
{
    return hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, in, out, static_cast<int>(n), 0, 64, stream);
}

hipError_t island_sort_pairs(hipStream_t stream, void* tmp, size_t& tmp_bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, uint32_t n)
{
    return hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, kin, kout, vin, vout, static_cast<int>(n), 0, 64, stream);
}

hipError_t launch_island_build(hipStream_t stream, const WorldView& w, const IslandParams& ip, bool orphans)
{
    if (ip.n_pairs) {
        const dim3 grid((ip.n_pairs + 255u) / 256u), block(256);
        hipLaunchKernelGGL(k_island_carry, grid, block, 0, stream, ip);
        hipLaunchKernelGGL(k_island_narrow, dim3((ip.n_pairs + 63u) / 64u), dim3(64), 0, stream, w, ip);
        hipLaunchKernelGGL(k_island_union, grid, block, 0, stream, ip);
        hipLaunchKernelGGL(k_island_members, grid, block, 0, stream, w, ip);
    }
    if (orphans) hipLaunchKernelGGL(k_island_orphans, dim3(static_cast<uint32_t>((ip.n_slots + 255) / 256)), dim3(256), 0, stream, w, ip);
    return hipGetLastError();
}

hipError_t launch_island_solve(hipStream_t stream, const WorldView& w, const GroundParams& g, const IslandParams& ip, bool bullet_basis)
{
    if (ip.n_bodies == 0) return hipSuccess;
    hipLaunchKernelGGL(k_island_flags, dim3((ip.n_bodies + 255u) / 256u), dim3(256), 0, stream, w, ip);
    const dim3 grid((ip.n_bodies + 63u) / 64u), block(64);
    const dim3 mid_grid((ip.n_bodies / (kIslLdsBodies + 1u) + 64u) / 64u); // (an island on the mid list has more than kIslLdsBodies bodies)
    if (bullet_basis) {
        hipLaunchKernelGGL(k_island_own<true>, grid, block, 0, stream, w, g, ip);
        hipLaunchKernelGGL((k_island_solve<true, false>), grid, block, 0, stream, w, g, ip);
        hipLaunchKernelGGL((k_island_solve<true, true>), mid_grid, block, 0, stream, w, g, ip);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_island_solve_big<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kIslBigLdsBytes));
        hipLaunchKernelGGL(k_island_solve_big<true>, dim3(256), dim3(256), kIslBigLdsBytes, stream, w, g, ip);
    } else {
        hipLaunchKernelGGL(k_island_own<false>, grid, block, 0, stream, w, g, ip);
        hipLaunchKernelGGL((k_island_solve<false, false>), grid, block, 0, stream, w, g, ip);
        hipLaunchKernelGGL((k_island_solve<false, true>), mid_grid, block, 0, stream, w, g, ip);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_island_solve_big<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kIslBigLdsBytes));
        hipLaunchKernelGGL(k_island_solve_big<false>, dim3(256), dim3(256), kIslBigLdsBytes, stream, w, g, ip);
    }
    return hipGetLastError();
}

} // namespace bge
