// bge_contact_device.hpp — device functions of the contact paths, shared by bge_contact.hip (the plane, the Static / Kinematic box colliders)
// and bge_island.hip (Dynamic boxes against each other): vector helpers in the reference's compiled associations, the collider as Bullet
// holds it, the 4-point persistent manifolds, the box-box narrowphase glue, the one-body solver rows and contact_body (collide + solve for
// ONE body with its own pairs).  Everything sits in an anonymous namespace: each translation unit gets its own copies.
#pragma once

#include <hip/hip_runtime.h>

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

} // namespace

} // namespace bge
