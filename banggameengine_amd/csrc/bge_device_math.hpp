// bge_device_math.hpp — gfx950 device arithmetic of the world tick.
//
// Everything here is IEEE-754 binary32 with one rounding per operation (the translation unit is
// compiled with -ffp-contract=off; divide and sqrt are HIP's correctly-rounded defaults), so the
// results are the ones the reference's scalar-SSE /fp:precise build produces for the same inputs:
//   bx_*      bx::floor / cos / sin / mtxSRT / mtxMul as called from src/ecs/Transform.cpp:18-36
//             (constants and operation order: SURVEY.md §8 a-3 / a-4)
//   bt_*      the Bullet pieces behind src/physics/PhysicsSystem.cpp:40-64, 863, 937-947
//             (setEulerZYX, setRotation, getEulerZYX, integrateTransform, btTransformAabb);
//             libm calls go through include/bge_detmath.h so host and device agree bit for bit.
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/bge_detmath.h"

namespace bge {
namespace dev {

struct F3 {
    float x, y, z;
};
struct Q4 {
    float x, y, z, w;
};

__device__ __forceinline__ F3 ld3(const float* __restrict__ base, uint32_t slot)
{
    const float* p = base + 3ull * slot;
    return F3{p[0], p[1], p[2]};
}
__device__ __forceinline__ void st3(float* __restrict__ base, uint32_t slot, const F3& v)
{
    float* p = base + 3ull * slot;
    p[0] = v.x;
    p[1] = v.y;
    p[2] = v.z;
}
__device__ __forceinline__ Q4 ld4(const float* __restrict__ base, uint32_t slot)
{
    const float4 v = reinterpret_cast<const float4*>(base)[slot];
    return Q4{v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ void st4(float* __restrict__ base, uint32_t slot, const Q4& q)
{
    reinterpret_cast<float4*>(base)[slot] = make_float4(q.x, q.y, q.z, q.w);
}

// ------------------------------------------------------------------ bx
typedef float f32x2 __attribute__((ext_vector_type(2)));

// bx::floor is built from int casts (a - fract(a) with a +1 fix-up for negatives); for |a| < 2^31 it equals
// floorf except that it returns +0 for -0.  bx_cos only feeds it `scaled` and uses the result as
// `real * kPiHalf` and `int(real) & 3`, where the sign of a zero cannot change the outcome (quadrant 0 multiplies
// by c0 = 1), so the single-instruction v_floor_f32 is used.  tests/test_gpu_parity.py compares against the
// cast-based restatement in the oracle, including negative, tiny and -0 angles.
__device__ __forceinline__ float bx_cos(float a)
{
    const float kPiHalf = 1.5707963267948966f;
    const float kInvPi = 0.31830988618379067f;
    const float scaled = (a * 2.0f) * kInvPi;
    const float real = __builtin_floorf(scaled);
    const float xx = a - real * kPiHalf;
    const int quadrant = static_cast<int>(real) & 3;

    // Both coefficient sets ride one packed Horner chain (v_pk_mul_f32 / v_pk_add_f32: one IEEE rounding per
    // component and operation, exactly the scalar sequence); .x = quadrants 0/2, .y = quadrants 1/3.
    const f32x2 c2 = {-0.5f, __uint_as_float(0xbe2aaaabu)};
    const f32x2 c4 = {__uint_as_float(0x3d2aaaa4u), __uint_as_float(0x3c088898u)};
    const f32x2 c6 = {__uint_as_float(0xbab60981u), __uint_as_float(0xb9501096u)};
    const f32x2 c8 = {__uint_as_float(0x37cfab9cu), __uint_as_float(0x363938a8u)};
    const f32x2 c10 = {__uint_as_float(0xb48b634du), __uint_as_float(0xb2d70013u)};
    const f32x2 one = {1.0f, 1.0f};
    const float xsq1 = xx * xx;
    const f32x2 xsq = {xsq1, xsq1};
    f32x2 acc = c10 * xsq + c8;
    acc = acc * xsq + c6;
    acc = acc * xsq + c4;
    acc = acc * xsq + c2;
    acc = acc * xsq + one;
    // result = acc * c0 with c0 = 1 (even quadrant: the product is acc itself) or xx (odd)
    const float odd = acc.y * xx;
    const float result = (quadrant & 1) ? odd : acc.x;
    // negate in quadrants 1 and 2
    const uint32_t sign = (static_cast<uint32_t>(quadrant + 1) & 2u) << 30;
    return __uint_as_float(__float_as_uint(result) ^ sign);
}

__device__ __forceinline__ float bx_sin(float a) { return bx_cos(a - 1.5707963267948966f); }

// local = S * R(euler) * T, row-major (bx::mtxSRT)
__device__ __forceinline__ void bx_mtx_srt(float (&m)[16], const F3& s, const F3& r, const F3& t)
{
    const float sx = bx_sin(r.x);
    const float cx = bx_cos(r.x);
    const float sy = bx_sin(r.y);
    const float cy = bx_cos(r.y);
    const float sz = bx_sin(r.z);
    const float cz = bx_cos(r.z);
    const float sxsz = sx * sz;
    const float cycz = cy * cz;

    m[0] = s.x * (cycz - sxsz * sy);
    m[1] = (s.x * -cx) * sz;
    m[2] = s.x * (sxsz * cy + cz * sy);
    m[3] = 0.0f;
    m[4] = s.y * ((cz * sx) * sy + sz * cy);
    m[5] = (s.y * cx) * cz;
    m[6] = s.y * (sz * sy - cycz * sx);
    m[7] = 0.0f;
    m[8] = (s.z * -cx) * sy;
    m[9] = s.z * sx;
    m[10] = (s.z * cx) * cy;
    m[11] = 0.0f;
    m[12] = t.x;
    m[13] = t.y;
    m[14] = t.z;
    m[15] = 1.0f;
}

// out = a * b, each element ((a0*b[j] + a1*b[4+j]) + a2*b[8+j]) + a3*b[12+j]   (bx::vec4MulMtx)
// Two output columns per instruction (v_pk_mul_f32 / v_pk_add_f32): 56 packed operations instead of 112 scalar ones,
// the same IEEE rounding per component and operation and the same association, so the bits do not change.
__device__ __forceinline__ void bx_mtx_mul(float (&o)[16], const float (&a)[16], const float (&b)[16])
{
#ifdef BGE_EXPERIMENT_SCALAR_MTXMUL /* timing-only A/B build */
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[4 * i + j] = ((a[4 * i + 0] * b[j] + a[4 * i + 1] * b[4 + j]) + a[4 * i + 2] * b[8 + j])
                           + a[4 * i + 3] * b[12 + j];
        }
    }
#else
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 b0 = {b[2 * h], b[2 * h + 1]}, b1 = {b[4 + 2 * h], b[5 + 2 * h]};
        const f32x2 b2 = {b[8 + 2 * h], b[9 + 2 * h]}, b3 = {b[12 + 2 * h], b[13 + 2 * h]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 a0 = {a[4 * i], a[4 * i]}, a1 = {a[4 * i + 1], a[4 * i + 1]};
            const f32x2 a2 = {a[4 * i + 2], a[4 * i + 2]}, a3 = {a[4 * i + 3], a[4 * i + 3]};
            const f32x2 r = ((a0 * b0 + a1 * b1) + a2 * b2) + a3 * b3;
            o[4 * i + 2 * h] = r.x;
            o[4 * i + 2 * h + 1] = r.y;
        }
    }
#endif
}

// One row of bx_mtx_mul: row i of a * b from row i of a — the same operations on the same operands, so the same bits.  Used where
// `a` arrives a row at a time (from LDS or memory) and the product leaves a row at a time: 8 live registers besides b
// instead of 32.
__device__ __forceinline__ float4 bx_mtx_mul_row(const float4& a, const float (&b)[16])
{
#ifdef BGE_EXPERIMENT_SCALAR_MTXMUL
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = ((a.x * b[j] + a.y * b[4 + j]) + a.z * b[8 + j]) + a.w * b[12 + j];
    return make_float4(o[0], o[1], o[2], o[3]);
#else
    const f32x2 a0 = {a.x, a.x}, a1 = {a.y, a.y}, a2 = {a.z, a.z}, a3 = {a.w, a.w};
    f32x2 r[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 b0 = {b[2 * h], b[2 * h + 1]}, b1 = {b[4 + 2 * h], b[5 + 2 * h]};
        const f32x2 b2 = {b[8 + 2 * h], b[9 + 2 * h]}, b3 = {b[12 + 2 * h], b[13 + 2 * h]};
        r[h] = ((a0 * b0 + a1 * b1) + a2 * b2) + a3 * b3;
    }
    return make_float4(r[0].x, r[0].y, r[1].x, r[1].y);
#endif
}

// normalMtx = transpose(inverse(world)) as Renderer::BeginFrame computes it per mesh entity
// (src/render/Renderer.cpp:633-636: bx::mtxInverse then bx::mtxTranspose).  bx::mtxInverse is the adjugate over the
// determinant with the cofactors expanded along the first row of each minor; the transpose is folded into the stores.
__device__ __forceinline__ void bx_normal_matrix(float (&r)[16], const float (&a)[16])
{
    const float xx = a[0], xy = a[1], xz = a[2], xw = a[3];
    const float yx = a[4], yy = a[5], yz = a[6], yw = a[7];
    const float zx = a[8], zy = a[9], zz = a[10], zw = a[11];
    const float wx = a[12], wy = a[13], wz = a[14], ww = a[15];

    float det = 0.0f;
    det += xx * (yy * (zz * ww - zw * wz) - yz * (zy * ww - zw * wy) + yw * (zy * wz - zz * wy));
    det -= xy * (yx * (zz * ww - zw * wz) - yz * (zx * ww - zw * wx) + yw * (zx * wz - zz * wx));
    det += xz * (yx * (zy * ww - zw * wy) - yy * (zx * ww - zw * wx) + yw * (zx * wy - zy * wx));
    det -= xw * (yx * (zy * wz - zz * wy) - yy * (zx * wz - zz * wx) + yz * (zx * wy - zy * wx));
    const float invDet = 1.0f / det;

    // inverse element [i][j] lands at r[4*j + i]
    r[0] = +(yy * (zz * ww - wz * zw) - yz * (zy * ww - wy * zw) + yw * (zy * wz - wy * zz)) * invDet;
    r[4] = -(xy * (zz * ww - wz * zw) - xz * (zy * ww - wy * zw) + xw * (zy * wz - wy * zz)) * invDet;
    r[8] = +(xy * (yz * ww - wz * yw) - xz * (yy * ww - wy * yw) + xw * (yy * wz - wy * yz)) * invDet;
    r[12] = -(xy * (yz * zw - zz * yw) - xz * (yy * zw - zy * yw) + xw * (yy * zz - zy * yz)) * invDet;

    r[1] = -(yx * (zz * ww - wz * zw) - yz * (zx * ww - wx * zw) + yw * (zx * wz - wx * zz)) * invDet;
    r[5] = +(xx * (zz * ww - wz * zw) - xz * (zx * ww - wx * zw) + xw * (zx * wz - wx * zz)) * invDet;
    r[9] = -(xx * (yz * ww - wz * yw) - xz * (yx * ww - wx * yw) + xw * (yx * wz - wx * yz)) * invDet;
    r[13] = +(xx * (yz * zw - zz * yw) - xz * (yx * zw - zx * yw) + xw * (yx * zz - zx * yz)) * invDet;

    r[2] = +(yx * (zy * ww - wy * zw) - yy * (zx * ww - wx * zw) + yw * (zx * wy - wx * zy)) * invDet;
    r[6] = -(xx * (zy * ww - wy * zw) - xy * (zx * ww - wx * zw) + xw * (zx * wy - wx * zy)) * invDet;
    r[10] = +(xx * (yy * ww - wy * yw) - xy * (yx * ww - wx * yw) + xw * (yx * wy - wx * yy)) * invDet;
    r[14] = -(xx * (yy * zw - zy * yw) - xy * (yx * zw - zx * yw) + xw * (yx * zy - zx * yy)) * invDet;

    r[3] = -(yx * (zy * wz - wy * zz) - yy * (zx * wz - wx * zz) + yz * (zx * wy - wx * zy)) * invDet;
    r[7] = +(xx * (zy * wz - wy * zz) - xy * (zx * wz - wx * zz) + xz * (zx * wy - wx * zy)) * invDet;
    r[11] = -(xx * (yy * wz - wy * yz) - xy * (yx * wz - wx * yz) + xz * (yx * wy - wx * yy)) * invDet;
    r[15] = +(xx * (yy * zz - zy * yz) - xy * (yx * zz - zx * yz) + xz * (yx * zy - zx * yy)) * invDet;
}

// ------------------------------------------------------------------ Bullet pieces
constexpr float kBtEpsilon = 1.1920928955078125e-07f;
constexpr float kBtPi = 3.1415926535897932384626433832795029f;
constexpr float kBtAngularMotionThreshold = 0.5f * (kBtPi * 0.5f);
constexpr float kBtContactBreakingThreshold = 0.02f;

struct M3 {
    float m[3][3];
};

// ToBtQuaternion(euler) = setEulerZYX(yaw = e.y, pitch = e.x, roll = e.z)
__device__ __forceinline__ Q4 bt_quat_from_transform_euler(const F3& e)
{
    const float halfYaw = e.y * 0.5f;
    const float halfPitch = e.x * 0.5f;
    const float halfRoll = e.z * 0.5f;
    const float cosYaw = bge_det_cosf(halfYaw);
    const float sinYaw = bge_det_sinf(halfYaw);
    const float cosPitch = bge_det_cosf(halfPitch);
    const float sinPitch = bge_det_sinf(halfPitch);
    const float cosRoll = bge_det_cosf(halfRoll);
    const float sinRoll = bge_det_sinf(halfRoll);
    Q4 q;
    // product order of bullet3's btQuaternion::setEulerZYX as the reference's build compiled it (roll factor first,
    // left-associated) — oracle/tools/check_bullet_order.py checks it against PhysicsSystem.obj
    q.x = sinRoll * cosPitch * cosYaw - cosRoll * sinPitch * sinYaw;
    q.y = cosRoll * sinPitch * cosYaw + sinRoll * cosPitch * sinYaw;
    q.z = cosRoll * cosPitch * sinYaw - sinRoll * sinPitch * cosYaw;
    q.w = cosRoll * cosPitch * cosYaw + sinRoll * sinPitch * sinYaw;
    return q;
}

__device__ __forceinline__ float bt_quat_length2(const Q4& q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }

__device__ __forceinline__ M3 bt_mat_from_quat(const Q4& q)
{
    const float d = bt_quat_length2(q);
    const float s = 2.0f / d;
    const float xs = q.x * s, ys = q.y * s, zs = q.z * s;
    const float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    const float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    const float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    M3 r;
    r.m[0][0] = 1.0f - (yy + zz);
    r.m[0][1] = xy - wz;
    r.m[0][2] = xz + wy;
    r.m[1][0] = xy + wz;
    r.m[1][1] = 1.0f - (xx + zz);
    r.m[1][2] = yz - wx;
    r.m[2][0] = xz - wy;
    r.m[2][1] = yz + wx;
    r.m[2][2] = 1.0f - (xx + yy);
    return r;
}

// btMatrix3x3::getRotation(q), scalar path (oracle/bullet_math.h QuatFromMat; both sides are checked against the
// reference's PhysicsSystem.obj by oracle/tools/check_bullet_order.py, the trace <= 0 side for each of its three i).  The
// run-time indices i, j, k of Bullet's code are spelled out as three cases: no dynamically indexed arrays (scratch).
__device__ __forceinline__ Q4 bt_quat_from_mat(const M3& a)
{
    const float trace = a.m[0][0] + a.m[1][1] + a.m[2][2];
    if (trace > 0.0f) {
        float s = __builtin_sqrtf(trace + 1.0f);
        const float w = s * 0.5f;
        s = 0.5f / s;
        return Q4{(a.m[2][1] - a.m[1][2]) * s, (a.m[0][2] - a.m[2][0]) * s, (a.m[1][0] - a.m[0][1]) * s, w};
    }
    const int i = a.m[0][0] < a.m[1][1] ? (a.m[1][1] < a.m[2][2] ? 2 : 1) : (a.m[0][0] < a.m[2][2] ? 2 : 0);
    // t[i] = s/2, t[3] = (m[k][j] - m[j][k]) s', t[j] = (m[j][i] + m[i][j]) s', t[k] = (m[k][i] + m[i][k]) s'
    if (i == 0) { // j = 1, k = 2
        float s = __builtin_sqrtf(a.m[0][0] - a.m[1][1] - a.m[2][2] + 1.0f);
        const float ti = s * 0.5f;
        s = 0.5f / s;
        return Q4{ti, (a.m[1][0] + a.m[0][1]) * s, (a.m[2][0] + a.m[0][2]) * s, (a.m[2][1] - a.m[1][2]) * s};
    }
    if (i == 1) { // j = 2, k = 0
        float s = __builtin_sqrtf(a.m[1][1] - a.m[2][2] - a.m[0][0] + 1.0f);
        const float ti = s * 0.5f;
        s = 0.5f / s;
        return Q4{(a.m[0][1] + a.m[1][0]) * s, ti, (a.m[2][1] + a.m[1][2]) * s, (a.m[0][2] - a.m[2][0]) * s};
    }
    // i = 2, j = 0, k = 1
    float s = __builtin_sqrtf(a.m[2][2] - a.m[0][0] - a.m[1][1] + 1.0f);
    const float ti = s * 0.5f;
    s = 0.5f / s;
    return Q4{(a.m[0][2] + a.m[2][0]) * s, (a.m[1][2] + a.m[2][1]) * s, ti, (a.m[1][0] - a.m[0][1]) * s};
}

// The same function without branches, for BGE_TICK_BULLET_BASIS's queue (one wave per workgroup over bodies of mixed cases).
// (Used THERE only.  Substituted everywhere it made one case of the random-edit campaign differ by a few ulps in rotationEuler — seed 6, the
//  AABB variant's inline step — although the two functions agree bit for bit on 8 M matrices in a kernel of their own and over six
//  iterated steps of 2 M bodies: something in that context, not found yet)
__device__ __forceinline__ Q4 bt_quat_from_mat_sel(const M3& a)
{
    // btMatrix3x3::getRotation has four cases (trace > 0; the largest diagonal element otherwise), each with one square root and
    // one division.  As four branches a wave of mixed bodies ran all four one after the other — two thirds of this function's
    // instructions, and the function runs twice per queued body in BGE_TICK_BULLET_BASIS.  Here every lane SELECTS its case's
    // radicand and numerators and the wave shares one square root and one division: the same operations on the same operands per
    // lane, so the same bits (tests/test_gpu_parity.py's BASIS cases, the fuzz campaign).
    const float m00 = a.m[0][0], m11 = a.m[1][1], m22 = a.m[2][2];
    const float trace = m00 + m11 + m22;
    const bool t = trace > 0.0f;
    const int i = m00 < m11 ? (m11 < m22 ? 2 : 1) : (m00 < m22 ? 2 : 0);
    const float r_t = trace + 1.0f;
    const float r_0 = m00 - m11 - m22 + 1.0f;
    const float r_1 = m11 - m22 - m00 + 1.0f;
    const float r_2 = m22 - m00 - m11 + 1.0f;
    float s = __builtin_sqrtf(t ? r_t : (i == 0 ? r_0 : (i == 1 ? r_1 : r_2)));
    const float half = s * 0.5f;
    s = 0.5f / s;
    const float da = a.m[2][1] - a.m[1][2], db = a.m[0][2] - a.m[2][0], dc = a.m[1][0] - a.m[0][1];
    const float sa = a.m[2][1] + a.m[1][2], sb = a.m[2][0] + a.m[0][2], sc = a.m[1][0] + a.m[0][1];
    //          x    y    z    w
    // trace    da   db   dc   half
    // i = 0    half sc   sb   da
    // i = 1    sc   half sa   db
    // i = 2    sb   sa   half dc        (the sums are commutative: (m[1][0] + m[0][1]) and (m[0][1] + m[1][0]) are one float)
    const float nx = t ? da : (i == 1 ? sc : sb);
    const float ny = t ? db : (i == 0 ? sc : sa);
    const float nz = t ? dc : (i == 0 ? sb : sa);
    const float nw = i == 0 ? da : (i == 1 ? db : dc);
    Q4 q{nx * s, ny * s, nz * s, nw * s};
    if (t) q.w = half;
    else if (i == 0) q.x = half;
    else if (i == 1) q.y = half;
    else q.z = half;
    return q;
}

// Transform::rotationEuler written by SyncRigidBodiesFromPhysics: {pitch, yaw, roll} of getEulerZYX
// (the reference reads worldTransform.getRotation() and builds btMatrix3x3(rotation) before getEulerZYX: the basis takes a
//  getRotation -> setRotation round trip first, oracle/bullet_math.h TransformEulerFromMat)
template <bool SEL = false>
__device__ __forceinline__ F3 bt_transform_euler_from_mat(const M3& basis)
{
    const M3 a = bt_mat_from_quat(SEL ? bt_quat_from_mat_sel(basis) : bt_quat_from_mat(basis));
    float yaw, pitch, roll;
    if (__builtin_fabsf(a.m[2][0]) >= 1.0f) {
        yaw = 0.0f;
        const float delta = bge_det_atan2f(a.m[0][0], a.m[0][2]);
        if (a.m[2][0] > 0.0f) {
            pitch = kBtPi / 2.0f;
            roll = pitch + delta;
        } else {
            pitch = -kBtPi / 2.0f;
            roll = -pitch + delta;
        }
    } else {
        float sp = a.m[2][0];
        sp = sp < -1.0f ? -1.0f : sp;
        sp = sp > 1.0f ? 1.0f : sp;
        pitch = -bge_det_asinf(sp);
        const float c = bge_det_cosf(pitch);
        roll = bge_det_atan2f(a.m[2][1] / c, a.m[2][2] / c);
        yaw = bge_det_atan2f(a.m[1][0] / c, a.m[0][0] / c);
    }
    return F3{pitch, yaw, roll};
}

// rotation part of btTransformUtil::integrateTransform
// (A body that does not turn — w = +-0 — and whose quaternion has no zero component: dorn = (+-0, +-0, +-0, cos 0 = 1), and
//  dorn * orn0 is orn0 BIT FOR BIT: 1 * c plus or minus signed zeros is c for c != 0.  With a zero component the signed zeros decide
//  its sign, so those take the general path.  A wave all of whose lanes are in the first case skips the exponential map and the
//  quaternion product — ~90 of the step's instructions; in BGE_TICK_BULLET_BASIS's steady state that is every queued wave.
//  Checked on the CPU against oracle/bullet_math.h's IntegrateOrientation over 5.4 M random quaternions: 0 differences.)
__device__ __forceinline__ Q4 bt_integrate_orientation(const Q4& orn0, const F3& w, float dt)
{
    const bool still = w.x == 0.0f && w.y == 0.0f && w.z == 0.0f && orn0.x != 0.0f && orn0.y != 0.0f && orn0.z != 0.0f && orn0.w != 0.0f;
    Q4 r = orn0;
    if (__builtin_amdgcn_ballot_w64(!still) != 0ull) {
        const float fAngle2 = w.x * w.x + w.y * w.y + w.z * w.z;
        float fAngle = 0.0f;
        if (fAngle2 > kBtEpsilon) fAngle = __builtin_sqrtf(fAngle2);
        if (fAngle * dt > kBtAngularMotionThreshold) fAngle = kBtAngularMotionThreshold / dt;
        float k;
        if (fAngle < 0.001f) {
            // association as compiled in the reference's build: (dt*dt) * (dt * 1/48), oracle/tools/check_bullet_order.py
            k = 0.5f * dt - ((dt * dt) * (dt * 0.020833333333f)) * fAngle * fAngle;
        } else {
            k = bge_det_sinf(0.5f * fAngle * dt) / fAngle;
        }
        const Q4 a{w.x * k, w.y * k, w.z * k, bge_det_cosf(fAngle * dt * 0.5f)};
        const Q4& b = orn0;
        Q4 g;
        g.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
        g.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
        g.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
        g.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
        if (!still) r = g;
    }
    // safeNormalize's length2 is summed pairwise in the compiled integrateTransform: (x^2 + y^2) + (z^2 + w^2)
    const float l2 = (r.x * r.x + r.y * r.y) + (r.z * r.z + r.w * r.w);
    if (l2 > kBtEpsilon) {
        const float s = 1.0f / __builtin_sqrtf(l2);
        r.x *= s;
        r.y *= s;
        r.z *= s;
        r.w *= s;
    }
    return bt_quat_length2(r) > kBtEpsilon ? r : orn0;
}

// btTransformAabb + updateSingleAabb's contact threshold
__device__ __forceinline__ void bt_aabb_of_pose(const F3& o, const M3& r, const F3& he, float (&mn)[3], float (&mx)[3])
{
    const float ex = __builtin_fabsf(r.m[0][0]) * he.x + __builtin_fabsf(r.m[0][1]) * he.y + __builtin_fabsf(r.m[0][2]) * he.z;
    const float ey = __builtin_fabsf(r.m[1][0]) * he.x + __builtin_fabsf(r.m[1][1]) * he.y + __builtin_fabsf(r.m[1][2]) * he.z;
    const float ez = __builtin_fabsf(r.m[2][0]) * he.x + __builtin_fabsf(r.m[2][1]) * he.y + __builtin_fabsf(r.m[2][2]) * he.z;
    mn[0] = (o.x - ex) - kBtContactBreakingThreshold;
    mn[1] = (o.y - ey) - kBtContactBreakingThreshold;
    mn[2] = (o.z - ez) - kBtContactBreakingThreshold;
    mx[0] = (o.x + ex) + kBtContactBreakingThreshold;
    mx[1] = (o.y + ey) + kBtContactBreakingThreshold;
    mx[2] = (o.z + ez) + kBtContactBreakingThreshold;
}

} // namespace dev
} // namespace bge
