// oracle/bx_math.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the four bx functions the reference's transform path calls
// (reference call sites: src/ecs/Transform.cpp:8-9 mtxIdentity, :20-23 mtxSRT,
// :30 mtxMul).  bx itself (bkaradzic/bx, the revision vcpkg paired with bgfx commit
// 4cc041f59f90fee9272ca9b57b292e2c23df1c69 — unpinned in the reference, no manifest)
// is NOT under /root/reference, so this follows bx's published algorithm
// (bx/src/math.cpp: cos/sin/floor, mtxSRT; bx/include/bx/inline/math.inl: mtxMul,
// vec4MulMtx) with the constants and operation order that SURVEY.md §8 a-3/a-4
// recovered from the reference's committed MSVC objects
// (build/SandboxCity.dir/RelWithDebInfo/Transform.obj: vec4MulMtx body;
//  build/bin/RelWithDebInfo/SandboxCity.exe: mtxSRT/cos/floor).
//
// PARITY STATUS: the reference holds no tests or golden vectors for this path;
// the known-answer vectors in tests/golden/bx_kat.json come from SURVEY.md §8
// (a restatement made from the disassembly).  Strictly: "parity unpinned".
// What does pin it: oracle/tools/check_bx_order.py executes the reference's COMPILED
// vec4MulMtx, mtxSRT, cos, floor and mtxInverse symbolically (from the committed
// Transform.obj / SandboxCity.exe, nothing is run) and finds their expression trees
// identical to the formulas below — operation order, association, constants, branches.
//
// Must be compiled with -ffp-contract=off (scalar mulss/addss, no FMA, as the
// reference's /fp:precise MSVC build).
#pragma once

#include <cstdint>
#include <cstring>

namespace orc {
namespace bxm {

inline float from_bits(uint32_t u)
{
    float f;
    std::memcpy(&f, &u, sizeof f);
    return f;
}

// bx::kPiHalf and bx::kInvPi as binary32.
inline constexpr float kPiHalf = 1.5707963267948966f; // 0x3fc90fdb
inline constexpr float kInvPi = 0.31830988618379067f; // 0x3ea2f983

// bx::trunc / bx::fract / bx::floor (math.inl): int-cast based.
inline float trunc_(float a) { return float(int32_t(a)); }
inline float fract_(float a) { return a - trunc_(a); }
inline float floor_(float a)
{
    if (a < 0.0f) {
        const float fr = fract_(-a);
        const float result = -a - fr;
        return -(0.0f != fr ? result + 1.0f : result);
    }
    return a - fract_(a);
}

// bx::cos (math.cpp): quadrant reduction by pi/2, two degree-10 even polynomials.
inline float cos_(float a)
{
    const float scaled = (a * 2.0f) * kInvPi;
    const float real = floor_(scaled);
    const float xx = a - real * kPiHalf;
    const int32_t quadrant = int32_t(real) & 3;

    float c0, c2, c4, c6, c8, c10;
    if (quadrant == 0 || quadrant == 2) {
        c0 = 1.0f;
        c2 = -0.5f;
        c4 = from_bits(0x3d2aaaa4u);
        c6 = from_bits(0xbab60981u);
        c8 = from_bits(0x37cfab9cu);
        c10 = from_bits(0xb48b634du);
    } else {
        c0 = xx;
        c2 = from_bits(0xbe2aaaabu);
        c4 = from_bits(0x3c088898u);
        c6 = from_bits(0xb9501096u);
        c8 = from_bits(0x363938a8u);
        c10 = from_bits(0xb2d70013u);
    }

    const float xsq = xx * xx;
    float acc = c10 * xsq + c8;
    acc = acc * xsq + c6;
    acc = acc * xsq + c4;
    acc = acc * xsq + c2;
    acc = acc * xsq + 1.0f;
    const float result = acc * c0;
    return (quadrant == 1 || quadrant == 2) ? -result : result;
}

// bx::sin: cos shifted by pi/2.
inline float sin_(float a) { return cos_(a - kPiHalf); }

inline void mtxIdentity(float* m)
{
    for (int i = 0; i < 16; ++i) m[i] = 0.0f;
    m[0] = m[5] = m[10] = m[15] = 1.0f;
}

// bx::mtxSRT(result, sx,sy,sz, ax,ay,az, tx,ty,tz): row-major, row-vector, translation in 12..14.
inline void mtxSRT(float* out, float sx_, float sy_, float sz_, float ax, float ay, float az, float tx,
                   float ty, float tz)
{
    const float sx = sin_(ax);
    const float cx = cos_(ax);
    const float sy = sin_(ay);
    const float cy = cos_(ay);
    const float sz = sin_(az);
    const float cz = cos_(az);

    const float sxsz = sx * sz;
    const float cycz = cy * cz;

    out[0] = sx_ * (cycz - sxsz * sy);
    out[1] = sx_ * -cx * sz;
    out[2] = sx_ * (sxsz * cy + cz * sy);
    out[3] = 0.0f;

    out[4] = sy_ * (cz * sx * sy + sz * cy);
    out[5] = sy_ * cx * cz;
    out[6] = sy_ * (sz * sy - cycz * sx);
    out[7] = 0.0f;

    out[8] = sz_ * -cx * sy;
    out[9] = sz_ * sx;
    out[10] = sz_ * cx * cy;
    out[11] = 0.0f;

    out[12] = tx;
    out[13] = ty;
    out[14] = tz;
    out[15] = 1.0f;
}

// bx::vec4MulMtx: out[j] = ((v0*m[j] + v1*m[4+j]) + v2*m[8+j]) + v3*m[12+j]
inline void vec4MulMtx(float* out, const float* v, const float* m)
{
    for (int j = 0; j < 4; ++j) {
        out[j] = ((v[0] * m[j] + v[1] * m[4 + j]) + v[2] * m[8 + j]) + v[3] * m[12 + j];
    }
}

// bx::mtxMul(result, a, b) = a·b, one vec4MulMtx per row of a.
inline void mtxMul(float* out, const float* a, const float* b)
{
    vec4MulMtx(out + 0, a + 0, b);
    vec4MulMtx(out + 4, a + 4, b);
    vec4MulMtx(out + 8, a + 8, b);
    vec4MulMtx(out + 12, a + 12, b);
}

// bx::mtxInverse (bx/src/math.cpp): adjugate / determinant, cofactors expanded along the first row of each minor.
// Used by the reference for the per-mesh normal matrix (src/render/Renderer.cpp:413-416, 633-636).
// Operation order checked against the compiled function in the reference's SandboxCity.exe
// (oracle/tools/check_bx_order.py: all 16 outputs identical as expression trees).
inline void mtxInverse(float* r, const float* a)
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

    r[0] = +(yy * (zz * ww - wz * zw) - yz * (zy * ww - wy * zw) + yw * (zy * wz - wy * zz)) * invDet;
    r[1] = -(xy * (zz * ww - wz * zw) - xz * (zy * ww - wy * zw) + xw * (zy * wz - wy * zz)) * invDet;
    r[2] = +(xy * (yz * ww - wz * yw) - xz * (yy * ww - wy * yw) + xw * (yy * wz - wy * yz)) * invDet;
    r[3] = -(xy * (yz * zw - zz * yw) - xz * (yy * zw - zy * yw) + xw * (yy * zz - zy * yz)) * invDet;

    r[4] = -(yx * (zz * ww - wz * zw) - yz * (zx * ww - wx * zw) + yw * (zx * wz - wx * zz)) * invDet;
    r[5] = +(xx * (zz * ww - wz * zw) - xz * (zx * ww - wx * zw) + xw * (zx * wz - wx * zz)) * invDet;
    r[6] = -(xx * (yz * ww - wz * yw) - xz * (yx * ww - wx * yw) + xw * (yx * wz - wx * yz)) * invDet;
    r[7] = +(xx * (yz * zw - zz * yw) - xz * (yx * zw - zx * yw) + xw * (yx * zz - zx * yz)) * invDet;

    r[8] = +(yx * (zy * ww - wy * zw) - yy * (zx * ww - wx * zw) + yw * (zx * wy - wx * zy)) * invDet;
    r[9] = -(xx * (zy * ww - wy * zw) - xy * (zx * ww - wx * zw) + xw * (zx * wy - wx * zy)) * invDet;
    r[10] = +(xx * (yy * ww - wy * yw) - xy * (yx * ww - wx * yw) + xw * (yx * wy - wx * yy)) * invDet;
    r[11] = -(xx * (yy * zw - zy * yw) - xy * (yx * zw - zx * yw) + xw * (yx * zy - zx * yy)) * invDet;

    r[12] = -(yx * (zy * wz - wy * zz) - yy * (zx * wz - wx * zz) + yz * (zx * wy - wx * zy)) * invDet;
    r[13] = +(xx * (zy * wz - wy * zz) - xy * (zx * wz - wx * zz) + xz * (zx * wy - wx * zy)) * invDet;
    r[14] = -(xx * (yy * wz - wy * yz) - xy * (yx * wz - wx * yz) + xz * (yx * wy - wx * yy)) * invDet;
    r[15] = +(xx * (yy * zz - zy * yz) - xy * (yx * zz - zx * yz) + xz * (yx * zy - zx * yy)) * invDet;
}

// bx::mtxTranspose
inline void mtxTranspose(float* r, const float* a)
{
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r[4 * i + j] = a[4 * j + i];
}

// normalMtx of Renderer::BeginFrame: transpose(inverse(world))
inline void normalMatrix(float* r, const float* world)
{
    float inv[16];
    mtxInverse(inv, world);
    mtxTranspose(r, inv);
}

} // namespace bxm
} // namespace orc
