/*
 * bge_detmath.h — deterministic single-precision sin/cos/asin/atan2.
 *
 * Why this exists.  The reference's physics half calls the platform libm through
 * Bullet (btSin/btCos/btAsin/btAtan2 → sinf/cosf/asinf/atan2f; reference call
 * sites src/physics/PhysicsSystem.cpp:40-45 setEulerZYX, :937-947 getEulerZYX).
 * No two libms (MSVC UCRT, glibc, ROCm ocml) agree to the last bit, so a GPU path
 * that called ocml could only ever be "close" to a CPU checker.  These routines
 * use nothing but IEEE-754 binary32 + - * / sqrt, comparisons and in-range
 * float<->int conversions, so the SAME source gives bit-identical results on the
 * host (g++ -ffp-contract=off) and on gfx950 (hipcc -ffp-contract=off, default
 * correctly-rounded divide/sqrt).  Accuracy is within a few ulp of the true
 * function (classic Cody–Waite reduction + minimax polynomials, the scheme
 * published with the Cephes single-precision library); tests/test_detmath.py
 * bounds the distance to the platform libm.
 *
 * Domain: |x| < 8192 for sin/cos (angles in the engine are a few radians).
 */
#ifndef BGE_DETMATH_H
#define BGE_DETMATH_H

#if defined(__HIPCC__)
#define BGE_HD __host__ __device__ __forceinline__
#else
#define BGE_HD inline
#endif

#if defined(__HIPCC__)
#define BGE_SQRTF(x) __builtin_sqrtf(x)
#else
#include <math.h>
#define BGE_SQRTF(x) sqrtf(x)
#endif

#define BGE_PI_F 3.14159265358979323846f
#define BGE_HALF_PI_F 1.57079632679489661923f
#define BGE_QUARTER_PI_F 0.78539816339744830962f

/* Octant reduction shared by sin and cos: returns r in [-pi/4, pi/4] and the
 * octant index j (already rounded up to even, masked to 0..7). */
BGE_HD float bge_det_reduce_(float ax, int* octant)
{
    const float four_over_pi = 1.27323954473516f;
    int j = (int)(four_over_pi * ax);
    float y = (float)j;
    if (j & 1) {
        j += 1;
        y += 1.0f;
    }
    *octant = j & 7;
    /* pi/4 split in three parts so that y*part is exact for small y */
    const float p1 = 0.78515625f;
    const float p2 = 2.4187564849853515625e-4f;
    const float p3 = 3.77489497744594108e-8f;
    return ((ax - y * p1) - y * p2) - y * p3;
}

BGE_HD float bge_det_sin_poly_(float r)
{
    const float z = r * r;
    return (((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r) + r;
}

BGE_HD float bge_det_cos_poly_(float r)
{
    const float z = r * r;
    float y = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    y -= 0.5f * z;
    y += 1.0f;
    return y;
}

BGE_HD float bge_det_sinf(float x)
{
    int neg = x < 0.0f;
    const float ax = neg ? -x : x;
    int j;
    const float r = bge_det_reduce_(ax, &j);
    if (j > 3) {
        j -= 4;
        neg = !neg;
    }
    const float y = (j == 1 || j == 2) ? bge_det_cos_poly_(r) : bge_det_sin_poly_(r);
    return neg ? -y : y;
}

BGE_HD float bge_det_cosf(float x)
{
    const float ax = x < 0.0f ? -x : x;
    int j;
    const float r = bge_det_reduce_(ax, &j);
    int neg = 0;
    if (j > 3) {
        j -= 4;
        neg = 1;
    }
    if (j > 1) {
        neg = !neg;
    }
    const float y = (j == 1 || j == 2) ? bge_det_sin_poly_(r) : bge_det_cos_poly_(r);
    return neg ? -y : y;
}

/* asin on [-1,1]; callers clamp first (Bullet's btAsin clamps). */
BGE_HD float bge_det_asinf(float x)
{
    const int neg = x < 0.0f;
    const float a = neg ? -x : x;
    float res;
    if (a < 1.0e-4f) {
        res = a;
    } else {
        float t, z;
        const int upper = a > 0.5f;
        if (upper) {
            z = 0.5f * (1.0f - a);
            t = BGE_SQRTF(z);
        } else {
            t = a;
            z = a * a;
        }
        res = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
               + 1.6666752422e-1f) * z * t + t;
        if (upper) {
            res = res + res;
            res = BGE_HALF_PI_F - res;
        }
    }
    return neg ? -res : res;
}

BGE_HD float bge_det_atanf(float x)
{
    const int neg = x < 0.0f;
    float t = neg ? -x : x;
    float base;
    if (t > 2.414213562373095f) { /* tan(3pi/8) */
        base = BGE_HALF_PI_F;
        t = -(1.0f / t);
    } else if (t > 0.4142135623730950f) { /* tan(pi/8) */
        base = BGE_QUARTER_PI_F;
        t = (t - 1.0f) / (t + 1.0f);
    } else {
        base = 0.0f;
    }
    const float z = t * t;
    const float y = base + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z
                             - 3.33329491539e-1f) * z * t + t);
    return neg ? -y : y;
}

/* atan2 with the quadrant rules of the C library, except that signed zeros are
 * not distinguished (atan2(-0,-1) = +pi here).  The quotient and the range
 * reduction of the arctangent share ONE division: |y| / |x| is compared with
 * tan(pi/8), tan(3 pi/8) as |y| against those multiples of |x|, and the reduced
 * argument is -|x| / |y|, (|y| - |x|) / (|y| + |x|) or |y| / |x| (round 3: it was
 * y / x followed by -1 / t or (t - 1) / (t + 1), three IEEE divisions in a
 * predicated instruction stream; the values move by an ulp here and there, which
 * no libm pins). */
BGE_HD float bge_det_atan2f(float y, float x)
{
    if (x == 0.0f) {
        if (y < 0.0f) return -BGE_HALF_PI_F;
        if (y == 0.0f) return 0.0f;
        return BGE_HALF_PI_F;
    }
    if (y == 0.0f) {
        return x < 0.0f ? BGE_PI_F : 0.0f;
    }
    float w = 0.0f;
    if (x < 0.0f) {
        w = y < 0.0f ? -BGE_PI_F : BGE_PI_F;
    }
    const float ax = x < 0.0f ? -x : x;
    const float ay = y < 0.0f ? -y : y;
    float base, num, den;
    if (ay > 2.414213562373095f * ax) { /* tan(3pi/8) */
        base = BGE_HALF_PI_F;
        num = -ax;
        den = ay;
    } else if (ay > 0.4142135623730950f * ax) { /* tan(pi/8) */
        base = BGE_QUARTER_PI_F;
        num = ay - ax;
        den = ay + ax;
    } else {
        base = 0.0f;
        num = ay;
        den = ax;
    }
    const float t = num / den;
    const float z = t * t;
    const float r = base + ((((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z
                             - 3.33329491539e-1f) * z * t + t);
    const int neg = (y < 0.0f) != (x < 0.0f);
    return w + (neg ? -r : r);
}

#endif /* BGE_DETMATH_H */
