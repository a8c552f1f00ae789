// bge_boxbox_device.hpp — Bullet's box-box narrowphase on the device: btBoxBoxDetector (dBoxBox2: 15-axis separating-axis
// test, reference / incident face clipping, the four-point cull) for the contacts of a Dynamic box with the Static / Kinematic
// box colliders of the scene (the reference's demo.json "Ground" is such a box,
// assets/scenes/demo.json:67-91, src/physics/PhysicsSystem.cpp:421-474).  Operation for operation what oracle/boxbox_ref.h
// restates from Bullet's published btBoxBoxDetector.cpp (that header says what is pinned and what is a specification choice);
// the library is compiled with -ffp-contract=off, divisions and square roots are IEEE, atan2 is bge_detmath.h's.
// One thread runs one pair; the small arrays are indexed at run time and live in scratch memory — this path is sized for
// correctness, it runs for the few bodies of a scene that touch a static box (bge_contact.hip routes them).
#pragma once

#include <hip/hip_runtime.h>

#include "bge_device_math.hpp"

namespace bge {
namespace boxbox {

using dev::F3;
using dev::M3;

struct Out {
    int n;
    F3 normalOnB; // -normal: the same for every point of a call
    F3 point[4];  // pointInWorld (on B)
    float depth[4];
};

// dDOTpq(a, b, p, q)
__device__ __forceinline__ float dotpq(const float* a, int p, const float* b, int q) { return a[0] * b[0] + a[p] * b[q] + a[2 * p] * b[2 * q]; }

__device__ inline void to_m34(const M3& b, float* r)
{
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        r[0 + 4 * j] = b.m[j][0];
        r[1 + 4 * j] = b.m[j][1];
        r[2 + 4 * j] = b.m[j][2];
        r[3 + 4 * j] = 0.0f;
    }
}

__device__ inline int intersect_rect_quad2(const float h[2], const float p[8], float ret[16])
{
    int nq = 4, nr = 0;
    float buffer[16];
    const float* q = p;
    float* r = ret;
    bool done = false;
    for (int dir = 0; dir <= 1 && !done; ++dir) {
        for (int sign = -1; sign <= 1 && !done; sign += 2) {
            const float fs = static_cast<float>(sign);
            const float* pq = q;
            float* pr = r;
            nr = 0;
            for (int i = nq; i > 0; --i) {
                if (fs * pq[dir] < h[dir]) {
                    pr[0] = pq[0];
                    pr[1] = pq[1];
                    pr += 2;
                    nr++;
                    if (nr & 8) {
                        q = r;
                        done = true;
                        break;
                    }
                }
                const float* nextq = (i > 1) ? pq + 2 : q;
                if ((fs * pq[dir] < h[dir]) ^ (fs * nextq[dir] < h[dir])) {
                    pr[1 - dir] = pq[1 - dir] + (nextq[1 - dir] - pq[1 - dir]) / (nextq[dir] - pq[dir]) * (fs * h[dir] - pq[dir]);
                    pr[dir] = fs * h[dir];
                    pr += 2;
                    nr++;
                    if (nr & 8) {
                        q = r;
                        done = true;
                        break;
                    }
                }
                pq += 2;
            }
            if (done) break;
            q = r;
            r = (q == ret) ? buffer : ret;
            nq = nr;
        }
    }
    if (q != ret) {
        for (int k = 0; k < nr * 2; ++k) ret[k] = q[k];
    }
    return nr;
}

__device__ inline void cull_points2(int n, const float p[], int m, int i0, int iret[])
{
    constexpr float kPi = 3.14159265f;
    float a, cx, cy, q;
    if (n == 1) {
        cx = p[0];
        cy = p[1];
    } else if (n == 2) {
        cx = 0.5f * (p[0] + p[2]);
        cy = 0.5f * (p[1] + p[3]);
    } else {
        a = 0.0f;
        cx = 0.0f;
        cy = 0.0f;
        for (int i = 0; i < n - 1; ++i) {
            q = p[i * 2] * p[i * 2 + 3] - p[i * 2 + 2] * p[i * 2 + 1];
            a += q;
            cx += q * (p[i * 2] + p[i * 2 + 2]);
            cy += q * (p[i * 2 + 1] + p[i * 2 + 3]);
        }
        q = p[n * 2 - 2] * p[1] - p[0] * p[n * 2 - 1];
        if (__builtin_fabsf(a + q) > dev::kBtEpsilon) {
            a = 0.3333333432674408f / (a + q); // (1.f / (3 * (a + q)) in the source: the reference's compiled code — MSVC /fp:fast — divides the constant 0x3eaaaaab)
        } else {
            a = 1.0e18f;
        }
        cx = a * (cx + q * (p[n * 2 - 2] + p[0]));
        cy = a * (cy + q * (p[n * 2 - 1] + p[1]));
    }
    float A[8];
    for (int i = 0; i < n; ++i) A[i] = bge_det_atan2f(p[i * 2 + 1] - cy, p[i * 2] - cx);
    int avail[8];
    for (int i = 0; i < n; ++i) avail[i] = 1;
    avail[i0] = 0;
    iret[0] = i0;
    int at = 1;
    for (int j = 1; j < m; ++j) {
        a = static_cast<float>(j) * (2 * kPi / m) + A[i0];
        if (a > kPi) a -= 2 * kPi;
        float maxdiff = 1e9f, diff;
        iret[at] = i0;
        for (int i = 0; i < n; ++i) {
            if (avail[i]) {
                diff = __builtin_fabsf(A[i] - a);
                if (diff > kPi) diff = 2 * kPi - diff;
                if (diff < maxdiff) {
                    maxdiff = diff;
                    iret[at] = i;
                }
            }
        }
        avail[iret[at]] = 0;
        at++;
    }
}

// dBoxBox2 (maxc = 4) + btBoxBoxDetector::getClosestPoints: box 1 = (p1, basis1, half1 with margin), box 2 likewise
__device__ inline int box_box(const F3& p1v, const M3& basis1, const F3& half1, const F3& p2v, const M3& basis2, const F3& half2, Out& out)
{
    out.n = 0;
    const float fudge_factor = 1.05f;
    float R1[12], R2[12];
    to_m34(basis1, R1);
    to_m34(basis2, R2);
    const float side1[3] = {half1.x * 2.0f, half1.y * 2.0f, half1.z * 2.0f}, side2[3] = {half2.x * 2.0f, half2.y * 2.0f, half2.z * 2.0f};
    const float p1[3] = {p1v.x, p1v.y, p1v.z}, p2[3] = {p2v.x, p2v.y, p2v.z};
    float p[3], pp[3], normalC[3] = {0.0f, 0.0f, 0.0f};
    const float* normalR = nullptr;
    float A[3], B[3], s, s2, l;
    int invert_normal, code;
    float normal[3];

    p[0] = p2[0] - p1[0];
    p[1] = p2[1] - p1[1];
    p[2] = p2[2] - p1[2];
    pp[0] = dotpq(R1 + 0, 4, p, 1);
    pp[1] = dotpq(R1 + 1, 4, p, 1);
    pp[2] = dotpq(R1 + 2, 4, p, 1);
    A[0] = side1[0] * 0.5f;
    A[1] = side1[1] * 0.5f;
    A[2] = side1[2] * 0.5f;
    B[0] = side2[0] * 0.5f;
    B[1] = side2[1] * 0.5f;
    B[2] = side2[2] * 0.5f;
    const float R11 = dotpq(R1 + 0, 4, R2 + 0, 4), R12 = dotpq(R1 + 0, 4, R2 + 1, 4), R13 = dotpq(R1 + 0, 4, R2 + 2, 4);
    const float R21 = dotpq(R1 + 1, 4, R2 + 0, 4), R22 = dotpq(R1 + 1, 4, R2 + 1, 4), R23 = dotpq(R1 + 1, 4, R2 + 2, 4);
    const float R31 = dotpq(R1 + 2, 4, R2 + 0, 4), R32 = dotpq(R1 + 2, 4, R2 + 1, 4), R33 = dotpq(R1 + 2, 4, R2 + 2, 4);
    float Q11 = __builtin_fabsf(R11), Q12 = __builtin_fabsf(R12), Q13 = __builtin_fabsf(R13);
    float Q21 = __builtin_fabsf(R21), Q22 = __builtin_fabsf(R22), Q23 = __builtin_fabsf(R23);
    float Q31 = __builtin_fabsf(R31), Q32 = __builtin_fabsf(R32), Q33 = __builtin_fabsf(R33);

    s = -3.402823466e+38f;
    invert_normal = 0;
    code = 0;
// (the four-term sums of expr2 are written (t0 + t1) + (t2 + t3): the association of the reference's COMPILED dBoxBox2 — Bullet is
    //  built with MSVC /fp:fast — read off the exe by oracle/tools/check_boxbox_order.py for all fifteen axes)
#define BGE_TST(expr1, expr2, norm, cc)        \
    s2 = __builtin_fabsf(expr1) - (expr2);     \
    if (s2 > 0) return 0;                      \
    if (s2 > s) {                              \
        s = s2;                                \
        normalR = norm;                        \
        invert_normal = ((expr1) < 0);         \
        code = (cc);                           \
    }
    BGE_TST(pp[0], ((A[0] + B[0] * Q11) + (B[1] * Q12 + B[2] * Q13)), R1 + 0, 1);
    BGE_TST(pp[1], ((A[1] + B[0] * Q21) + (B[1] * Q22 + B[2] * Q23)), R1 + 1, 2);
    BGE_TST(pp[2], ((A[2] + B[0] * Q31) + (B[1] * Q32 + B[2] * Q33)), R1 + 2, 3);
    BGE_TST(dotpq(R2 + 0, 4, p, 1), ((A[0] * Q11 + A[1] * Q21) + (A[2] * Q31 + B[0])), R2 + 0, 4);
    BGE_TST(dotpq(R2 + 1, 4, p, 1), ((A[0] * Q12 + A[1] * Q22) + (A[2] * Q32 + B[1])), R2 + 1, 5);
    BGE_TST(dotpq(R2 + 2, 4, p, 1), ((A[0] * Q13 + A[1] * Q23) + (A[2] * Q33 + B[2])), R2 + 2, 6);
#undef BGE_TST
// (edge axes: the reference's compiled code — MSVC /fp:fast — forms ONE reciprocal 1 / l and multiplies: s2 * (1 / l), n * (1 / l), with
//  (-1 / l) * r for the negated component, which is the same bits; oracle/tools/check_boxbox_order.py reads it off the exe)
#define BGE_TST(expr1, expr2, n1, n2, n3, cc)                            \
    s2 = __builtin_fabsf(expr1) - (expr2);                                          \
    if (s2 > dev::kBtEpsilon) return 0;                                              \
    l = __builtin_sqrtf((n1) * (n1) + (n2) * (n2) + (n3) * (n3));                   \
    if (l > dev::kBtEpsilon) {                                                       \
        const float il = 1.0f / l;                                       \
        s2 *= il;                                                        \
        if (s2 * fudge_factor > s) {                                     \
            s = s2;                                                      \
            normalR = nullptr;                                           \
            normalC[0] = (n1) * il;                                      \
            normalC[1] = (n2) * il;                                      \
            normalC[2] = (n3) * il;                                      \
            invert_normal = ((expr1) < 0);                               \
            code = (cc);                                                 \
        }                                                                \
    }
    const float fudge2 = 1.0e-5f;
    Q11 += fudge2;
    Q12 += fudge2;
    Q13 += fudge2;
    Q21 += fudge2;
    Q22 += fudge2;
    Q23 += fudge2;
    Q31 += fudge2;
    Q32 += fudge2;
    Q33 += fudge2;
    BGE_TST(pp[2] * R21 - pp[1] * R31, ((A[1] * Q31 + A[2] * Q21) + (B[1] * Q13 + B[2] * Q12)), 0.0f, -R31, R21, 7);
    BGE_TST(pp[2] * R22 - pp[1] * R32, ((A[1] * Q32 + A[2] * Q22) + (B[0] * Q13 + B[2] * Q11)), 0.0f, -R32, R22, 8);
    BGE_TST(pp[2] * R23 - pp[1] * R33, ((A[1] * Q33 + A[2] * Q23) + (B[0] * Q12 + B[1] * Q11)), 0.0f, -R33, R23, 9);
    BGE_TST(pp[0] * R31 - pp[2] * R11, ((A[0] * Q31 + A[2] * Q11) + (B[1] * Q23 + B[2] * Q22)), R31, 0.0f, -R11, 10);
    BGE_TST(pp[0] * R32 - pp[2] * R12, ((A[0] * Q32 + A[2] * Q12) + (B[0] * Q23 + B[2] * Q21)), R32, 0.0f, -R12, 11);
    BGE_TST(pp[0] * R33 - pp[2] * R13, ((A[0] * Q33 + A[2] * Q13) + (B[0] * Q22 + B[1] * Q21)), R33, 0.0f, -R13, 12);
    BGE_TST(pp[1] * R11 - pp[0] * R21, ((A[0] * Q21 + A[1] * Q11) + (B[1] * Q33 + B[2] * Q32)), -R21, R11, 0.0f, 13);
    BGE_TST(pp[1] * R12 - pp[0] * R22, ((A[0] * Q22 + A[1] * Q12) + (B[0] * Q33 + B[2] * Q31)), -R22, R12, 0.0f, 14);
    BGE_TST(pp[1] * R13 - pp[0] * R23, ((A[0] * Q23 + A[1] * Q13) + (B[0] * Q32 + B[1] * Q31)), -R23, R13, 0.0f, 15);
#undef BGE_TST
    if (!code) return 0;

    if (normalR) {
        normal[0] = normalR[0];
        normal[1] = normalR[4];
        normal[2] = normalR[8];
    } else {
        normal[0] = dotpq(R1 + 0, 1, normalC, 1);
        normal[1] = dotpq(R1 + 4, 1, normalC, 1);
        normal[2] = dotpq(R1 + 8, 1, normalC, 1);
    }
    if (invert_normal) {
        normal[0] = -normal[0];
        normal[1] = -normal[1];
        normal[2] = -normal[2];
    }
    const float depth = -s;
    out.normalOnB = F3{-normal[0], -normal[1], -normal[2]};

    if (code > 6) {
        float pa[3], pb[3];
        for (int i = 0; i < 3; ++i) pa[i] = p1[i];
        for (int j = 0; j < 3; ++j) {
            const float sign = (dotpq(normal, 1, R1 + j, 4) > 0) ? 1.0f : -1.0f;
            for (int i = 0; i < 3; ++i) pa[i] += sign * A[j] * R1[i * 4 + j];
        }
        for (int i = 0; i < 3; ++i) pb[i] = p2[i];
        for (int j = 0; j < 3; ++j) {
            const float sign = (dotpq(normal, 1, R2 + j, 4) > 0) ? -1.0f : 1.0f;
            for (int i = 0; i < 3; ++i) pb[i] += sign * B[j] * R2[i * 4 + j];
        }
        float ua[3], ub[3];
        for (int i = 0; i < 3; ++i) ua[i] = R1[(code - 7) / 3 + i * 4];
        for (int i = 0; i < 3; ++i) ub[i] = R2[(code - 7) % 3 + i * 4];
        // dLineClosestApproach
        float alpha, beta;
        {
            const float d3[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
            const float uaub = dotpq(ua, 1, ub, 1);
            const float q1 = dotpq(ua, 1, d3, 1);
            const float q2 = -dotpq(ub, 1, d3, 1);
            float d = 1.0f - uaub * uaub;
            if (d <= 0.0001f) {
                alpha = 0.0f;
                beta = 0.0f;
            } else {
                // (d = 1 / d and two products in the source: the reference's compiled code — MSVC /fp:fast — divides by d once,
                //  oracle/boxbox_ref.h LineClosestApproach)
                alpha = (q1 + uaub * q2) / d;
                beta = (uaub * q1 + q2) / d;
            }
        }
        for (int i = 0; i < 3; ++i) pa[i] += ua[i] * alpha;
        for (int i = 0; i < 3; ++i) pb[i] += ub[i] * beta;
        out.point[0] = F3{pb[0], pb[1], pb[2]};
        out.depth[0] = -depth;
        out.n = 1;
        return 1;
    }

    const float *Ra, *Rb, *pa, *pb, *Sa, *Sb;
    if (code <= 3) {
        Ra = R1;
        Rb = R2;
        pa = p1;
        pb = p2;
        Sa = A;
        Sb = B;
    } else {
        Ra = R2;
        Rb = R1;
        pa = p2;
        pb = p1;
        Sa = B;
        Sb = A;
    }
    float normal2[3], nr[3], anr[3];
    if (code <= 3) {
        normal2[0] = normal[0];
        normal2[1] = normal[1];
        normal2[2] = normal[2];
    } else {
        normal2[0] = -normal[0];
        normal2[1] = -normal[1];
        normal2[2] = -normal[2];
    }
    nr[0] = dotpq(Rb + 0, 4, normal2, 1);
    nr[1] = dotpq(Rb + 1, 4, normal2, 1);
    nr[2] = dotpq(Rb + 2, 4, normal2, 1);
    anr[0] = __builtin_fabsf(nr[0]);
    anr[1] = __builtin_fabsf(nr[1]);
    anr[2] = __builtin_fabsf(nr[2]);
    int lanr, a1, a2;
    if (anr[1] > anr[0]) {
        if (anr[1] > anr[2]) {
            a1 = 0;
            lanr = 1;
            a2 = 2;
        } else {
            a1 = 0;
            a2 = 1;
            lanr = 2;
        }
    } else {
        if (anr[0] > anr[2]) {
            lanr = 0;
            a1 = 1;
            a2 = 2;
        } else {
            a1 = 0;
            a2 = 1;
            lanr = 2;
        }
    }
    float center[3];
    if (nr[lanr] < 0) {
        for (int i = 0; i < 3; ++i) center[i] = pb[i] - pa[i] + Sb[lanr] * Rb[i * 4 + lanr];
    } else {
        for (int i = 0; i < 3; ++i) center[i] = pb[i] - pa[i] - Sb[lanr] * Rb[i * 4 + lanr];
    }
    int codeN, code1, code2;
    if (code <= 3)
        codeN = code - 1;
    else
        codeN = code - 4;
    if (codeN == 0) {
        code1 = 1;
        code2 = 2;
    } else if (codeN == 1) {
        code1 = 0;
        code2 = 2;
    } else {
        code1 = 0;
        code2 = 1;
    }
    float quad[8];
    float c1, c2, m11, m12, m21, m22;
    c1 = dotpq(center, 1, Ra + code1, 4);
    c2 = dotpq(center, 1, Ra + code2, 4);
    m11 = dotpq(Ra + code1, 4, Rb + a1, 4);
    m12 = dotpq(Ra + code1, 4, Rb + a2, 4);
    m21 = dotpq(Ra + code2, 4, Rb + a1, 4);
    m22 = dotpq(Ra + code2, 4, Rb + a2, 4);
    {
        const float k1 = m11 * Sb[a1];
        const float k2 = m21 * Sb[a1];
        const float k3 = m12 * Sb[a2];
        const float k4 = m22 * Sb[a2];
        quad[0] = c1 - k1 - k3;
        quad[1] = c2 - k2 - k4;
        quad[2] = c1 - k1 + k3;
        quad[3] = c2 - k2 + k4;
        quad[4] = c1 + k1 + k3;
        quad[5] = c2 + k2 + k4;
        quad[6] = c1 + k1 - k3;
        quad[7] = c2 + k2 - k4;
    }
    float rect[2];
    rect[0] = Sa[code1];
    rect[1] = Sa[code2];
    float ret[16];
    const int n = intersect_rect_quad2(rect, quad, ret);
    if (n < 1) return 0;
    float point[3 * 8];
    float dep[8];
    const float det1 = 1.0f / (m11 * m22 - m12 * m21);
    m11 *= det1;
    m12 *= det1;
    m21 *= det1;
    m22 *= det1;
    int cnum = 0;
    for (int j = 0; j < n; ++j) {
        const float k1 = m22 * (ret[j * 2] - c1) - m12 * (ret[j * 2 + 1] - c2);
        const float k2 = -m21 * (ret[j * 2] - c1) + m11 * (ret[j * 2 + 1] - c2);
        for (int i = 0; i < 3; ++i) point[cnum * 3 + i] = center[i] + k1 * Rb[i * 4 + a1] + k2 * Rb[i * 4 + a2];
        dep[cnum] = Sa[codeN] - dotpq(normal2, 1, point + cnum * 3, 1);
        if (dep[cnum] >= 0) {
            ret[cnum * 2] = ret[j * 2];
            ret[cnum * 2 + 1] = ret[j * 2 + 1];
            cnum++;
        }
    }
    if (cnum < 1) return 0;
    int maxc = 4;
    if (maxc > cnum) maxc = cnum;
    if (maxc < 1) maxc = 1;
    if (cnum <= maxc) {
        for (int j = 0; j < cnum; ++j) {
            float wv[3];
            if (code < 4) {
                for (int i = 0; i < 3; ++i) wv[i] = point[j * 3 + i] + pa[i];
            } else {
                for (int i = 0; i < 3; ++i) wv[i] = point[j * 3 + i] + pa[i] - normal[i] * dep[j];
            }
            out.point[j] = F3{wv[0], wv[1], wv[2]};
            out.depth[j] = -dep[j];
        }
        out.n = cnum;
    } else {
        int i1 = 0;
        float maxdepth = dep[0];
        for (int i = 1; i < cnum; ++i) {
            if (dep[i] > maxdepth) {
                maxdepth = dep[i];
                i1 = i;
            }
        }
        int iret[8];
        cull_points2(cnum, ret, maxc, i1, iret);
        for (int j = 0; j < maxc; ++j) {
            float wv[3];
            for (int i = 0; i < 3; ++i) wv[i] = point[iret[j] * 3 + i] + pa[i];
            if (code < 4) {
                out.point[j] = F3{wv[0], wv[1], wv[2]};
            } else {
                out.point[j] = F3{wv[0] - normal[0] * dep[iret[j]], wv[1] - normal[1] * dep[iret[j]], wv[2] - normal[2] * dep[iret[j]]};
            }
            out.depth[j] = -dep[iret[j]];
        }
        cnum = maxc;
        out.n = maxc;
    }
    return cnum;
}

} // namespace boxbox
} // namespace bge
