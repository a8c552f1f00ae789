// oracle/boxbox_ref.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of Bullet's box-box narrowphase for the contacts of a Dynamic box with Static / Kinematic box colliders
// (VERDICT r02 item 4).  The reference creates every RigidBody, Static ones included, as a Bullet
// body with its btBoxShape (src/physics/PhysicsSystem.cpp:421-474, CreateShape :686-707) and steps with
// btDefaultCollisionConfiguration / btCollisionDispatcher (:122-128), whose algorithm for two boxes is
// btBoxBoxCollisionAlgorithm; assets/scenes/demo.json:67-91's "Ground" is such a box.
//
// Bullet's source is not under /root/reference; restated from its PUBLISHED source, function by function, and then corrected to
// the COMPILED code of the reference's exe wherever MSVC /fp:fast made something else of it (PARITY STATUS below):
//   BulletCollision/CollisionDispatch/btBoxBoxDetector.cpp   dBoxBox2 (15-axis separating-axis test, reference / incident face,
//                                                            intersectRectQuad2, the four-point cull cullPoints2),
//                                                            dLineClosestApproach, btBoxBoxDetector::getClosestPoints
//   BulletCollision/CollisionDispatch/btBoxBoxCollisionAlgorithm.cpp   processCollision (USE_PERSISTENT_CONTACTS: the detector
//                                                            adds points to the persistent manifold, then refreshContactPoints)
//   BulletCollision/CollisionDispatch/btManifoldResult.cpp   addContactPoint (breaking threshold, localA / localB, the cache
//                                                            entry, combined friction = product clamped to +-10, combined
//                                                            restitution = product)
//   BulletCollision/NarrowPhaseCollision/btPersistentManifold.cpp      getCacheEntry, replaceContactPoint, addManifoldPoint,
//                                                            sortCachedPoints, refreshContactPoints — as contact_ref.h has them
//                                                            for the plane, with a normal per point and a moving frame for B
//   LinearMath/btVector3.h                                   btPlaneSpace1 (the friction direction when nothing slides)
//
// PARITY STATUS: "parity unpinned" in the strict sense (Bullet is a third-party dependency whose version is not pinned and the
// reference holds no fixtures), but every function here is compared, as expression trees, with the reference's compiled code:
// dBoxBox2's separating-axis phase (oracle/tools/check_boxbox_order.py), its contact generation — face contacts for every code /
// incident axis / sign, intersectRectQuad2's clipping, the cull, dLineClosestApproach — and cullPoints2
// (check_boxbox_contacts.py: integers concrete, floats symbolic), addContactPoint / getCacheEntry / sortCachedPoints /
// refreshContactPoints and the solver's row set-up (check_solver_setup.py), the solver's parameters (check_contact_order.py).
// What the compiled code does differently from the source, and this file follows: pairwise four-term sums and ONE reciprocal per
// edge axis in the SAT, 0x3eaaaaab / (a + q) in cullPoints2, beta by one division in dLineClosestApproach, DotXZY /
// InvMassPlusDot / XformPoint / XformPointB and no friction warm start in the rows (contact_ref.h's header).
// Specification choices where Bullet's behaviour depends on history the reference leaves open (stated, not hidden):
//   * which body is "A": Bullet orders a pair by broadphase proxy id = creation order, which in the reference follows the
//     iteration order of a std::unordered_map and every re-creation; here the DYNAMIC body is always A (the manifold's body0);
//   * pairs are the history-free core of the pair cache, as everywhere: fed AABBs overlap, filter passes both ways; a manifold
//     lives exactly as long as its pair (Bullet's fat leaf volumes keep an EMPTY manifold a little longer — no observable difference);
//   * a body keeps at most kMaxBoxManifolds manifolds with boxes (the lowest entity ids); Bullet has no such limit.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstring>

#include "contact_ref.h"

namespace orc {
namespace ct {

constexpr int kMaxBoxManifolds = 4;

// btManifoldPoint for a box pair: a normal per point, localB in B's frame
struct BoxPoint {
    Vec3 localA{0, 0, 0}, localB{0, 0, 0};
    Vec3 normalB{0, 0, 0};      // m_normalWorldOnB, as of the step that added (or replaced) the point — refresh does not touch it
    float distance = 0.0f;      // m_distance1, refreshed every step
    float appliedImpulse = 0.0f, appliedImpulseLateral1 = 0.0f;
    Vec3 worldA{0, 0, 0}, worldB{0, 0, 0};
};

struct BoxManifold {
    uint32_t other = 0; // entity id of the Static / Kinematic box
    int n = 0;
    BoxPoint p[4];
    float breaking = 0.02f;    // min of the two shapes' getContactBreakingThreshold (btCollisionDispatcher::getNewManifold)
    float friction = 0.0f;     // m_combinedFriction of its points (the same for all: both bodies' values are per body)
    float restitution = 0.0f;  // m_combinedRestitution
};

// pose + shape of one box as the detector sees it
struct BoxPose {
    Vec3 origin;
    Mat3 basis;
    Vec3 halfWithMargin; // btBoxShape::getHalfExtentsWithMargin()
};

// btPlaneSpace1(n, p, q): only p is used (one friction direction)
inline Vec3 PlaneSpace1(const Vec3& n)
{
    if (std::fabs(n.z) > 0.7071067811865475244008443621048490f) { // SIMDSQRT12
        const float a = n.y * n.y + n.z * n.z;
        const float k = 1.0f / std::sqrt(a); // btRecipSqrt
        return V(0.0f, -n.z * k, n.y * k);
    }
    const float a = n.x * n.x + n.y * n.y;
    const float k = 1.0f / std::sqrt(a);
    return V(-n.y * k, n.x * k, 0.0f);
}

namespace bb {

// dMatrix3: 3 rows of 4 (the fourth column unused), element (i, j) at [4 i + j]
struct M34 {
    float m[12];
};
inline M34 FromBasis(const Mat3& b)
{
    M34 r;
    for (int j = 0; j < 3; ++j) {
        r.m[0 + 4 * j] = b.m[j][0];
        r.m[1 + 4 * j] = b.m[j][1];
        r.m[2 + 4 * j] = b.m[j][2];
        r.m[3 + 4 * j] = 0.0f;
    }
    return r;
}
// dDOTpq(a, b, p, q) = a[0] b[0] + a[p] b[q] + a[2p] b[2q]
inline float Dot(const float* a, int p, const float* b, int q) { return a[0] * b[0] + a[p] * b[q] + a[2 * p] * b[2 * q]; }

struct Out {
    int n = 0;
    Vec3 normalOnB[4]; // as handed to addContactPoint: -normal
    Vec3 point[4];     // pointInWorld (on B)
    float depth[4];    // signed distance (negative: penetration)
    int code = 0;
};

inline void LineClosestApproach(const Vec3& pa, const Vec3& ua, const Vec3& pb, const Vec3& ub, float* alpha, float* beta)
{
    const float p[3] = {pb.x - pa.x, pb.y - pa.y, pb.z - pa.z};
    const float a[3] = {ua.x, ua.y, ua.z}, b[3] = {ub.x, ub.y, ub.z};
    const float uaub = Dot(a, 1, b, 1);
    const float q1 = Dot(a, 1, p, 1);
    const float q2 = -Dot(b, 1, p, 1);
    float d = 1.0f - uaub * uaub;
    if (d <= 0.0001f) {
        *alpha = 0.0f;
        *beta = 0.0f;
    } else {
        // (d = 1 / d and two products in the source; only beta reaches the contact point, and the reference's compiled code —
        //  MSVC /fp:fast, read off the exe by oracle/tools/check_boxbox_order.py — divides by d once instead)
        *alpha = (q1 + uaub * q2) / d;
        *beta = (uaub * q1 + q2) / d;
    }
}

// intersection of the rectangle (+-h[0], +-h[1]) with the quadrilateral p[0..7]; points as x, y pairs in ret; count 0..8
inline int IntersectRectQuad2(const float h[2], const float p[8], float ret[16])
{
    int nq = 4, nr = 0;
    float buffer[16];
    const float* q = p;
    float* r = ret;
    for (int dir = 0; dir <= 1; ++dir) {
        for (int sign = -1; sign <= 1; sign += 2) {
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
                        goto done;
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
                        goto done;
                    }
                }
                pq += 2;
            }
            q = r;
            r = (q == ret) ? buffer : ret;
            nq = nr;
        }
    }
done:
    if (q != ret) std::memcpy(ret, q, static_cast<size_t>(nr) * 2 * sizeof(float));
    return nr;
}

// cullPoints2: m of the n points (2 n floats) that represent the polygon best; i0 is always the first
inline void CullPoints2(int n, const float p[], int m, int i0, int iret[])
{
    constexpr float kPi = 3.14159265f; // M__PI of the file
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
        if (std::fabs(a + q) > bt::kEpsilon) {
            a = 0.3333333432674408f / (a + q); // (1.f / (3 * (a + q)) in the source: the reference's compiled code — MSVC /fp:fast — divides the constant 0x3eaaaaab)
        } else {
            a = 1.0e18f; // BT_LARGE_FLOAT
        }
        cx = a * (cx + q * (p[n * 2 - 2] + p[0]));
        cy = a * (cy + q * (p[n * 2 - 1] + p[1]));
    }
    float A[8];
    for (int i = 0; i < n; ++i) A[i] = bt::Atan2(p[i * 2 + 1] - cy, p[i * 2] - cx);
    int avail[8];
    for (int i = 0; i < n; ++i) avail[i] = 1;
    avail[i0] = 0;
    iret[0] = i0;
    iret++;
    for (int j = 1; j < m; ++j) {
        a = static_cast<float>(j) * (2 * kPi / m) + A[i0];
        if (a > kPi) a -= 2 * kPi;
        float maxdiff = 1e9f, diff;
        *iret = i0;
        for (int i = 0; i < n; ++i) {
            if (avail[i]) {
                diff = std::fabs(A[i] - a);
                if (diff > kPi) diff = 2 * kPi - diff;
                if (diff < maxdiff) {
                    maxdiff = diff;
                    *iret = i;
                }
            }
        }
        avail[*iret] = 0;
        iret++;
    }
}

// dBoxBox2 with maxc = 4; the contact points go to `out` as btBoxBoxDetector hands them to Result::addContactPoint
inline int BoxBox2(const Vec3& p1v, const M34& R1m, const Vec3& side1, const Vec3& p2v, const M34& R2m, const Vec3& side2, Out& out)
{
    const float fudge_factor = 1.05f;
    const float* R1 = R1m.m;
    const float* R2 = R2m.m;
    const float p1[3] = {p1v.x, p1v.y, p1v.z}, p2[3] = {p2v.x, p2v.y, p2v.z};
    float p[3], pp[3], normalC[3] = {0.0f, 0.0f, 0.0f};
    const float* normalR = nullptr;
    float A[3], B[3], s, s2, l;
    int invert_normal, code;
    float normal[3];

    p[0] = p2[0] - p1[0];
    p[1] = p2[1] - p1[1];
    p[2] = p2[2] - p1[2];
    pp[0] = Dot(R1 + 0, 4, p, 1); // dMULTIPLY1_331: dDOT41(R1 + i, p)
    pp[1] = Dot(R1 + 1, 4, p, 1);
    pp[2] = Dot(R1 + 2, 4, p, 1);
    A[0] = side1.x * 0.5f;
    A[1] = side1.y * 0.5f;
    A[2] = side1.z * 0.5f;
    B[0] = side2.x * 0.5f;
    B[1] = side2.y * 0.5f;
    B[2] = side2.z * 0.5f;
    const float R11 = Dot(R1 + 0, 4, R2 + 0, 4), R12 = Dot(R1 + 0, 4, R2 + 1, 4), R13 = Dot(R1 + 0, 4, R2 + 2, 4);
    const float R21 = Dot(R1 + 1, 4, R2 + 0, 4), R22 = Dot(R1 + 1, 4, R2 + 1, 4), R23 = Dot(R1 + 1, 4, R2 + 2, 4);
    const float R31 = Dot(R1 + 2, 4, R2 + 0, 4), R32 = Dot(R1 + 2, 4, R2 + 1, 4), R33 = Dot(R1 + 2, 4, R2 + 2, 4);
    float Q11 = std::fabs(R11), Q12 = std::fabs(R12), Q13 = std::fabs(R13);
    float Q21 = std::fabs(R21), Q22 = std::fabs(R22), Q23 = std::fabs(R23);
    float Q31 = std::fabs(R31), Q32 = std::fabs(R32), Q33 = std::fabs(R33);

    s = -3.402823466e+38f; // -dInfinity = -FLT_MAX
    invert_normal = 0;
    code = 0;
// (the four-term sums of expr2 are written (t0 + t1) + (t2 + t3): the association of the reference's COMPILED dBoxBox2 — Bullet is
    //  built with MSVC /fp:fast — read off the exe by oracle/tools/check_boxbox_order.py for all fifteen axes)
#define BGE_TST(expr1, expr2, norm, cc)    \
    s2 = std::fabs(expr1) - (expr2);       \
    if (s2 > 0) return 0;                  \
    if (s2 > s) {                          \
        s = s2;                            \
        normalR = norm;                    \
        invert_normal = ((expr1) < 0);     \
        code = (cc);                       \
    }
    BGE_TST(pp[0], ((A[0] + B[0] * Q11) + (B[1] * Q12 + B[2] * Q13)), R1 + 0, 1);
    BGE_TST(pp[1], ((A[1] + B[0] * Q21) + (B[1] * Q22 + B[2] * Q23)), R1 + 1, 2);
    BGE_TST(pp[2], ((A[2] + B[0] * Q31) + (B[1] * Q32 + B[2] * Q33)), R1 + 2, 3);
    BGE_TST(Dot(R2 + 0, 4, p, 1), ((A[0] * Q11 + A[1] * Q21) + (A[2] * Q31 + B[0])), R2 + 0, 4);
    BGE_TST(Dot(R2 + 1, 4, p, 1), ((A[0] * Q12 + A[1] * Q22) + (A[2] * Q32 + B[1])), R2 + 1, 5);
    BGE_TST(Dot(R2 + 2, 4, p, 1), ((A[0] * Q13 + A[1] * Q23) + (A[2] * Q33 + B[2])), R2 + 2, 6);
#undef BGE_TST
// (edge axes: the reference's compiled code — MSVC /fp:fast — forms ONE reciprocal 1 / l and multiplies: s2 * (1 / l), n * (1 / l), with
//  (-1 / l) * r for the negated component, which is the same bits; oracle/tools/check_boxbox_order.py reads it off the exe)
#define BGE_TST(expr1, expr2, n1, n2, n3, cc)                            \
    s2 = std::fabs(expr1) - (expr2);                                          \
    if (s2 > bt::kEpsilon) return 0;                                              \
    l = std::sqrt((n1) * (n1) + (n2) * (n2) + (n3) * (n3));                   \
    if (l > bt::kEpsilon) {                                                       \
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
        normal[0] = Dot(R1 + 0, 1, normalC, 1); // dMULTIPLY0_331: dDOT(R1 + 4 i, normalC)
        normal[1] = Dot(R1 + 4, 1, normalC, 1);
        normal[2] = Dot(R1 + 8, 1, normalC, 1);
    }
    if (invert_normal) {
        normal[0] = -normal[0];
        normal[1] = -normal[1];
        normal[2] = -normal[2];
    }
    const float depth = -s;
    out.code = code;
    const Vec3 minusNormal = V(-normal[0], -normal[1], -normal[2]);

    if (code > 6) {
        // an edge of box 1 touches an edge of box 2
        float pa[3], pb[3];
        for (int i = 0; i < 3; ++i) pa[i] = p1[i];
        for (int j = 0; j < 3; ++j) {
            const float sign = (Dot(normal, 1, R1 + j, 4) > 0) ? 1.0f : -1.0f;
            for (int i = 0; i < 3; ++i) pa[i] += sign * A[j] * R1[i * 4 + j];
        }
        for (int i = 0; i < 3; ++i) pb[i] = p2[i];
        for (int j = 0; j < 3; ++j) {
            const float sign = (Dot(normal, 1, R2 + j, 4) > 0) ? -1.0f : 1.0f;
            for (int i = 0; i < 3; ++i) pb[i] += sign * B[j] * R2[i * 4 + j];
        }
        float alpha, beta;
        float ua[3], ub[3];
        for (int i = 0; i < 3; ++i) ua[i] = R1[(code - 7) / 3 + i * 4];
        for (int i = 0; i < 3; ++i) ub[i] = R2[(code - 7) % 3 + i * 4];
        LineClosestApproach(V(pa[0], pa[1], pa[2]), V(ua[0], ua[1], ua[2]), V(pb[0], pb[1], pb[2]), V(ub[0], ub[1], ub[2]), &alpha, &beta);
        for (int i = 0; i < 3; ++i) pa[i] += ua[i] * alpha;
        for (int i = 0; i < 3; ++i) pb[i] += ub[i] * beta;
        out.normalOnB[0] = minusNormal; // output.addContactPoint(-normal, pb, -*depth)
        out.point[0] = V(pb[0], pb[1], pb[2]);
        out.depth[0] = -depth;
        out.n = 1;
        return 1;
    }

    // face - something: reference face a (the normal is perpendicular to it), incident face b
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
    nr[0] = Dot(Rb + 0, 4, normal2, 1); // dMULTIPLY1_331(nr, Rb, normal2)
    nr[1] = Dot(Rb + 1, 4, normal2, 1);
    nr[2] = Dot(Rb + 2, 4, normal2, 1);
    anr[0] = std::fabs(nr[0]);
    anr[1] = std::fabs(nr[1]);
    anr[2] = std::fabs(nr[2]);
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
    c1 = Dot(center, 1, Ra + code1, 4); // dDOT14
    c2 = Dot(center, 1, Ra + code2, 4);
    m11 = Dot(Ra + code1, 4, Rb + a1, 4);
    m12 = Dot(Ra + code1, 4, Rb + a2, 4);
    m21 = Dot(Ra + code2, 4, Rb + a1, 4);
    m22 = Dot(Ra + code2, 4, Rb + a2, 4);
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
    const int n = IntersectRectQuad2(rect, quad, ret);
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
        dep[cnum] = Sa[codeN] - Dot(normal2, 1, point + cnum * 3, 1);
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
            float w[3];
            if (code < 4) {
                for (int i = 0; i < 3; ++i) w[i] = point[j * 3 + i] + pa[i];
            } else {
                for (int i = 0; i < 3; ++i) w[i] = point[j * 3 + i] + pa[i] - normal[i] * dep[j];
            }
            out.normalOnB[j] = minusNormal;
            out.point[j] = V(w[0], w[1], w[2]);
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
        CullPoints2(cnum, ret, maxc, i1, iret);
        for (int j = 0; j < maxc; ++j) {
            float w[3];
            for (int i = 0; i < 3; ++i) w[i] = point[iret[j] * 3 + i] + pa[i];
            if (code < 4) {
                out.point[j] = V(w[0], w[1], w[2]);
            } else {
                // posInWorld - normal * dep: btVector3 operator* then operator-
                out.point[j] = V(w[0] - normal[0] * dep[iret[j]], w[1] - normal[1] * dep[iret[j]], w[2] - normal[2] * dep[iret[j]]);
            }
            out.normalOnB[j] = minusNormal;
            out.depth[j] = -dep[iret[j]];
        }
        cnum = maxc;
        out.n = maxc;
    }
    return cnum;
}

} // namespace bb

// btPersistentManifold::sortCachedPoints for a box manifold (the same rule as contact_ref.h's SortCachedPoints)
inline int SortCachedBoxPoints(const BoxManifold& m, const BoxPoint& pt)
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

// btBoxBoxCollisionAlgorithm::processCollision for body0 = the Dynamic box `a`, body1 = the Static / Kinematic box `b`
inline void CollideBoxBox(BoxManifold& m, const BoxPose& a, const BoxPose& b)
{
    bb::Out out;
    bb::BoxBox2(a.origin, bb::FromBasis(a.basis), Scale(a.halfWithMargin, 2.0f), b.origin, bb::FromBasis(b.basis), Scale(b.halfWithMargin, 2.0f), out);
    for (int k = 0; k < out.n; ++k) {
        // btManifoldResult::addContactPoint(normalOnBInWorld, pointInWorld, depth)
        const float depth = out.depth[k];
        if (depth > m.breaking) continue;
        const Vec3 normalOnB = out.normalOnB[k];
        const Vec3 pointInWorld = out.point[k];
        const Vec3 pointA = Add(pointInWorld, Scale(normalOnB, depth));
        BoxPoint np;
        np.localA = MatTVec(a.basis, Sub(pointA, a.origin));        // body0 transform .invXform(pointA)
        np.localB = MatTVec(b.basis, Sub(pointInWorld, b.origin));  // body1 transform .invXform(pointInWorld)
        np.normalB = normalOnB;
        np.distance = depth;
        np.worldA = pointA;
        np.worldB = pointInWorld;
        float shortest = m.breaking * m.breaking; // getCacheEntry
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
            np.appliedImpulse = m.p[nearest].appliedImpulse; // replaceContactPoint
            np.appliedImpulseLateral1 = m.p[nearest].appliedImpulseLateral1;
            m.p[nearest] = np;
        } else {
            int insert = m.n; // addManifoldPoint
            if (insert == 4) {
                insert = SortCachedBoxPoints(m, np);
            } else {
                m.n++;
            }
            if (insert < 0) insert = 0;
            m.p[insert] = np;
        }
    }
    // refreshContactPoints(body0 transform, body1 transform)
    for (int i = m.n - 1; i >= 0; --i) {
        BoxPoint& c = m.p[i];
        c.worldA = XformPoint(a.basis, a.origin, c.localA);
        c.worldB = XformPointB(b.basis, b.origin, c.localB);
        c.distance = Dot(Sub(c.worldA, c.worldB), c.normalB);
    }
    for (int i = m.n - 1; i >= 0; --i) {
        BoxPoint& c = m.p[i];
        bool remove = !(c.distance <= m.breaking);
        if (!remove) {
            const Vec3 projectedPoint = Sub(c.worldA, Scale(c.normalB, c.distance));
            const Vec3 projectedDifference = Sub(c.worldB, projectedPoint);
            const float distance2d = Dot(projectedDifference, projectedDifference);
            remove = distance2d > m.breaking * m.breaking;
        }
        if (remove) {
            const int last = m.n - 1;
            if (i != last) m.p[i] = m.p[last];
            m.p[last] = BoxPoint{};
            m.n--;
        }
    }
}

// (the solver for these manifolds: island_ref.h — SolveBody for the island {body}, SolveIsland for several Dynamic bodies)

} // namespace ct
} // namespace orc
