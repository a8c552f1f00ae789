// oracle/bullet_math.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the Bullet (bullet3, single-precision build, version unpinned
// by the reference — vcpkg port, >= 2.87 per SURVEY.md §2) arithmetic that the
// reference's physics slice reaches through these call sites:
//   src/physics/PhysicsSystem.cpp:40-45   btQuaternion::setEulerZYX(e.y, e.x, e.z)
//   src/physics/PhysicsSystem.cpp:57-64   MakeBtTransform (origin + rotation)
//   src/physics/PhysicsSystem.cpp:130     setGravity
//   src/physics/PhysicsSystem.cpp:426-429 mass = max(mass, 0.01) for Dynamic
//   src/physics/PhysicsSystem.cpp:686-707 btBoxShape / btCapsuleShape sizes
//   src/physics/PhysicsSystem.cpp:863     stepSimulation(dt, 4, fixedStep)
//   src/physics/PhysicsSystem.cpp:937-947 btMatrix3x3(q).getEulerZYX
// Bullet's source is NOT under /root/reference and nothing in the reference pins its
// output, so every function here restates Bullet's PUBLISHED algorithm (file names
// below are bullet3 paths) for the scalar (non-SSE) code path.
//
// PARITY STATUS: "parity unpinned" — spec-derived; EXCEPT the header-inline conversions
// (setEulerZYX through the reference's ToBtQuaternion, btMatrix3x3::setRotation / getEulerZYX /
// getRotation), whose compiled bodies sit in the reference's committed PhysicsSystem.obj and were
// executed symbolically against the formulas below (oracle/tools/check_bullet_order.py: operation
// order, association, constants, branch structure and the euler.y/x/z -> yaw/pitch/roll mapping);
// AND integrateTransform, setGravity / the force impulse, btBoxShape / btCapsuleShape (margins, getAabb)
// and updateSingleAabb, which the same script finds inside the symbol-less SandboxCity.exe.
//
// libm: Bullet calls the platform sinf/cosf/asinf/atan2f.  g_libm selects either the
// platform libm (what Bullet does) or the deterministic routines of
// include/bge_detmath.h (what the GPU path uses so that results can be compared
// bit for bit); tests bound the distance between the two.
#pragma once

#include <cmath>
#include <cstdint>

#include "../include/bge_detmath.h"

namespace orc {
namespace bt {

enum Libm : int { kLibmDeterministic = 0, kLibmPlatform = 1 };
inline int g_libm = kLibmDeterministic;

inline float Sin(float x) { return g_libm == kLibmPlatform ? sinf(x) : bge_det_sinf(x); }
inline float Cos(float x) { return g_libm == kLibmPlatform ? cosf(x) : bge_det_cosf(x); }
inline float Asin(float x)
{
    // btAsin (LinearMath/btScalar.h) clamps to [-1, 1]
    if (x < -1.0f) x = -1.0f;
    if (x > 1.0f) x = 1.0f;
    return g_libm == kLibmPlatform ? asinf(x) : bge_det_asinf(x);
}
inline float Atan2(float y, float x) { return g_libm == kLibmPlatform ? atan2f(y, x) : bge_det_atan2f(y, x); }

inline constexpr float kEpsilon = 1.1920928955078125e-07f;             // SIMD_EPSILON = FLT_EPSILON
inline constexpr float kPi = 3.1415926535897932384626433832795029f;    // SIMD_PI
inline constexpr float kHalfPi = kPi * 0.5f;                           // SIMD_HALF_PI
inline constexpr float kAngularMotionThreshold = 0.5f * kHalfPi;       // ANGULAR_MOTION_THRESHOLD
inline constexpr float kContactBreakingThreshold = 0.02f;              // gContactBreakingThreshold
inline constexpr float kConvexDistanceMargin = 0.04f;                  // CONVEX_DISTANCE_MARGIN

struct Quat {
    float x, y, z, w;
};
struct Vec3 {
    float x, y, z;
};
struct Mat3 {
    float m[3][3]; // m[row][col], btMatrix3x3::m_el[row]
};

// btQuaternion::setEulerZYX(yawZ, pitchY, rollX)   (LinearMath/btQuaternion.h)
inline Quat QuatFromEulerZYX(float yawZ, float pitchY, float rollX)
{
    const float halfYaw = yawZ * 0.5f;
    const float halfPitch = pitchY * 0.5f;
    const float halfRoll = rollX * 0.5f;
    const float cosYaw = Cos(halfYaw);
    const float sinYaw = Sin(halfYaw);
    const float cosPitch = Cos(halfPitch);
    const float sinPitch = Sin(halfPitch);
    const float cosRoll = Cos(halfRoll);
    const float sinRoll = Sin(halfRoll);
    Quat q;
    // product order of bullet3's btQuaternion::setEulerZYX as the reference's build compiled it (roll factor first,
    // left-associated) — oracle/tools/check_bullet_order.py checks it against PhysicsSystem.obj
    q.x = sinRoll * cosPitch * cosYaw - cosRoll * sinPitch * sinYaw;
    q.y = cosRoll * sinPitch * cosYaw + sinRoll * cosPitch * sinYaw;
    q.z = cosRoll * cosPitch * sinYaw - sinRoll * sinPitch * cosYaw;
    q.w = cosRoll * cosPitch * cosYaw + sinRoll * sinPitch * sinYaw;
    return q;
}

// ToBtQuaternion(euler) of the reference: setEulerZYX(euler.y, euler.x, euler.z)
// (src/physics/PhysicsSystem.cpp:40-45)
inline Quat QuatFromTransformEuler(float ex, float ey, float ez) { return QuatFromEulerZYX(ey, ex, ez); }

// btQuaternion operator*(q1, q2), scalar path
inline Quat QuatMul(const Quat& a, const Quat& b)
{
    Quat r;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    return r;
}

inline float QuatLength2(const Quat& q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }

// btQuaternion::safeNormalize: normalize (multiply by 1/length) iff length2 > SIMD_EPSILON
// (inside the compiled integrateTransform the squares are summed pairwise, (x^2 + y^2) + (z^2 + w^2) —
//  check_bullet_order.py; btMatrix3x3::setRotation's length2 is the left-associated sum of QuatLength2)
inline Quat QuatSafeNormalize(Quat q)
{
    const float l2 = (q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w);
    if (l2 > kEpsilon) {
        const float s = 1.0f / std::sqrt(l2);
        q.x *= s;
        q.y *= s;
        q.z *= s;
        q.w *= s;
    }
    return q;
}

// btMatrix3x3::setRotation(q)   (LinearMath/btMatrix3x3.h)
inline Mat3 MatFromQuat(const Quat& q)
{
    const float d = QuatLength2(q);
    const float s = 2.0f / d;
    const float xs = q.x * s, ys = q.y * s, zs = q.z * s;
    const float wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
    const float xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
    const float yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
    Mat3 r;
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

// btMatrix3x3::getRotation(q), scalar path
inline Quat QuatFromMat(const Mat3& a)
{
    const float trace = a.m[0][0] + a.m[1][1] + a.m[2][2];
    float t[4];
    if (trace > 0.0f) {
        float s = std::sqrt(trace + 1.0f);
        t[3] = s * 0.5f;
        s = 0.5f / s;
        t[0] = (a.m[2][1] - a.m[1][2]) * s;
        t[1] = (a.m[0][2] - a.m[2][0]) * s;
        t[2] = (a.m[1][0] - a.m[0][1]) * s;
    } else {
        const int i = a.m[0][0] < a.m[1][1] ? (a.m[1][1] < a.m[2][2] ? 2 : 1) : (a.m[0][0] < a.m[2][2] ? 2 : 0);
        const int j = (i + 1) % 3;
        const int k = (i + 2) % 3;
        float s = std::sqrt(a.m[i][i] - a.m[j][j] - a.m[k][k] + 1.0f);
        t[i] = s * 0.5f;
        s = 0.5f / s;
        t[3] = (a.m[k][j] - a.m[j][k]) * s;
        t[j] = (a.m[j][i] + a.m[i][j]) * s;
        t[k] = (a.m[k][i] + a.m[i][k]) * s;
    }
    return Quat{t[0], t[1], t[2], t[3]};
}

// btMatrix3x3::getEulerZYX(yaw, pitch, roll), solution 1
inline void EulerZYXFromMat(const Mat3& a, float& yaw, float& pitch, float& roll)
{
    if (std::fabs(a.m[2][0]) >= 1.0f) {
        yaw = 0.0f;
        const float delta = Atan2(a.m[0][0], a.m[0][2]);
        if (a.m[2][0] > 0.0f) { // gimbal locked up
            pitch = kPi / 2.0f;
            roll = pitch + delta;
        } else { // gimbal locked down
            pitch = -kPi / 2.0f;
            roll = -pitch + delta;
        }
    } else {
        pitch = -Asin(a.m[2][0]);
        const float c = Cos(pitch);
        roll = Atan2(a.m[2][1] / c, a.m[2][2] / c);
        yaw = Atan2(a.m[1][0] / c, a.m[0][0] / c);
    }
}

// What SyncRigidBodiesFromPhysics stores into Transform::rotationEuler
// (src/physics/PhysicsSystem.cpp:943-947): {x: pitch, y: yaw, z: roll}
// The reference does not take the euler angles from the body's basis directly: it reads
//   rotation = worldTransform.getRotation();  btMatrix3x3(rotation).getEulerZYX(yaw, pitch, roll);
// i.e. basis -> quaternion (getRotation) -> matrix (setRotation) -> angles.  The compiled function calls exactly
// getRotation, setRotation, getEulerZYX in that order (relocations of ?SyncRigidBodiesFromPhysics@ in PhysicsSystem.obj,
// oracle/tools/check_bullet_order.py); the round trip moves the matrix by an ulp now and then.
inline Vec3 TransformEulerFromMat(const Mat3& a)
{
    const Mat3 m = MatFromQuat(QuatFromMat(a));
    float yaw, pitch, roll;
    EulerZYXFromMat(m, yaw, pitch, roll);
    return Vec3{pitch, yaw, roll};
}

// btTransformUtil::integrateTransform, rotation part (LinearMath/btTransformUtil.h):
// exponential map of angvel*dt applied on the left of the current orientation.  Checked against the compiled function
// in the reference's SandboxCity.exe (found through its unique reference to the constant 1/48) on its three control
// paths — regular, Taylor (fAngle < 0.001) and clamped (fAngle*dt > pi/4) — by oracle/tools/check_bullet_order.py.
inline Quat IntegrateOrientation(const Quat& orn0, const Vec3& angvel, float dt)
{
    const float fAngle2 = angvel.x * angvel.x + angvel.y * angvel.y + angvel.z * angvel.z;
    float fAngle = 0.0f;
    if (fAngle2 > kEpsilon) {
        fAngle = std::sqrt(fAngle2);
    }
    if (fAngle * dt > kAngularMotionThreshold) {
        fAngle = kAngularMotionThreshold / dt;
    }
    float k;
    if (fAngle < 0.001f) {
        // association as compiled in the reference's build ((dt*dt) * (dt * 1/48)): oracle/tools/check_bullet_order.py
        k = 0.5f * dt - ((dt * dt) * (dt * 0.020833333333f)) * fAngle * fAngle;
    } else {
        k = Sin(0.5f * fAngle * dt) / fAngle;
    }
    const Quat dorn{angvel.x * k, angvel.y * k, angvel.z * k, Cos(fAngle * dt * 0.5f)};
    Quat predicted = QuatSafeNormalize(QuatMul(dorn, orn0));
    if (QuatLength2(predicted) > kEpsilon) {
        return predicted;
    }
    return orn0;
}

// Effective AABB half extents of a collider in its own frame.
//  Box  (btBoxShape ctor + setSafeMargin + getAabb → btTransformAabb with margin):
//       implicit = he - 0.04; margin = min(0.04, 0.1*min(he)) re-applied through
//       btBoxShape::setMargin; extents = implicit + margin.
//  Capsule (btCapsuleShape::getAabb, up axis Y): (r, r + halfHeight, r).
// he is the collider size after the reference's clamps (PhysicsSystem.cpp:692-703).
inline Vec3 BoxAabbHalfExtents(float hx, float hy, float hz)
{
    const float m0 = kConvexDistanceMargin;
    float ix = hx - m0, iy = hy - m0, iz = hz - m0;
    const float minDim = hx < hy ? (hx < hz ? hx : hz) : (hy < hz ? hy : hz);
    const float safe = 0.1f * minDim;
    float margin = m0;
    if (safe < margin) {
        // btBoxShape::setMargin: dims = (dims + oldMargin) - newMargin
        ix = (ix + m0) - safe;
        iy = (iy + m0) - safe;
        iz = (iz + m0) - safe;
        margin = safe;
    }
    return Vec3{ix + margin, iy + margin, iz + margin};
}

inline Vec3 CapsuleAabbHalfExtents(float radius, float halfHeight)
{
    return Vec3{radius, radius + halfHeight, radius};
}

// btTransformAabb: centre ± |R|·he, then btCollisionWorld::updateSingleAabb's contact threshold.
inline void AabbOfPose(const Vec3& origin, const Mat3& r, const Vec3& he, float* mn, float* mx)
{
    const float ex = std::fabs(r.m[0][0]) * he.x + std::fabs(r.m[0][1]) * he.y + std::fabs(r.m[0][2]) * he.z;
    const float ey = std::fabs(r.m[1][0]) * he.x + std::fabs(r.m[1][1]) * he.y + std::fabs(r.m[1][2]) * he.z;
    const float ez = std::fabs(r.m[2][0]) * he.x + std::fabs(r.m[2][1]) * he.y + std::fabs(r.m[2][2]) * he.z;
    mn[0] = (origin.x - ex) - kContactBreakingThreshold;
    mn[1] = (origin.y - ey) - kContactBreakingThreshold;
    mn[2] = (origin.z - ez) - kContactBreakingThreshold;
    mx[0] = (origin.x + ex) + kContactBreakingThreshold;
    mx[1] = (origin.y + ey) + kContactBreakingThreshold;
    mx[2] = (origin.z + ez) + kContactBreakingThreshold;
}

} // namespace bt
} // namespace orc
