// oracle/synth.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Deterministic synthetic inputs for the benchmark configurations:
// counter-based splitmix64, one 24-bit uniform per (seed, entity, field).  The same
// generator is restated in numpy in banggameengine_amd/synth.py; tests require both
// to agree bit for bit.  The reference ships a 3-entity scene only
// (assets/scenes/demo.json:48-108); everything larger is synthetic.
#pragma once

#include <cstdint>

namespace orc {
namespace synth {

inline uint64_t mix64(uint64_t z)
{
    z ^= z >> 30;
    z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27;
    z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

// u in [0,1) with 24 random bits
inline float uniform(uint64_t seed, uint64_t entity, uint32_t field)
{
    const uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ull * (16ull * entity + field + 1ull));
    return static_cast<float>(h >> 40) * 0x1p-24f;
}

inline float range(uint64_t seed, uint64_t entity, uint32_t field, float lo, float hi)
{
    return lo + (hi - lo) * uniform(seed, entity, field);
}

enum Field : uint32_t {
    kPosX = 0, kPosY = 1, kPosZ = 2,
    kEulerX = 3, kEulerY = 4, kEulerZ = 5,
    kScaleX = 6, kScaleY = 7, kScaleZ = 8,
    kVelX = 9, kVelY = 10, kVelZ = 11,
};

enum Shape : int { kFlat = 0, kChains4 = 1, kSubtree64 = 2 };
enum PosBox : int { kWorldSlab = 0, kDenseCube = 1 };

// parent entity index (0-based) or -1
inline int64_t parent_of(int shape, int64_t i)
{
    switch (shape) {
    case kChains4:
        return (i % 4 != 0) ? i - 1 : -1;
    case kSubtree64: {
        const int64_t base = i - (i % 64);
        const int64_t k = i % 64;
        if (k == 0) return -1;
        if (k < 4) return base;
        if (k < 16) return base + 1 + (k - 4) / 4;
        return base + 4 + (k - 16) / 4;
    }
    default:
        return -1;
    }
}

inline void trs(uint64_t seed, int64_t i, int posBox, float* pos, float* euler, float* scale)
{
    if (posBox == kDenseCube) {
        pos[0] = range(seed, i, kPosX, 0.0f, 262.0f);
        pos[1] = range(seed, i, kPosY, 0.0f, 262.0f);
        pos[2] = range(seed, i, kPosZ, 0.0f, 262.0f);
    } else {
        pos[0] = range(seed, i, kPosX, -250.0f, 250.0f);
        pos[1] = range(seed, i, kPosY, 1.0f, 50.0f);
        pos[2] = range(seed, i, kPosZ, -250.0f, 250.0f);
    }
    euler[0] = range(seed, i, kEulerX, -1.5f, 1.5f);
    euler[1] = range(seed, i, kEulerY, -3.1f, 3.1f);
    euler[2] = range(seed, i, kEulerZ, -3.1f, 3.1f);
    scale[0] = range(seed, i, kScaleX, 0.5f, 2.0f);
    scale[1] = range(seed, i, kScaleY, 0.5f, 2.0f);
    scale[2] = range(seed, i, kScaleZ, 0.5f, 2.0f);
}

inline void velocity(uint64_t seed, int64_t i, float* v)
{
    v[0] = range(seed, i, kVelX, -1.0f, 1.0f);
    v[1] = range(seed, i, kVelY, -1.0f, 1.0f);
    v[2] = range(seed, i, kVelZ, -1.0f, 1.0f);
}

} // namespace synth
} // namespace orc
