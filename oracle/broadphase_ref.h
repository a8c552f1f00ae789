// oracle/broadphase_ref.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Pair-set specification for the broadphase step.  The reference delegates to Bullet's
// btDbvtBroadphase (src/physics/PhysicsSystem.cpp:124), whose pair cache depends on the
// history of fat-AABB updates and cannot be reproduced without Bullet itself.  The
// specification used here is the history-free core of it:
//
//   pair (a,b), a<b, is reported  <=>  the AABBs fed to the broadphase this step overlap on all
//   three axes with NON-strict comparisons (btDbvtAabbMm Intersect: a.min <= b.max && a.max >= b.min)
//   AND the collision filter passes both ways (btOverlapFilterCallback default:
//   (groupA & maskB) && (groupB & maskA), group = layer ? layer : 1 — PhysicsSystem.cpp:407-408,473)
//   AND at least one of the two bodies is not Static.  That last clause is a SPECIFICATION CHOICE, not Bullet's pair cache:
//   btDbvtBroadphase::createProxy / ::setAabb collide a new or moved leaf against BOTH of its trees and
//   btHashedOverlappingPairCache::needsBroadphaseCollision tests group / mask only (oracle/tools/check_pair_cache.py reads
//   both off the reference's exe), and the reference hands Bullet custom groups (addRigidBody(body, layer, mask),
//   PhysicsSystem.cpp:473), so two overlapping Static bodies DO share a cache entry there.  The entry is inert:
//   btCollisionDispatcher::needsCollision rejects a pair of inactive objects before any narrowphase, and the reference never
//   exposes the body-body pair list.  The list specified here is "the pairs that can reach the narrowphase"; where cache
//   membership is observable — trigger ghosts, which list Static bodies and each other — physics_ref.h follows the cache.
//
// Two implementations: O(n^2) brute force and sort-and-sweep on x; tests require they agree.
// PARITY STATUS: "parity unpinned" (spec-derived).
#pragma once

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <utility>
#include <vector>

namespace orc {

struct BroadphaseBody {
    float mn[3];
    float mx[3];
    uint32_t group;
    uint32_t mask;
    uint8_t isStatic;
};

inline bool AabbOverlap(const BroadphaseBody& a, const BroadphaseBody& b)
{
    return a.mn[0] <= b.mx[0] && a.mx[0] >= b.mn[0] && a.mn[1] <= b.mx[1] && a.mx[1] >= b.mn[1] &&
           a.mn[2] <= b.mx[2] && a.mx[2] >= b.mn[2];
}

inline bool PairAllowed(const BroadphaseBody& a, const BroadphaseBody& b)
{
    if (a.isStatic && b.isStatic) return false;
    return (a.group & b.mask) != 0 && (b.group & a.mask) != 0;
}

using PairList = std::vector<std::pair<uint32_t, uint32_t>>;

inline PairList PairsBruteForce(const std::vector<BroadphaseBody>& bodies)
{
    PairList out;
    const uint32_t n = static_cast<uint32_t>(bodies.size());
    for (uint32_t i = 0; i < n; ++i) {
        for (uint32_t j = i + 1; j < n; ++j) {
            if (PairAllowed(bodies[i], bodies[j]) && AabbOverlap(bodies[i], bodies[j])) out.emplace_back(i, j);
        }
    }
    return out; // already sorted lexicographically
}

inline PairList PairsSweep(const std::vector<BroadphaseBody>& bodies)
{
    const uint32_t n = static_cast<uint32_t>(bodies.size());
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    std::sort(order.begin(), order.end(),
              [&](uint32_t a, uint32_t b) { return bodies[a].mn[0] < bodies[b].mn[0]; });
    PairList out;
    for (uint32_t s = 0; s < n; ++s) {
        const BroadphaseBody& a = bodies[order[s]];
        for (uint32_t t = s + 1; t < n; ++t) {
            const BroadphaseBody& b = bodies[order[t]];
            if (b.mn[0] > a.mx[0]) break; // sorted by min x: nothing further can touch a on x
            if (PairAllowed(a, b) && AabbOverlap(a, b)) {
                const uint32_t i = std::min(order[s], order[t]);
                const uint32_t j = std::max(order[s], order[t]);
                out.emplace_back(i, j);
            }
        }
    }
    std::sort(out.begin(), out.end());
    return out;
}

} // namespace orc
