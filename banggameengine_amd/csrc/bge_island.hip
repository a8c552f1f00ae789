// bge_island.hip — Dynamic boxes against each other: the pair cache with a persistent manifold per pair, simulation islands by
// union-find over the pairs, a solver thread (or, for a big island, a workgroup that keeps Bullet's row order level by level) per island.
// DESIGN.md section 4.10; oracle/island_ref.h and oracle/physics_ref.h CollideDynamicPairs / StepIsland are what it is compared with.
#include <hip/hip_runtime.h>

#include <hipcub/hipcub.hpp>

#include "bge_contact_device.hpp"

namespace bge {

namespace {

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
// What every sweep reads of a row, and nothing else: 96 bytes.  The sweeps are bound by these bytes — the rows of 100 k two-box stacks were
// 205 MB of 128-byte records read ten times a tick, 2.18 GB per launch by FETCH_SIZE, 0.65 of the HBM peak (profiles/r03/pmc_islands.md).
// A contact row's limits are constants (0 and 1e10), a friction row's follow from its contact row's impulse at every visit.
struct IslRow {
    F3 normal, relposCrossN, angularComp, relpos2CrossN, angularCompB;
    float jacDiagABInv, rhs, cfm, friction, applied;
    uint32_t a, b;            // positions in the sorted body list; b = kNone: the fixed solver body
    float invMassA, invMassB; // the two bodies' inverse masses (B's 0 without a second body): a resolve out of LDS state then needs no load
                              // of its own — one issued behind the next row's would have to wait for that one first (loads return in order)
};
static_assert(sizeof(IslRow) == kIslRowBytes, "IslandParams::rows");
// ... and what only the split-impulse sweeps and the write-back need of a CONTACT row: 32 bytes in an array of their own
struct IslRowCold {
    float* out;          // the manifold point's appliedImpulse
    uint32_t lateral_at; // ... and how many floats behind it appliedImpulseLateral1 is
    float rhsPenetration, appliedPush;
    uint32_t pad[3];
};
static_assert(sizeof(IslRowCold) == kIslRowColdBytes, "IslandParams::rows_cold");

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

__global__ void __launch_bounds__(256) k_island_flags(WorldView w, IslandParams ip)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ip.n_bodies) return;
    const uint32_t s = ip.body_slot[i];
    ip.index_of_slot[s] = i;
    ip.pair_first[i] = isl_first_pair_of(ip, ip.entity_of_slot[s]);
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
__device__ void isl_resolve_row(IslBody* sb, IslRow& c, float lower, float upper, bool withUpperLimit)
{
    IslBody& a = sb[c.a];
    const bool two = c.b != kNone;
    float deltaImpulse = c.rhs - c.applied * c.cfm;
    const float dv1 = isl_dpps(c.relposCrossN, a.dAng) + isl_dpps(c.normal, a.dLin);
    const float dv2 = two ? isl_dpps(neg3(c.normal), sb[c.b].dLin) + isl_dpps(c.relpos2CrossN, sb[c.b].dAng) : 0.0f + 0.0f;
    deltaImpulse = __builtin_fmaf(-dv1, c.jacDiagABInv, deltaImpulse);
    deltaImpulse = __builtin_fmaf(-dv2, c.jacDiagABInv, deltaImpulse);
    const float sum = c.applied + deltaImpulse;
    if (lower < sum) {
        if (withUpperLimit && !(sum < upper)) {
            deltaImpulse = upper - c.applied;
            c.applied = upper;
        } else {
            c.applied = sum;
        }
    } else {
        deltaImpulse = lower - c.applied;
        c.applied = lower;
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
__device__ void isl_resolve_split(IslBody* sb, const IslRow& c, float rhsPenetration, float& appliedPush)
{
    if (!rhsPenetration) return;
    IslBody& a = sb[c.a];
    const bool two = c.b != kNone;
    float deltaImpulse = rhsPenetration - appliedPush * c.cfm;
    const float dv1 = isl_dot3s(c.normal, a.push) + isl_dot3s(c.relposCrossN, a.turn);
    const float dv2 = two ? isl_dot3s(neg3(c.normal), sb[c.b].push) + isl_dot3s(c.relpos2CrossN, sb[c.b].turn) : 0.0f + 0.0f;
    deltaImpulse = deltaImpulse - dv1 * c.jacDiagABInv;
    deltaImpulse = deltaImpulse - dv2 * c.jacDiagABInv;
    const float sum = appliedPush + deltaImpulse;
    if (sum < 0.0f) {
        deltaImpulse = 0.0f - appliedPush;
        appliedPush = 0.0f;
    } else {
        appliedPush = sum;
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
__device__ void isl_add_contact(IslBody* sb, IslRow& c, IslRow& fr, IslRowCold& cc, uint32_t ia, uint32_t ib, float invTimeStep, const F3& worldA, const F3& worldB,
                                const F3& n, float distance, float friction, float combinedRestitution, float* out, uint32_t lateral_at)
{
    constexpr float kErp2 = 0.2f, kSplitThreshold = -0.04f, kWarmstart = 0.85f, kSor = 1.0f, kRestitutionVelocityThreshold = 0.2f;
    IslBody& A = sb[ia];
    const bool two = ib != kNone;
    const F3 zero{0.0f, 0.0f, 0.0f};
    c.a = fr.a = ia;
    c.b = fr.b = ib;
    cc.out = out;
    cc.lateral_at = lateral_at;
    cc.pad[0] = cc.pad[1] = cc.pad[2] = 0u;
    c.invMassA = fr.invMassA = A.invMass;
    c.invMassB = fr.invMassB = two ? sb[ib].invMass : 0.0f;
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
    cc.appliedPush = 0.0f;
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
            cc.rhsPenetration = 0.0f;
        } else {
            c.rhs = velocityImpulse;
            cc.rhsPenetration = penetrationImpulse;
        }
        c.cfm = 0.0f * c.jacDiagABInv;
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
        fr.cfm = 0.0f;
    }
    fr.applied = 0.0f;
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
__device__ __forceinline__ void isl_resolve_split_lds(const Local& L, const IslBody* sb, const IslRow& c, float rhsPenetration, float& appliedPush)
{
    if (!rhsPenetration) return;
    const bool two = c.b != kNone;
    const float invMassA = c.invMassA;
    const F3 aPush = L.get(c.a, 2), aTurn = L.get(c.a, 3);
    float deltaImpulse = rhsPenetration - appliedPush * c.cfm;
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
    if (sum < 0.0f) {
        deltaImpulse = 0.0f - appliedPush;
        appliedPush = 0.0f;
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

// contact points of the pairs body i of the sorted list owns (it is their lower entity)
__device__ uint32_t isl_pair_points(const IslandParams& ip, uint32_t i)
{
    const uint32_t entity = ip.entity_of_slot[ip.body_slot[i]];
    uint32_t n = 0;
    for (uint32_t k = ip.pair_first[i]; k < ip.n_pairs && static_cast<uint32_t>(ip.keys[k] >> 32) == entity; ++k) {
        n += ip.man[static_cast<uint64_t>(k) * kBoxManifoldWords];
    }
    return n;
}

// convertContacts for body i: its plane manifold, its manifolds with obstacles (ascending entity), its pairs with Dynamic boxes of higher
// entity (ascending) — rows j, j + 1, ... of the island; returns the row after its last
template <bool WARM>
__device__ uint32_t isl_build_body_rows(const WorldView& w, const GroundParams& g, const IslandParams& ip, IslBody* sb, uint32_t i, float invTimeStep,
                                        IslRow* normalRow, IslRow* frictionRow, IslRowCold* coldRow, uint32_t j, uint32_t only = kNone)
{
    // (only != kNone: just that row of the island's list — k_island_rows has a thread for every row)
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
                if (only != kNone && j != only) {
                    j++;
                    continue;
                }
                const F3 worldA = xform_point(basis, pos, F3{mp[8 * k], mp[8 * k + 1], mp[8 * k + 2]});
                isl_add_contact<WARM>(sb, normalRow[j], frictionRow[j], coldRow[j], i, kNone, invTimeStep, worldA, F3{0.0f, 0.0f, 0.0f}, F3{0.0f, 1.0f, 0.0f}, mp[8 * k + 5],
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
                    if (only != kNone && j != only) {
                        j++;
                        continue;
                    }
                    const F3 worldA = xform_point(basis, pos, bp_get3(c, 0));
                    isl_add_contact<WARM>(sb, normalRow[j], frictionRow[j], coldRow[j], i, kNone, invTimeStep, worldA, F3{0.0f, 0.0f, 0.0f}, bp_get3(c, 6), c[9], combinedFriction,
                                    combinedRestitution, c + 10, 1u);
                    j++;
                }
            }
        }
        const uint64_t owner = static_cast<uint64_t>(ip.entity_of_slot[slot]) << 32;
        const uint32_t lo = ip.pair_first[i]; // (k_island_flags looked it up: a bisection here is 17 loads that wait for each other)
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
                if (only != kNone && j != only) {
                    j++;
                    continue;
                }
                const F3 worldA = xform_point(basis, pos, bp_get3(c, 0));
                const F3 worldB = xform_point_b(basis_b, sb[ib].origin, bp_get3(c, 3));
                isl_add_contact<WARM>(sb, normalRow[j], frictionRow[j], coldRow[j], i, ib, invTimeStep, worldA, worldB, bp_get3(c, 6), c[9], combinedFriction, combinedRestitution,
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

// convertBodies for every body of an island that stays awake, one thread a body (the solver threads would do it one body after the
// other), and the contact points the body brings into its island's row list: its own manifolds' and those of the pairs it owns
template <bool BASIS>
__global__ void __launch_bounds__(64) k_island_bodies(WorldView w, GroundParams g, IslandParams ip)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ip.n_bodies) return;
    uint32_t n = 0;
    if (w.cinfo[ip.body_slot[i]] & kCiIsland) n = isl_prepare_body<BASIS>(w, g, ip, static_cast<IslBody*>(ip.solver_bodies), i) + isl_pair_points(ip, i);
    ip.row_count[i] = n;
}

// convertContacts, one thread a ROW: rows row_first[i] .. of body i (row_first: the exclusive sum of the counts over the sorted body
// list, so an island's rows are one run of the two arrays), without the warm start — that is order-dependent and the solvers' first sweep
template <bool BASIS>
__global__ void __launch_bounds__(64) k_island_rows(WorldView w, GroundParams g, IslandParams ip)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t last = ip.n_bodies - 1u;
    if (j >= ip.row_first[last] + ip.row_count[last]) return;
    if (j >= ip.row_cap / 2u) {
        if (j == ip.row_cap / 2u) atomicOr(&ip.counts[3], 1u); // (cannot happen: the host sizes the arrays for every point the manifolds can hold)
        return;
    }
    // the body whose run holds row j: the last one that starts at or before it (bodies without rows share their successor's start)
    uint32_t lo = 0, hi = ip.n_bodies;
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (ip.row_first[mid] <= j) lo = mid;
        else hi = mid;
    }
    IslRow* normalRow = static_cast<IslRow*>(ip.rows);
    isl_build_body_rows<false>(w, g, ip, static_cast<IslBody*>(ip.solver_bodies), lo, 1.0f / g.dt, normalRow, normalRow + ip.row_cap / 2u,
                               static_cast<IslRowCold*>(ip.rows_cold), ip.row_first[lo], j);
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
    // ---- the island's rows: k_island_rows built them at rows row_first[first] .. of the two arrays
    const uint32_t row0 = ip.row_first[first];
    const uint32_t n_points = ip.row_first[end - 1u] + ip.row_count[end - 1u] - row0;
    const bool small = end - first <= kIslLdsBodies && n_points <= kIslLdsPoints;
    if (!MID && !small) {
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
    if (row0 + n_points > ip.row_cap / 2u) return; // (reported by k_island_rows)
    IslRow* normalRow = static_cast<IslRow*>(ip.rows) + row0;
    IslRow* frictionRow = static_cast<IslRow*>(ip.rows) + ip.row_cap / 2u + row0;
    IslRowCold* coldRow = static_cast<IslRowCold*>(ip.rows_cold) + row0;
    // ---- solveGroupCacheFriendlySplitImpulseIterations, solveGroupCacheFriendlyIterations
    if (!MID && small) {
        const IslLocal L{s_isl + (threadIdx.x & 63u), first};
        for (uint32_t i = first; i < end; ++i) {
            for (uint32_t f = 0; f < 4u; ++f) L.set(i, f, F3{0.0f, 0.0f, 0.0f});
        }
        for (uint32_t r = 0; r < n_points; ++r) { // (the warm start, in the rows' order)
            const IslRow c = normalRow[r];
            isl_warm_start_lds(L, sb, c);
            L.row(r, 0) = c.applied;
            L.row(r, 1) = 0.0f;
            L.row(r, 2) = 0.0f;
        }
        // (a row's constants are requested one row ahead: the load of row r + 1 is in flight while row r is resolved — what a sweep
        //  waits for is then the LDS round trip of the bodies it shares with the row before, not a global load per row)
        for (int it = 0; it < kIterations; ++it) {
            bool any = false;
            for (uint32_t r = 0; r < n_points; ++r) any = any || coldRow[r].rhsPenetration != 0.0f;
            if (!any) break; // (no row takes the split impulse: every sweep would return at its first test)
            IslRow cur = normalRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = normalRow[r + 1 < n_points ? r + 1 : r];
                isl_resolve_split_lds(L, sb, cur, coldRow[r].rhsPenetration, L.row(r, 1));
                cur = nxt;
            }
        }
        for (int it = 0; it < kIterations; ++it) {
            if (n_points == 0) break;
            IslRow cur = normalRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = normalRow[r + 1 < n_points ? r + 1 : r];
                isl_resolve_row_lds(L, sb, cur, L.row(r, 0), 0.0f, 1e10f, false);
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
            for (uint32_t f = 0; f < 4u; ++f) L.set(i, f, F3{0.0f, 0.0f, 0.0f});
        }
        for (uint32_t r = 0; r < n_points; ++r) isl_warm_start_lds(L, sb, normalRow[r]); // (the warm start, in the rows' order)
        for (int it = 0; it < kIterations; ++it) {
            bool any = false;
            for (uint32_t r = 0; r < n_points; ++r) any = any || coldRow[r].rhsPenetration != 0.0f;
            if (!any) break;
            IslRow cur = normalRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = normalRow[r + 1 < n_points ? r + 1 : r];
                const float rhsPenetration = coldRow[r].rhsPenetration;
                if (rhsPenetration) {
                    float appliedPush = coldRow[r].appliedPush;
                    isl_resolve_split_lds(L, sb, cur, rhsPenetration, appliedPush);
                    coldRow[r].appliedPush = appliedPush;
                }
                cur = nxt;
            }
        }
        for (int it = 0; it < kIterations; ++it) {
            if (n_points == 0) break;
            IslRow cur = normalRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = normalRow[r + 1 < n_points ? r + 1 : r];
                isl_resolve_row_lds(L, sb, cur, cur.applied, 0.0f, 1e10f, false);
                normalRow[r].applied = cur.applied;
                cur = nxt;
            }
            cur = frictionRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = frictionRow[r + 1 < n_points ? r + 1 : r];
                const float totalImpulse = normalRow[r].applied;
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
        for (uint32_t r = 0; r < n_points; ++r) isl_warm_start(sb, normalRow[r]);
        for (int it = 0; it < kIterations; ++it) {
            bool any = false;
            for (uint32_t r = 0; r < n_points; ++r) any = any || coldRow[r].rhsPenetration != 0.0f;
            if (!any) break;
            IslRow cur = normalRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = normalRow[r + 1 < n_points ? r + 1 : r];
                const float rhsPenetration = coldRow[r].rhsPenetration;
                if (rhsPenetration) {
                    float appliedPush = coldRow[r].appliedPush;
                    isl_resolve_split(sb, cur, rhsPenetration, appliedPush);
                    coldRow[r].appliedPush = appliedPush;
                }
                cur = nxt;
            }
        }
        for (int it = 0; it < kIterations; ++it) {
            if (n_points == 0) break;
            IslRow cur = normalRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = normalRow[r + 1 < n_points ? r + 1 : r];
                isl_resolve_row(sb, cur, 0.0f, 1e10f, false);
                normalRow[r].applied = cur.applied;
                cur = nxt;
            }
            cur = frictionRow[0];
            for (uint32_t r = 0; r < n_points; ++r) {
                const IslRow nxt = frictionRow[r + 1 < n_points ? r + 1 : r];
                const float totalImpulse = normalRow[r].applied;
                if (totalImpulse > 0.0f) {
                    isl_resolve_row(sb, cur, -(cur.friction * totalImpulse), cur.friction * totalImpulse, true);
                    frictionRow[r].applied = cur.applied;
                }
                cur = nxt;
            }
        }
    }
    // ---- solveGroupCacheFriendlyFinish
    for (uint32_t r = 0; r < n_points; ++r) {
        coldRow[r].out[0] = normalRow[r].applied;
        coldRow[r].out[coldRow[r].lateral_at] = frictionRow[r].applied;
    }
    for (uint32_t i = first; i < end; ++i) isl_finish_body<BASIS>(w, g, sb, i);
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
    __shared__ uint32_t s_ticket, s_total, s_depth, s_ints_at, s_fail, s_any;
    IslBody* sb = static_cast<IslBody*>(ip.solver_bodies);
    const uint32_t tid = threadIdx.x;
    const int kIterations = static_cast<int>(ip.iterations);
    for (;;) {
        __syncthreads();
        if (tid == 0) s_ticket = atomicAdd(&ip.counts[5], 1u);
        __syncthreads();
        const uint32_t t = s_ticket;
        if (t >= ip.counts[4]) return;
        const uint32_t first = ip.big_list[2u * t], end = ip.big_list[2u * t + 1u], nb = end - first;
        // (k_island_bodies prepared the bodies, k_island_rows built the rows: rows row_first[first] .. of the two arrays)
        const uint32_t row0 = ip.row_first[first];
        const uint32_t P = ip.row_first[end - 1u] + ip.row_count[end - 1u] - row0;
        if (P == 0) { // (bodies in each other's AABBs, nothing touches: gravity and the gyroscopic term only)
            for (uint32_t i = first + tid; i < end; i += 256u) isl_finish_body<BASIS>(w, g, sb, i);
            continue;
        }
        if (tid == 0) {
            s_fail = 0u;
            s_ints_at = atomicAdd(&ip.counts[6], 4u * P + 8u);
            if (row0 + P > ip.row_cap / 2u || s_ints_at + 4u * P + 8u > ip.int_cap) {
                atomicOr(&ip.counts[3], 1u); // (cannot happen: both pools hold every point the manifolds can hold)
                s_fail = 1u;
            }
        }
        __syncthreads();
        if (s_fail) continue;
        IslRow* normalRow = static_cast<IslRow*>(ip.rows) + row0;
        IslRow* frictionRow = static_cast<IslRow*>(ip.rows) + ip.row_cap / 2u + row0;
        IslRowCold* coldRow = static_cast<IslRowCold*>(ip.rows_cold) + row0;
        uint32_t* level = ip.ints + s_ints_at;  // [P] level of row r (1 ..)
        uint32_t* order = level + P;            // [P] rows in level order
        uint32_t* start = order + P;            // [depth + 2] first entry of level l in `order`
        uint32_t* cursor = start + P + 4u;      // [depth + 2]
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
                any |= coldRow[r].rhsPenetration != 0.0f ? 1u : 0u;
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
                    const float rhsPenetration = coldRow[r].rhsPenetration;
                    if (rhsPenetration) {
                        float appliedPush = coldRow[r].appliedPush;
                        if (bodies_in_lds) isl_resolve_split_lds(L, sb, c, rhsPenetration, appliedPush);
                        else isl_resolve_split(sb, c, rhsPenetration, appliedPush);
                        coldRow[r].appliedPush = appliedPush;
                    }
                });
            }
        }
        // solveGroupCacheFriendlyIterations: all contact rows, then all friction rows
        for (int it = 0; it < kIterations; ++it) {
            sweep(normalRow, [&](IslRow& c, uint32_t r) {
                if (bodies_in_lds) isl_resolve_row_lds(L, sb, c, c.applied, 0.0f, 1e10f, false);
                else isl_resolve_row(sb, c, 0.0f, 1e10f, false);
                normalRow[r].applied = c.applied;
            });
            sweep(frictionRow, [&](IslRow& c, uint32_t r) {
                const float totalImpulse = normalRow[r].applied;
                if (totalImpulse > 0.0f) {
                    const float lower = -(c.friction * totalImpulse), upper = c.friction * totalImpulse;
                    if (bodies_in_lds) isl_resolve_row_lds(L, sb, c, c.applied, lower, upper, true);
                    else isl_resolve_row(sb, c, lower, upper, true);
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
            coldRow[r].out[0] = normalRow[r].applied;
            coldRow[r].out[coldRow[r].lateral_at] = frictionRow[r].applied;
        }
        for (uint32_t i = first + tid; i < end; i += 256u) isl_finish_body<BASIS>(w, g, sb, i);
    }
}

} // namespace

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

size_t island_scan_bytes(uint32_t n)
{
    size_t bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), static_cast<int>(n));
    return bytes;
}

hipError_t launch_island_solve(hipStream_t stream, const WorldView& w, const GroundParams& g, const IslandParams& ip, bool bullet_basis)
{
    if (ip.n_bodies == 0) return hipSuccess;
    hipLaunchKernelGGL(k_island_flags, dim3((ip.n_bodies + 255u) / 256u), dim3(256), 0, stream, w, ip);
    const dim3 grid((ip.n_bodies + 63u) / 64u), block(64);
    size_t scan_bytes = ip.scan_tmp_bytes;
    const dim3 row_grid((ip.row_cap / 2u + 64u) / 64u); // (a thread for every row the arrays can hold; those past the last row leave at once)
    const dim3 mid_grid((ip.n_bodies / (kIslLdsBodies + 1u) + 64u) / 64u); // (an island on the mid list has more than kIslLdsBodies bodies)
    if (bullet_basis) {
        hipLaunchKernelGGL(k_island_own<true>, grid, block, 0, stream, w, g, ip);
        hipLaunchKernelGGL(k_island_bodies<true>, grid, block, 0, stream, w, g, ip);
        (void)hipcub::DeviceScan::ExclusiveSum(ip.scan_tmp, scan_bytes, ip.row_count, ip.row_first, static_cast<int>(ip.n_bodies), stream);
        hipLaunchKernelGGL(k_island_rows<true>, row_grid, block, 0, stream, w, g, ip);
        hipLaunchKernelGGL((k_island_solve<true, false>), grid, block, 0, stream, w, g, ip);
        hipLaunchKernelGGL((k_island_solve<true, true>), mid_grid, block, 0, stream, w, g, ip);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_island_solve_big<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kIslBigLdsBytes));
        hipLaunchKernelGGL(k_island_solve_big<true>, dim3(256), dim3(256), kIslBigLdsBytes, stream, w, g, ip);
    } else {
        hipLaunchKernelGGL(k_island_own<false>, grid, block, 0, stream, w, g, ip);
        hipLaunchKernelGGL(k_island_bodies<false>, grid, block, 0, stream, w, g, ip);
        (void)hipcub::DeviceScan::ExclusiveSum(ip.scan_tmp, scan_bytes, ip.row_count, ip.row_first, static_cast<int>(ip.n_bodies), stream);
        hipLaunchKernelGGL(k_island_rows<false>, row_grid, block, 0, stream, w, g, ip);
        hipLaunchKernelGGL((k_island_solve<false, false>), grid, block, 0, stream, w, g, ip);
        hipLaunchKernelGGL((k_island_solve<false, true>), mid_grid, block, 0, stream, w, g, ip);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_island_solve_big<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kIslBigLdsBytes));
        hipLaunchKernelGGL(k_island_solve_big<false>, dim3(256), dim3(256), kIslBigLdsBytes, stream, w, g, ip);
    }
    return hipGetLastError();
}

} // namespace bge
