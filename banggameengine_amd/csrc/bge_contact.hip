// bge_contact.hip — the static ground plane of the reference's physics world and Bullet's contact handling for it, one
// thread per body.
//
// The reference adds an infinite static plane at y = 0 (friction 1, restitution 0) to every world
// (src/physics/PhysicsSystem.cpp:149-166) and steps with Bullet's default narrowphase and sequential-impulse solver
// (:122-128, :863).  Bodies do not collide with each other on this path: every Dynamic body is an island of its own whose
// only manifold is the one with the plane, so the step is independent per body and maps onto one thread:
//   k_ground   performDiscreteCollisionDetection for (plane, body) — btConvexPlaneCollisionAlgorithm: one contact per step at
//              the shape's support vertex, kept in a 4-point persistent manifold (nearest-point replacement, area-based
//              eviction, refresh with removal beyond the breaking threshold) — and solveConstraints for the island {body}:
//              external force impulse, implicit gyroscopic impulse, contact and friction rows with warm starting,
//              10 split-impulse iterations, 10 velocity iterations, write-back of velocities, impulses and the pushed pose.
//              A body without a contact and without angular velocity is left to k_tick's plain update (the same arithmetic).
//   k_tick     then integrates the pose with the velocities found here (cinfo bit kCiSolved: gravity is already in them).
// The arithmetic — operation order, where Bullet's compiled row solvers fuse multiply-adds, which dot products add in
// which order — is oracle/contact_ref.h's (that header states what is pinned by the reference's exe and what is not);
// tests/test_gpu_parity.py compares every body bit for bit.  Heavy per-thread state (four contact rows and four friction
// rows) makes this a register-hungry kernel; it touches only bodies at the ground, so it is sized for correctness first.
// Also here: Dynamic boxes on the Static / Kinematic box colliders of the scene (k_obstacles, k_obstacle_grid, k_contact_boxes: one island
// per body, the rows in scratch memory).  The device functions all of this is made of are in bge_contact_device.hpp; Dynamic
// boxes against EACH OTHER (pair cache, simulation islands, their solvers) are bge_island.hip.
#include <hip/hip_runtime.h>

#include "bge_contact_device.hpp"

namespace bge {

namespace {

// One workgroup builds the whole index: bounds, counts per cell (LDS), scan, fill.  A few thousand obstacles are microseconds.
__global__ void __launch_bounds__(1024) k_obstacle_grid(GroundParams g)
{
    __shared__ uint32_t s_cnt[kObstacleGridAxis * kObstacleGridAxis];
    __shared__ uint32_t s_scan[1024];
    __shared__ float s_red[4][16];
    __shared__ uint32_t s_wide, s_bad;
    const uint32_t tid = threadIdx.x, K = g.n_obstacles;
    uint32_t* hdr = g.obstacle_grid;
    float mnx = INFINITY, mnz = INFINITY, mxx = -INFINITY, mxz = -INFINITY;
    for (uint32_t k = tid; k < K; k += 1024u) {
        const ObstacleRec& o = g.obstacles[k];
        if (!o.live || !(o.aabb[0] <= o.aabb[3]) || !(o.aabb[2] <= o.aabb[5])) continue;
        mnx = fminf(mnx, o.aabb[0]);
        mxx = fmaxf(mxx, o.aabb[3]);
        mnz = fminf(mnz, o.aabb[2]);
        mxz = fmaxf(mxz, o.aabb[5]);
    }
    for (int d = 32; d > 0; d >>= 1) {
        mnx = fminf(mnx, __shfl_down(mnx, d));
        mnz = fminf(mnz, __shfl_down(mnz, d));
        mxx = fmaxf(mxx, __shfl_down(mxx, d));
        mxz = fmaxf(mxz, __shfl_down(mxz, d));
    }
    if ((tid & 63u) == 0u) {
        s_red[0][tid >> 6] = mnx;
        s_red[1][tid >> 6] = mnz;
        s_red[2][tid >> 6] = mxx;
        s_red[3][tid >> 6] = mxz;
    }
    if (tid == 0) s_wide = s_bad = 0u;
    for (uint32_t c = tid; c < kObstacleGridAxis * kObstacleGridAxis; c += 1024u) s_cnt[c] = 0u;
    __syncthreads();
    mnx = mnz = INFINITY;
    mxx = mxz = -INFINITY;
    for (int k = 0; k < 16; ++k) {
        mnx = fminf(mnx, s_red[0][k]);
        mnz = fminf(mnz, s_red[1][k]);
        mxx = fmaxf(mxx, s_red[2][k]);
        mxz = fmaxf(mxz, s_red[3][k]);
    }
    int n = 1;
    while (n < static_cast<int>(kObstacleGridAxis) && static_cast<uint32_t>(n * n) < K) n *= 2;
    const bool any = mnx <= mxx && mnz <= mxz && mxx - mnx < INFINITY && mxz - mnz < INFINITY;
    const float ux = any && mxx > mnx ? static_cast<float>(n) / (mxx - mnx) : 0.0f, uz = any && mxz > mnz ? static_cast<float>(n) / (mxz - mnz) : 0.0f;
    // pass 1: counts (an obstacle that covers more than 64 cells goes on the wide list instead)
    for (uint32_t k = tid; k < K; k += 1024u) {
        const ObstacleRec& o = g.obstacles[k];
        if (!any || !o.live || !(o.aabb[0] <= o.aabb[3]) || !(o.aabb[2] <= o.aabb[5])) continue;
        const int x0 = obs_cell(o.aabb[0], mnx, ux, n), x1 = obs_cell(o.aabb[3], mnx, ux, n), z0 = obs_cell(o.aabb[2], mnz, uz, n), z1 = obs_cell(o.aabb[5], mnz, uz, n);
        if ((x1 - x0 + 1) * (z1 - z0 + 1) > 64) {
            const uint32_t at = atomicAdd(&s_wide, 1u);
            if (at < kObstacleGridWide) hdr[8 + at] = k;
            else s_bad = 1u;
            continue;
        }
        for (int z = z0; z <= z1; ++z) {
            for (int x = x0; x <= x1; ++x) atomicAdd(&s_cnt[z * n + x], 1u);
        }
    }
    __syncthreads();
    // exclusive scan of the n * n counts: four cells per thread, then the 1024 partial sums
    const uint32_t cells = static_cast<uint32_t>(n * n);
    uint32_t mine[4], sum = 0;
    for (int j = 0; j < 4; ++j) {
        const uint32_t c = tid * 4u + j;
        mine[j] = c < cells ? s_cnt[c] : 0u;
        sum += mine[j];
    }
    s_scan[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024u; d <<= 1) {
        const uint32_t v = tid >= d ? s_scan[tid - d] : 0u;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
    }
    uint32_t run = s_scan[tid] - sum;
    const uint32_t total = s_scan[1023];
    uint32_t* start = hdr + kObstacleGridStart;
    for (int j = 0; j < 4; ++j) {
        const uint32_t c = tid * 4u + j;
        if (c < cells) {
            start[c] = run;
            s_cnt[c] = run; // (from here on: the cell's write cursor)
        }
        run += mine[j];
    }
    if (tid == 0) start[cells] = total;
    __syncthreads();
    const bool fits = total <= g.obstacle_grid_cap && !s_bad;
    if (fits) {
        uint32_t* items = hdr + kObstacleGridItems;
        for (uint32_t k = tid; k < K; k += 1024u) {
            const ObstacleRec& o = g.obstacles[k];
            if (!any || !o.live || !(o.aabb[0] <= o.aabb[3]) || !(o.aabb[2] <= o.aabb[5])) continue;
            const int x0 = obs_cell(o.aabb[0], mnx, ux, n), x1 = obs_cell(o.aabb[3], mnx, ux, n), z0 = obs_cell(o.aabb[2], mnz, uz, n), z1 = obs_cell(o.aabb[5], mnz, uz, n);
            if ((x1 - x0 + 1) * (z1 - z0 + 1) > 64) continue;
            for (int z = z0; z <= z1; ++z) {
                for (int x = x0; x <= x1; ++x) items[atomicAdd(&s_cnt[z * n + x], 1u)] = k;
            }
        }
    }
    if (tid == 0) {
        hdr[0] = fits ? 1u : 0u;
        hdr[1] = static_cast<uint32_t>(n);
        hdr[2] = s_wide < kObstacleGridWide ? s_wide : kObstacleGridWide;
        hdr[4] = __float_as_uint(mnx);
        hdr[5] = __float_as_uint(mnz);
        hdr[6] = __float_as_uint(ux);
        hdr[7] = __float_as_uint(uz);
    }
}

// ---- two launches per sub-step
// ground_body needs 246 VGPRs (two waves per SIMD) — and most bodies of a scene need none of it: they sleep, or are nowhere
// near the plane.  As ONE kernel over all slots (the first version) even those paid for the solver's occupancy: two waves per
// SIMD cannot keep enough loads in flight, and the tests sat behind five dependent round trips (flags -> contact word ->
// palette -> shape -> position).  1 M bodies, per tick on top of the 24.5 us tick: asleep +38 us, airborne +22 us.
//   k_ground_select  256 threads, a handful of registers: the teleport rule of PhysicsSystem::Update for dirty bodies (what
//                    k_pose_only did as a third launch), then the tests in two batches of loads — flags + contact word +
//                    deactivation record; shape + position for those still in — and the slots that need the solver appended
//                    to a list (one atomic per wave)
//   k_ground         a resident-sized grid walks the list: full waves of bodies that are all in contact
// The list's order depends on the atomics, the results do not: a body touches nothing but its own records.
template <bool BASIS>
__global__ void __launch_bounds__(256) k_ground_select(WorldView w, GroundParams g)
{
    const uint64_t slot64 = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    const bool in_range = slot64 < g.n_slots;
    const uint32_t slot = in_range ? static_cast<uint32_t>(slot64) : 0u;
    uint32_t f = w.flags[slot];
    const uint32_t ci0 = w.cinfo[slot];
    uint32_t dz = w.deact[slot]; // (read whether or not kDrowsy says it is meaningful: one round trip instead of two)
    asm volatile("" : "+v"(f), "+v"(dz));
    bool need = in_range;
    const uint32_t type = f & kTypeMask;
    if (need && g.repose && (f & kValid) && type != 0 && (f & (kTDirty | kBDirty))) {
        // EnsureRigidBody / SyncKinematicBodiesToPhysics before stepSimulation (PhysicsSystem.cpp:952-989): pose from the Transform,
        // zero velocities — k_pose_only's re-pose, for the body types it applies to; the tick kernel that follows (no_repose) marks
        // Dynamic transforms dirty and clears kBDirty as always
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
    // (a body that wants to sleep is still collided this step — only a sleeping one is skipped)
    // (a body of an island of several bodies that stays awake is collided and solved by the island kernels, which ran before)
    const bool awake = in_range && type == 2u && !((f & kDrowsy) && dz == kDeactSleeping) && !(ci0 & kCiIsland);
    // Static / Kinematic box colliders on: a Dynamic BOX that holds manifolds with boxes, or whose reach (conservative: the L1 norm
    // of its half extents bounds its AABB at any orientation, plus this step's motion, plus Bullet's 0.02) touches an obstacle's
    // fed AABB, goes to k_contact_boxes — which decides the pairs exactly and handles the plane for that body too
    bool boxes = false;
    if (g.n_obstacles != 0u || (ci0 & kCiBoxes)) {
        if (awake && !(ci0 & kCiCapsule)) {
            boxes = (ci0 & kCiBoxes) != 0;
            if (!boxes) {
                const float4 cs = w.cshape[slot];
                const F3 pos = ld3(w.pos, slot);
                const F3 v = ld3(w.vel, slot);
                const float reach = (__builtin_fabsf(cs.x) + __builtin_fabsf(cs.y) + __builtin_fabsf(cs.z)) * 1.01f + 0.05f;
                const float rx = reach + __builtin_fabsf(v.x) * g.dt * 1.01f, ry = reach + __builtin_fabsf(v.y) * g.dt * 1.01f,
                            rz = reach + __builtin_fabsf(v.z) * g.dt * 1.01f;
                for_each_obstacle_near(g, pos.x - rx, pos.x + rx, pos.z - rz, pos.z + rz, [&](uint32_t k) {
                    const float* bb = g.obstacles[k].aabb;
                    boxes = pos.x - rx <= bb[3] && pos.x + rx >= bb[0] && pos.y - ry <= bb[4] && pos.y + ry >= bb[1] && pos.z - rz <= bb[5] &&
                            pos.z + rz >= bb[2];
                    return boxes;
                });
            }
        }
    }
    need = awake && !boxes && g.plane != 0u && (ci0 & kCiGroundMask) != 0;
    if (__any(need)) {
        const uint32_t n = (ci0 >> kCiCountShift) & 7u;
        if (need && n == 0u && !(f & kSpin)) {
            // cheap reject, as in ground_body: no vertex of the shape can be within the breaking threshold of the plane
            const float4 cs = w.cshape[slot];
            const F3 pos = ld3(w.pos, slot);
            CtShape shape;
            shape.capsule = (ci0 & kCiCapsule) != 0;
            shape.dims = F3{cs.x, cs.y, cs.z};
            const float reach = (__builtin_fabsf(cs.x) + __builtin_fabsf(cs.y) + __builtin_fabsf(cs.z)) * 1.01f + 0.01f;
            if (pos.y - reach > ct_breaking_threshold(shape)) need = false;
        }
    }
    {
        const unsigned long long mb = __ballot(boxes);
        if (mb != 0) { // (rare: one list, one atomic per wave)
            const uint32_t lane = threadIdx.x & 63u;
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&g.box_count[0], static_cast<uint32_t>(__popcll(mb)));
            base = __shfl(base, 0, 64);
            if (boxes) g.box_list[base + static_cast<uint32_t>(__popcll(mb & ((1ull << lane) - 1ull)))] = slot;
        }
    }
    const unsigned long long m = __ballot(need);
    if (m == 0) return;
    // one atomic per wave, on the counter of this workgroup's shard: a single counter word takes ~10^8 atomics a second, and
    // 15,625 waves of a million resting bodies queued on it for 130 us.  Workgroup b appends to shard b % kGroundShards, whose
    // segment holds every slot those workgroups could ever send: it cannot overflow.
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t shard = blockIdx.x % kGroundShards;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&g.list_count[16u * shard], static_cast<uint32_t>(__popcll(m)));
    base = __shfl(base, 0, 64);
    const uint32_t at = base + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
    if (need && at < g.shard_cap) g.list[static_cast<uint64_t>(shard) * g.shard_cap + at] = slot; // (the bound cannot bite while the counts are reset: a guard, not a path)
}

template <bool BASIS>
#ifndef BGE_GROUND_MIN_BLOCKS
#define BGE_GROUND_MIN_BLOCKS 2 /* waves per SIMD the register budget is set for: 2 -> 246 VGPRs, no scratch; 4 -> 128 VGPRs and spills (measured slower) */
#endif
__global__ void __launch_bounds__(128, BGE_GROUND_MIN_BLOCKS) k_ground(WorldView w, GroundParams g)
{
    // workgroup b works on shard b % kGroundShards, together with the other gridDim.x / kGroundShards workgroups of that shard
    const uint32_t shard = blockIdx.x % kGroundShards;
    const uint32_t n_list = min(g.list_count[16u * shard], static_cast<uint32_t>(g.shard_cap));
    const uint32_t* list = g.list + static_cast<uint64_t>(shard) * g.shard_cap;
    const uint32_t step = (gridDim.x / kGroundShards) * blockDim.x;
#ifndef BGE_GROUND_EMPTY /* timing experiment: the launch without the solver (and so without scratch) */
    for (uint32_t i = (blockIdx.x / kGroundShards) * blockDim.x + threadIdx.x; i < n_list; i += step) {
        const uint32_t slot = list[i];
        if (slot < g.n_slots) ground_body<BASIS>(w, g, slot);
    }
#else
    if (n_list == 0xffffffffu) w.cinfo[list[step]] = 0;
#endif
    // the workgroup that draws the shard's last ticket empties its list for the next sub-step's k_ground_select: by then every
    // workgroup of the shard has read the count (it did so before it drew its own ticket).  (One ticket word for all 1024
    // workgroups made an EMPTY launch take 13.6 us: a thousand atomics on one address.)
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(&g.list_count[16u * shard + 1u], 1u) == gridDim.x / kGroundShards - 1u) {
            g.list_count[16u * shard] = 0;
            g.list_count[16u * shard + 1u] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Round 3: Dynamic boxes on the Static / Kinematic BOX colliders of the scene (Bullet's btBoxBoxCollisionAlgorithm; the reference's
// demo.json "Ground" is one).  oracle/boxbox_ref.h + physics_ref.h (CollideWithBoxes, ct::SolveBody) are the specification; the
// choices Bullet's history-dependence forces are stated there (the Dynamic body is body A; pairs = fed AABBs overlap + filter; at
// most kBoxManifolds manifolds, lowest entity ids).  Capsules against boxes (GJK) are not built.
//   k_obstacles       one thread per Static / Kinematic box body (a compact list the host keeps, ascending entity): pose as Bullet
//                     holds it at this sub-step (a dirty body's from its Transform), fed AABB, material -> ObstacleRec
//   k_ground_select   routes a Dynamic box that holds box manifolds, or whose reach touches an obstacle's AABB, to box_list
//   k_contact_boxes   one thread per listed body: exact pairs, box-box detector into the body's persistent manifolds (rows of
//                     bmanifold, kept in global memory), the plane manifold as k_ground has it, then ONE solver for the island
//                     {body}: plane rows, then every box manifold's rows in ascending entity — with rows in scratch memory and
//                     loops, not unrolled registers: this kernel serves the handful of bodies that rest on a static box, and is
//                     sized for correctness (k_ground stays the fast path for everything that touches only the plane).
__global__ void __launch_bounds__(64) k_obstacles(WorldView w, GroundParams g)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= g.n_obstacles) return;
    const uint32_t slot = g.obstacle_slots[k];
    ObstacleRec r;
    const uint32_t f = w.flags[slot];
    const uint32_t type = f & kTypeMask;
    const uint32_t ci = w.cinfo[slot];
    r.live = (type == 1u || type == 3u) && !(ci & kCiCapsule) ? 1u : 0u;
    const F3 pos = ld3(w.pos, slot);
    // SyncKinematicBodiesToPhysics runs before the step: a dirty body is posed from its Transform (k_ground_select / k_tick store that
    // quaternion; this kernel runs before them)
    const bool repose = g.repose && (f & kValid) && (f & (kTDirty | kBDirty));
    const Q4 q = repose ? bt_quat_from_transform_euler(ld3(w.euler, slot)) : ld4(w.quat, slot);
    const M3 basis = bt_mat_from_quat(q);
    const float4 cs = w.cshape[slot];
    CtShape shape;
    shape.capsule = false;
    shape.dims = F3{cs.x, cs.y, cs.z};
    float mn[3], mx[3];
    bt_aabb_of_pose(pos, basis, ld3(w.half_extent, slot), mn, mx);
    r.origin[0] = pos.x; r.origin[1] = pos.y; r.origin[2] = pos.z;
    r.half[0] = cs.x; r.half[1] = cs.y; r.half[2] = cs.z;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) r.basis[3 * i + j] = basis.m[i][j];
        r.aabb[i] = mn[i];
        r.aabb[3 + i] = mx[i];
    }
    r.friction = w.cfriction[slot];
    r.restitution = w.crestitution[slot];
    r.breaking = ct_breaking_threshold(shape);
    r.entity = g.entity_of_slot[slot];
    r.group = w.group[slot];
    r.mask = w.mask[slot];
    r.generation = g.obstacle_gen[k];
    r.pad[0] = r.pad[1] = r.pad[2] = 0u;
    g.obstacles[k] = r;
}

template <bool BASIS>
__global__ void __launch_bounds__(64) k_contact_boxes(WorldView w, GroundParams g)
{
    const uint32_t n_list = g.box_count[0];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_list; i += gridDim.x * blockDim.x) {
        const uint32_t slot = g.box_list[i];
        if (slot < g.n_slots) contact_body<BASIS>(w, g, slot);
    }
    // the workgroup that finishes last empties the list for the next sub-step's k_ground_select (every workgroup has read the count)
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(&g.box_count[1], 1u) == gridDim.x - 1u) {
            g.box_count[0] = 0;
            g.box_count[1] = 0;
        }
    }
}

} // namespace

hipError_t launch_ground(hipStream_t stream, const WorldView& w, const GroundParams& g, bool bullet_basis)
{
    if (g.n_slots == 0) return hipSuccess;
    const dim3 sgrid(static_cast<uint32_t>((g.n_slots + 255) / 256)), sblock(256);
    // what is resident: 2 x BGE_GROUND_MIN_BLOCKS workgroups of 128 threads on each of the 256 CUs (a multiple of the shard count)
    const dim3 grid(512u * BGE_GROUND_MIN_BLOCKS), block(128);
    const bool boxes = g.box_list != nullptr; // Static / Kinematic box colliders are on
    if (!g.obstacles_ready) {
        if (boxes && g.n_obstacles) hipLaunchKernelGGL(k_obstacles, dim3((g.n_obstacles + 63u) / 64u), dim3(64), 0, stream, w, g);
        if (boxes && g.obstacle_grid) hipLaunchKernelGGL(k_obstacle_grid, dim3(1), dim3(1024), 0, stream, g);
    }
    if (bullet_basis) {
        hipLaunchKernelGGL(k_ground_select<true>, sgrid, sblock, 0, stream, w, g);
        if (g.plane) hipLaunchKernelGGL(k_ground<true>, grid, block, 0, stream, w, g);
        if (boxes) hipLaunchKernelGGL(k_contact_boxes<true>, dim3(256), dim3(64), 0, stream, w, g);
    } else {
        hipLaunchKernelGGL(k_ground_select<false>, sgrid, sblock, 0, stream, w, g);
        if (g.plane) hipLaunchKernelGGL(k_ground<false>, grid, block, 0, stream, w, g);
        if (boxes) hipLaunchKernelGGL(k_contact_boxes<false>, dim3(256), dim3(64), 0, stream, w, g);
    }
    return hipGetLastError();
}

hipError_t launch_obstacles(hipStream_t stream, const WorldView& w, const GroundParams& g)
{
    if (g.box_list && g.n_obstacles) hipLaunchKernelGGL(k_obstacles, dim3((g.n_obstacles + 63u) / 64u), dim3(64), 0, stream, w, g);
    if (g.box_list && g.obstacle_grid) hipLaunchKernelGGL(k_obstacle_grid, dim3(1), dim3(1024), 0, stream, g);
    return hipGetLastError();
}

} // namespace bge
