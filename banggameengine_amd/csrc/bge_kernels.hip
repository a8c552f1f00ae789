// bge_kernels.hip — hand-written gfx950 kernels of the world tick.
//
// k_tick<PHYS, XFORM, AABB, NORMAL, BASIS>: one workgroup (4 wave64) per 256-slot tile.
//   PHYS   rigid-body slice of PhysicsSystem::Update for free bodies
//          (src/physics/PhysicsSystem.cpp:952-989 re-pose rule, :863 one Bullet sub-step,
//           :916-950 write-back + mark dirty)
//   XFORM  TransformSystem::Update (src/ecs/TransformSystem.cpp:10-46): local = mtxSRT, then
//          world = parentWorld * local level by level inside the tile, the parents' world matrices
//          staged in LDS (16 KiB per workgroup).  Wave-local tiles (every subtree inside one 64-slot
//          group) run without any workgroup barrier; block tiles use one barrier per level
//   AABB   the per-body AABB Bullet feeds its broadphase (current pose U predicted pose, +0.02)
//   NORMAL the render feed's per-entity normal matrix transpose(inverse(world)) (src/render/Renderer.cpp:633-636)
//   BASIS  Bullet's own orientation bookkeeping for every Dynamic body (BGE_TICK_BULLET_BASIS): basis -> getRotation ->
//          integrateTransform -> setRotation each step, rotationEuler rewritten from it; without it a body with zero angular
//          velocity keeps its orientation and euler triple
//
// Memory plan (all streams indexed by slot, 256 consecutive slots per workgroup):
//   reads   flags 4 B (body type, dirty bits, level, in-tile parent index, mass class), pos/euler/scale 12 B
//           each, vel 12 B (Dynamic bodies only); parent slot 4 B only for nodes whose parent is in an earlier pass
//   writes  pos 12 B + vel 12 B (Dynamic only), world 64 B; flags only when a bit changed
//   The world matrices leave through LDS so that every wave-level store instruction writes 1 KiB of
//   contiguous memory (16 B per lane), whatever the per-node compute layout was.
//   The kernel is HBM-bound (~250 flop against >= 113 B per entity): no MFMA.
#include <hip/hip_runtime.h>

#include "bge_device_math.hpp"
#include "bge_flatten.hpp"
#include "bge_kernels.hpp"

namespace bge {

using namespace dev;

namespace {

// LDS image of the tile's world matrices: node n, row r lives at float4 index n*4 + (r ^ ((n>>2)&3)).
// The XOR spreads the four 64-B rows of consecutive nodes over all 64 banks for ds_read_b128
// (lanes n, n+4, n+8, n+12 of a 16-lane group would otherwise share a bank).
__device__ __forceinline__ uint32_t lds_row(uint32_t n, uint32_t r) { return n * 4u + (r ^ ((n >> 2) & 3u)); }

__device__ __forceinline__ void lds_put(float4* lds, uint32_t n, const float (&m)[16])
{
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) lds[lds_row(n, r)] = make_float4(m[4 * r], m[4 * r + 1], m[4 * r + 2], m[4 * r + 3]);
}

__device__ __forceinline__ void lds_get(const float4* lds, uint32_t n, float (&m)[16])
{
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) {
        const float4 v = lds[lds_row(n, r)];
        m[4 * r] = v.x;
        m[4 * r + 1] = v.y;
        m[4 * r + 2] = v.z;
        m[4 * r + 3] = v.w;
    }
}

// world = parent * local, a row at a time: the parent's row comes from the LDS image (or from memory, for a parent resolved by
// an earlier launch), the product's row goes straight into this node's place in the image.  (bx_mtx_mul on whole matrices held
// 48 registers here — parent, local, product — and with them the kernel's register allocation.)
__device__ __forceinline__ void lds_mul_put(float4* lds, uint32_t parent, uint32_t self, const float (&local)[16])
{
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) lds[lds_row(self, r)] = bx_mtx_mul_row(lds[lds_row(parent, r)], local);
}
__device__ __forceinline__ void world_mul_put(float4* lds, const float* __restrict__ world, uint32_t parent_slot, uint32_t self, const float (&local)[16])
{
    const float4* src = reinterpret_cast<const float4*>(world) + 4ull * parent_slot;
#pragma unroll
    for (uint32_t r = 0; r < 4; ++r) lds[lds_row(self, r)] = bx_mtx_mul_row(src[r], local);
}

// Root table for the per-frame all-gather, filled by the roots themselves (consecutive roots of a tile are
// consecutive lanes, so the 64-byte rows of a wave form one contiguous run).
__device__ __forceinline__ void store_root(float* __restrict__ root_out, uint32_t index, const float (&m)[16])
{
    // compact row, 48 bytes: a root's world matrix is its local matrix (Transform.cpp:32-35), whose fourth column is
    // exactly (0, 0, 0, 1) (bx::mtxSRT) — the gather moves the 12 floats that carry information
    float4* dst = reinterpret_cast<float4*>(root_out) + 3ull * index;
    dst[0] = make_float4(m[0], m[1], m[2], m[4]);
    dst[1] = make_float4(m[5], m[6], m[8], m[9]);
    dst[2] = make_float4(m[10], m[12], m[13], m[14]);
}

// A frozen root (WorldView::frozen): its stored world matrix goes where the freshly built local matrix would have gone — a
// row at a time, so that the rare path costs the kernel no registers.
__device__ __forceinline__ void lds_put_stored(float4* lds, uint32_t n, const float* __restrict__ world, uint32_t slot)
{
    const float4* src = reinterpret_cast<const float4*>(world) + 4ull * slot;
#pragma unroll 1
    for (uint32_t r = 0; r < 4; ++r) lds[lds_row(n, r)] = src[r];
}
__device__ __forceinline__ void store_root_stored(float* __restrict__ root_out, uint32_t index, const float* __restrict__ world, uint32_t slot)
{
    const float* m = world + 16ull * slot;
    float* dst = root_out + 12ull * index; // the packing of store_root
    const int pick[12] = {0, 1, 2, 4, 5, 6, 8, 9, 10, 12, 13, 14};
#pragma unroll 1
    for (int k = 0; k < 12; ++k) dst[k] = m[pick[k]];
}

// Stores of results the tick never reads again (world / normal matrices).  Non-temporal when the host asks for it:
// measured on MI355X, plain stores win while the tick's working set fits the 256 MiB Infinity Cache (1 M entities:
// 22.6 us plain, 24.5 us nt) and nt stores win beyond it (4 M: 95.0 us plain, 89.5 us nt) — the 64 B/entity output
// stream then no longer evicts lines the next tick re-reads.
__device__ __forceinline__ void store_out(float4* p, const float4& v, bool nt)
{
    if (nt) {
        __builtin_nontemporal_store(v.x, &p->x);
        __builtin_nontemporal_store(v.y, &p->y);
        __builtin_nontemporal_store(v.z, &p->z);
        __builtin_nontemporal_store(v.w, &p->w);
    } else {
        *p = v;
    }
}

// LDS hand-over between lanes of ONE wave: the hardware keeps a wave's DS operations in order; the fences
// keep the compiler from moving them across this point.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave64 reductions with DPP (one VALU instruction per step, no LDS): after the six steps lane 63 holds the result.
// row_shr 1/2/4/8 make an inclusive scan inside each row of 16 lanes (min / max are idempotent, so overlapping windows are
// fine); row_bcast:15 carries each row's last lane into the next row (rows 1 and 3), row_bcast:31 lane 31 into rows 2 and 3.
// Lanes that must not contribute pass the identity.  Call with all 64 lanes active.
template <bool IS_MIN> __device__ __forceinline__ float wave_reduce_to_lane63(float v)
{
    const int ident = __float_as_int(IS_MIN ? INFINITY : -INFINITY);
#define BGE_DPP_STEP(ctrl, rmask)                                                                                      \
    {                                                                                                                 \
        const float o = __int_as_float(__builtin_amdgcn_update_dpp(ident, __float_as_int(v), ctrl, rmask, 0xf, false)); \
        v = IS_MIN ? fminf(v, o) : fmaxf(v, o);                                                                        \
    }
    BGE_DPP_STEP(0x111, 0xf) // row_shr:1
    BGE_DPP_STEP(0x112, 0xf) // row_shr:2
    BGE_DPP_STEP(0x114, 0xf) // row_shr:4
    BGE_DPP_STEP(0x118, 0xf) // row_shr:8
    BGE_DPP_STEP(0x142, 0xa) // row_bcast:15 -> rows 1, 3
    BGE_DPP_STEP(0x143, 0xc) // row_bcast:31 -> rows 2, 3
#undef BGE_DPP_STEP
    return v;
}

#ifdef BGE_EXPERIMENT_NO_SLEEP /* timing-only A/B build: bodies never fall asleep */
constexpr bool kSleepEnabled = false;
#else
constexpr bool kSleepEnabled = true;
#endif
#ifndef BGE_AABB_MIN_WAVES
#define BGE_AABB_MIN_WAVES 4 /* waves per SIMD the AABB variant is compiled for: at 8 it spills 20 VGPRs (measured ~2 % slower at 4 M bodies) */
#endif
#ifndef BGE_BASIS_MIN_WAVES
#define BGE_BASIS_MIN_WAVES 8 /* the BGE_TICK_BULLET_BASIS variants without AABBs / normal matrices: 55 VGPRs since parent * local is formed a row at a time (80 before, 6 waves per SIMD) */
#endif
#ifndef BGE_XFORM_MIN_WAVES
#define BGE_XFORM_MIN_WAVES 8 /* the variants with the transform part and nothing else heavy (the headline kernel).  While parent * local was formed on whole matrices (48 registers) they needed 72 VGPRs — at 64 they spilled 12 B per lane, 8 B per entity of scratch writes that rocprofv3 WRITE_SIZE showed — and ran at 7 waves per SIMD; with the product formed a row at a time (lds_mul_put) they need 55 / 48 */
#endif
template <bool PHYS, bool XFORM, bool AABB, bool NORMAL, bool BASIS>
// 8 waves per SIMD (<= 64 VGPRs): the kernel waits on memory and, in block tiles, on barriers; occupancy hides both
// (the NORMAL variant carries a 4x4 inverse: it gets 128 VGPRs instead of spilling)
__global__ void __launch_bounds__(kTile, NORMAL ? 4 : (AABB ? BGE_AABB_MIN_WAVES : (BASIS ? BGE_BASIS_MIN_WAVES : (XFORM ? BGE_XFORM_MIN_WAVES : 8)))) k_tick(WorldView w, TickParams p)
{
    __shared__ float4 lds[kTile * 4];
    // BASIS: Bullet's orientation step (getRotation -> exponential map -> safeNormalize -> setRotation: four square roots, ten
    // divisions) and the euler write-back behind it (asin, two atan2) change nothing for a body whose quaternion is a fixed
    // point of that round trip and does not spin — 88 % of the bodies in steady state; those carry kSettled and skip both.
    // The others are scattered over all waves, so no wave could skip the code: they queue (quaternion, angular velocity) here
    // and, after a barrier, the first n threads of the workgroup do the step for the n queued bodies (usually one wave
    // instead of four), writing the results over their queue entries, where the owners pick them up.
    // The queue lives in the matrix staging area, which the transform stage only touches after the barrier that ends its use.
    __shared__ uint32_t eq_count;
    float4* const eq_quat = lds;        // [kTile] in: quaternion; out: the new one
    float4* const eq_av = lds + kTile;  // [kTile] in: angular velocity | kEq* bits; out: euler angles | kEq* bits
    if (PHYS && BASIS) {
        if (threadIdx.x == 0) eq_count = 0;
        __syncthreads();
    }
    constexpr uint32_t kEqNone = 0xffffffffu;
    constexpr uint32_t kEqIntegrate = 1u;  // in: run the orientation step (else only the angles are wanted)
    constexpr uint32_t kEqForce = 2u;      // in: the pose was set or corrected in this tick: the angles are rewritten whatever the step does
    constexpr uint32_t kEqStoreQuat = 4u;  // out: the quaternion changed
    constexpr uint32_t kEqStoreEuler = 8u; // out: the angles are new
    constexpr uint32_t kEqSettled = 16u;   // out: the step left the quaternion of a non-spinning body as it was
    uint32_t eq_at = kEqNone; // this lane's queue entry

    const uint32_t tile = p.tile_begin + blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const uint32_t slot = tile * kTile + tid;
    const uint32_t hdr = w.tile_hdr[tile];
    const uint32_t count = (hdr >> kHdrCountShift) & kHdrCountMask;
    const uint32_t max_level = hdr & kHdrLevelMask;

    const uint32_t f0 = w.flags[slot];
    // The component loads do not wait for the flag word: every array is allocated for whole tiles, so the loads of a slot
    // that turns out to be empty (or of a velocity nobody integrates) are harmless, and issuing them together with the flags
    // takes one memory round trip instead of two out of a workgroup's life (BGE_SPECULATIVE_LOADS=0: A/B build).  Measured at
    // 1 M flat bodies: 24.4 -> 23.7 us per tick; velocities are only loaded early in tiles whose slots all carry a Dynamic body
    // (loading them everywhere cost 64-node subtrees, 1 body in 64 entities, 4 % more time)
#ifndef BGE_SPECULATIVE_LOADS
#define BGE_SPECULATIVE_LOADS 1
#endif
    F3 pos{0.0f, 0.0f, 0.0f}, eul{0.0f, 0.0f, 0.0f}, scl{1.0f, 1.0f, 1.0f}, vel_early{0.0f, 0.0f, 0.0f};
    if (BGE_SPECULATIVE_LOADS) {
        pos = ld3(w.pos, slot);
        eul = ld3(w.euler, slot);
        if (XFORM) scl = ld3(w.scale, slot);
#ifndef BGE_SPECULATIVE_VEL
#define BGE_SPECULATIVE_VEL 1
#endif
        if (BGE_SPECULATIVE_VEL && PHYS && (hdr & kHdrAllDynamic)) vel_early = ld3(w.vel, slot); // (elsewhere most slots have no velocity to read)
    }
    // The mass palette entry (gravity force, inverse mass) is one more dependent round trip behind the flag word.  In tiles whose
    // slots all carry a Dynamic body every lane fetches palette entry (lane) with the loads above and takes its class's entry
    // from lane (class) once the flags are there (classes are < 64; outside any divergent branch: a permute reads active lanes).
    // 1 M flat bodies: 23.4 -> 23.15 us per tick (three alternating runs).  Doing this — and the early velocity load — in every tile
    // with at least one Dynamic body in eight slots made depth-4 chains (bodies on the roots) 3 % SLOWER (22.5 -> 23.2 us): there
    // the extra loads cost more than the round trip they hide.  Not in the Bullet-basis variant either: 31.9 -> 32.1 us with it (and
    // with the queue counter's barrier moved behind the first loads), three alternating runs.
#ifndef BGE_PALETTE_SHUFFLE
#define BGE_PALETTE_SHUFFLE 1
#endif
    float4 gf_early = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    const bool palette_early = BGE_PALETTE_SHUFFLE && BGE_SPECULATIVE_LOADS && PHYS && !BASIS && (hdr & kHdrAllDynamic);
    if (palette_early) {
        const float4 mine = w.grav_palette[tid & 63u];
        const int cls0 = static_cast<int>((f0 >> kMassShift) & 63u);
        gf_early = make_float4(__shfl(mine.x, cls0, 64), __shfl(mine.y, cls0, 64), __shfl(mine.z, cls0, 64), __shfl(mine.w, cls0, 64));
    }
    // Ground plane on (p.cinfo_in): the contact word k_ground left and the deactivation record — most bodies of a scene at rest
    // are asleep — are two more dependent round trips behind the flag word; in all-dynamic tiles they are requested with the
    // first loads (8 B per body more, only in ticks that follow a k_ground launch).
    uint32_t ci_early = 0, dz_early = 0;
    const bool ground_early = BGE_SPECULATIVE_LOADS && PHYS && p.cinfo_in != nullptr && (hdr & kHdrAllDynamic);
    if (ground_early) {
        ci_early = p.cinfo_in[slot];
        dz_early = w.deact[slot];
    }
    uint32_t f = f0;
    const bool valid = (f & kValid) != 0;
    // A slot with a body but no Transform: the entity lost its Transform while its RigidBody stayed.  The reference keeps that
    // Bullet body in the world (EnsureRigidBody returns before it looks at the runtime, PhysicsSystem.cpp:389-393; nothing
    // removes it while the component exists): it is stepped, collides and enters trigger volumes like any other, but it is
    // never re-posed or re-created and nothing is written back.  Here: the physics part runs, the transform part does not.
    const bool orphan_body = PHYS && !valid && (f & kTypeMask) != 0;
    if (BGE_SPECULATIVE_LOADS) {
        if (!valid && !orphan_body) {
            pos = eul = F3{0.0f, 0.0f, 0.0f};
            scl = F3{1.0f, 1.0f, 1.0f};
        }
    } else if (valid || orphan_body) {
        pos = ld3(w.pos, slot);
        eul = ld3(w.euler, slot);
        if (XFORM) scl = ld3(w.scale, slot);
    }

    // (AABB variants) this lane's fed box for the wave's broadphase partial: identity when the slot carries no body
    float box_mn[3] = {INFINITY, INFINITY, INFINITY}, box_mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    bool has_box = false;
    if (PHYS) {
        const uint32_t type = f & kTypeMask;
        if (type != 0) { // (type bits exist only on slots that carry a body: with a Transform, or orphaned)
            const bool dynamic = type == 2u;
            // (p.no_repose: a later sub-step of the same stepSimulation call — the teleport rule ran before the first one)
            const bool repose = valid && p.no_repose == 0u && (f & (kTDirty | kBDirty)) != 0;
            bool spin = (f & kSpin) != 0;
            // BASIS (BGE_TICK_BULLET_BASIS): Bullet's own orientation scheme.  Its state is the 3x3 basis, which every step
            // goes basis -> getRotation -> exponential map -> safeNormalize -> setRotation for EVERY non-static body, spinning
            // or not.  The basis is always setRotation(q) of a stored quaternion, so q stays the state here and the basis
            // is recomputed from it: same bits, 16 instead of 36 bytes.
            bool advanced = false;
            // ground plane on: k_ground collided this body with the plane before this kernel ran and, if it is in contact (or
            // spinning), solved it — its velocities are final (gravity impulse included), its fed AABB is written
            uint32_t ci = 0;
            if (p.cinfo_in && dynamic) ci = ground_early ? ci_early : p.cinfo_in[slot];
            const bool solved = (ci & kCiSolved) != 0;
            // whatever writes the quaternion (re-pose, spin, the split impulse) takes kSettled away
            if (repose || spin || (ci & kCiMoved)) f &= ~kSettled;
            bool turn = BASIS ? (dynamic && !(f & kSettled)) : spin;
            Q4 q{0.0f, 0.0f, 0.0f, 1.0f};
            F3 v{0.0f, 0.0f, 0.0f};
            F3 av{0.0f, 0.0f, 0.0f};
            if (repose) {
                // SyncKinematicBodiesToPhysics / EnsureRigidBody: pose from the LOCAL Transform, zero velocities
                q = bt_quat_from_transform_euler(eul);
                st4(w.quat, slot, q);
                if (spin) { // (of whatever type the re-created body is: the record the download reads is the new body's)
                    st3(w.angvel, slot, av);
                    spin = false;
                    f &= ~kSpin;
                    if (!BASIS) turn = false; // its angular velocity is zero now: the default scheme leaves the re-posed quaternion alone
                }
            } else {
                if (dynamic) v = (BGE_SPECULATIVE_LOADS && BGE_SPECULATIVE_VEL && (hdr & kHdrAllDynamic)) ? vel_early : ld3(w.vel, slot);
                if (spin) av = ld3(w.angvel, slot);
                if (turn || AABB) q = ld4(w.quat, slot);
            }

            if (AABB && solved) {
                const float* bb = w.aabb + 6ull * slot; // written by k_ground from the pre-solve state
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    box_mn[a] = bb[a];
                    box_mx[a] = bb[3 + a];
                }
                has_box = true;
            }
            if (AABB && !solved) {
                const F3 he = ld3(w.half_extent, slot);
                const M3 r = bt_mat_from_quat(q);
                float mn[3], mx[3];
                bt_aabb_of_pose(pos, r, he, mn, mx);
                if (dynamic) {
                    // interpolation transform of predictUnconstraintMotion: velocity before the gravity impulse
                    const F3 pp{pos.x + v.x * p.dt, pos.y + v.y * p.dt, pos.z + v.z * p.dt};
                    float mn2[3], mx2[3];
                    if (turn) {
                        const M3 r2 = bt_mat_from_quat(bt_integrate_orientation(BASIS ? bt_quat_from_mat(r) : q, av, p.dt));
                        bt_aabb_of_pose(pp, r2, he, mn2, mx2);
                    } else {
                        bt_aabb_of_pose(pp, r, he, mn2, mx2);
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
                    box_mn[a] = mn[a];
                    box_mx[a] = mx[a];
                }
                has_box = true;
            }

            if (dynamic) {
                // mass class -> (gravity force, inv_mass) from the world's palette (a few distinct masses per scene,
                // L1/L2-resident); class 63 falls back to the per-slot array.  The force is btRigidBody::setGravity's
                // m_gravity = acceleration / m_inverseMass — a DIVISION per component in the reference's build
                // (oracle/tools/check_bullet_order.py) — which the host evaluates once per class and gravity vector.
                const uint32_t cls = f >> kMassShift;
                float inv_mass;
                F3 force;
                if (cls != kMassClassArray) {
                    const float4 gf = palette_early ? gf_early : w.grav_palette[cls];
                    inv_mass = gf.w;
                    force = F3{gf.x, gf.y, gf.z};
                } else {
                    inv_mass = w.inv_mass[slot];
                    force = F3{p.gx / inv_mass, p.gy / inv_mass, p.gz / inv_mass};
                }
                // Deactivation record: untouched (and unread) while the body is fast and its timer is zero.
                // (kDrowsy <=> record != 0; body (re)creation clears the bit: a new btRigidBody is ACTIVE_TAG, timer 0)
                const uint32_t dz0 = (kSleepEnabled && (f & kDrowsy)) ? (ground_early ? dz_early : w.deact[slot]) : 0u;
                uint32_t dz = dz0;
                // buildIslands: a free body is an island of its own; WANTS_DEACTIVATION -> ISLAND_SLEEPING
                // (kCiIsland: an island of several bodies with an active body in it keeps this one awake — bge_island.hip k_island_flags)
                if (dz == kDeactWants && !(ci & kCiIsland)) dz = kDeactSleeping;
                if (inv_mass != 0.0f && dz == kDeactSleeping) {
                    // asleep: no gravity, not solved, not integrated; updateActivationState zeroes the velocities
                    v = F3{0.0f, 0.0f, 0.0f};
                    st3(w.vel, slot, v);
                    if (spin) {
                        st3(w.angvel, slot, v);
                        spin = false;
                        f &= ~kSpin;
                    }
                } else if (inv_mass != 0.0f) {
                    // applyGravity (F = g * (1/invMass)) + solver write-back of the external force impulse
                    if (!solved) {
                        v.x = v.x + (force.x * inv_mass) * p.dt;
                        v.y = v.y + (force.y * inv_mass) * p.dt;
                        v.z = v.z + (force.z * inv_mass) * p.dt;
                    }
                    // integrateTransforms
                    pos.x = pos.x + v.x * p.dt;
                    pos.y = pos.y + v.y * p.dt;
                    pos.z = pos.z + v.z * p.dt;
                    if (turn && BASIS && !AABB) {
                        // queued: the first threads of the workgroup run the step after the barrier below
                        eq_at = atomicAdd(&eq_count, 1u);
                        eq_quat[eq_at] = make_float4(q.x, q.y, q.z, q.w);
                        eq_av[eq_at] = make_float4(av.x, av.y, av.z, __uint_as_float(kEqIntegrate | ((repose || (ci & kCiMoved)) ? kEqForce : 0u)));
                    } else if (turn) {
                        const Q4 qn = bt_integrate_orientation(BASIS ? bt_quat_from_mat(bt_mat_from_quat(q)) : q, av, p.dt);
                        // BASIS: a quaternion the round trip maps onto itself (88 % of the non-spinning bodies in steady state)
                        // needs neither the store nor new angles — rotationEuler already holds getEulerZYX of exactly these bits
                        const bool same = BASIS && !repose && __float_as_uint(qn.x) == __float_as_uint(q.x) && __float_as_uint(qn.y) == __float_as_uint(q.y) &&
                                          __float_as_uint(qn.z) == __float_as_uint(q.z) && __float_as_uint(qn.w) == __float_as_uint(q.w);
                        q = qn;
                        if (!same) {
                            st4(w.quat, slot, q);
                            advanced = true;
                        } else if (!spin && !(ci & kCiMoved)) {
                            f |= kSettled;
                        }
                    }
                    st3(w.vel, slot, v);
                    st3(w.pos, slot, pos);
                    // updateActivationState: updateDeactivation + wantsSleeping.  The kernel is close to VALU-bound, so the
                    // common case is decided by one compare: |v.y| >= threshold implies |v|^2 >= threshold^2 in float
                    // arithmetic too (rounding is monotone and the other two squares are >= 0), i.e. "not slow".
                    // (Measured at 1 M bodies: sleeping support costs 1.4 % of the tick this way, 2.0 % with the full test on
                    // every body, 2.9 % with the record handling moved behind a separate branch.)
                    if (kSleepEnabled && (!(fabsf(v.y) >= p.sleep_lin) || dz != 0u)) {
                        const float lin2 = v.x * v.x + v.y * v.y + v.z * v.z;
                        const float ang2 = av.x * av.x + av.y * av.y + av.z * av.z;
                        const bool slow = lin2 < p.sleep_lin2 && ang2 < p.sleep_ang2;
                        if (dz == kDeactWants) {
                            // kept awake by its island: wantsSleeping() is true for the state itself while the body is slow; faster
                            // than the thresholds it is ACTIVE_TAG again, timer 0
                            if (!slow) dz = 0u;
                        } else if (slow || dz != 0u) {
                            const float t = slow ? __uint_as_float(dz) + p.dt : 0.0f;
                            dz = (p.sleep_time != 0.0f && t > p.sleep_time) ? kDeactWants : __float_as_uint(t);
                        }
                    }
                }
                if (dz != dz0) {
                    w.deact[slot] = dz;
                    f = dz ? (f | kDrowsy) : (f & ~kDrowsy);
                }
                if (inv_mass == 0.0f && repose) {
                    st3(w.vel, slot, v);
                }
                // SyncRigidBodiesFromPhysics: rotationEuler <- getEulerZYX(basis) whenever the orientation was
                // (re)posed or advanced; a non-spinning body keeps its euler triple bit for bit
                if (eq_at == kEqNone && (repose || advanced || (ci & kCiMoved))) {
                    if (!(repose || advanced) && !(turn || AABB)) q = ld4(w.quat, slot); // the split impulse moved a body whose quaternion was not read
                    if (BASIS) {
                        eq_at = atomicAdd(&eq_count, 1u);
                        eq_quat[eq_at] = make_float4(q.x, q.y, q.z, q.w);
                        eq_av[eq_at] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(kEqForce));
                    } else {
                        eul = bt_transform_euler_from_mat(bt_mat_from_quat(q));
                        st3(w.euler, slot, eul);
                    }
                }
                if (ci & (kCiSolved | kCiMoved | kCiIsland)) w.cinfo[slot] = ci & ~(kCiSolved | kCiMoved | kCiIsland); // consumed
                if (valid) f |= kTDirty; // transform->MarkDirty()
            }
            if (valid) f &= ~kBDirty; // (an orphaned body is not re-created: its dirty bit waits for the Transform)
        }
    }

    if (PHYS && BASIS) {
        __syncthreads();
        const uint32_t n_queued = eq_count;
        // (rotating the wave that takes the first 64 entries with the workgroup index, so that not every resident workgroup's
        //  wave 0 does this work, changed nothing: 32.1 against 31.8 us)
        const uint32_t et = tid;
#ifndef BGE_EXPERIMENT_BASIS /* timing-only A/B builds (results are wrong): 1 = no orientation step for the queued bodies, 2 = no euler angles either */
#define BGE_EXPERIMENT_BASIS 0
#endif
        if (et < n_queued) {
            const float4 qq = eq_quat[et], aa = eq_av[et];
            const uint32_t in = __float_as_uint(aa.w);
            Q4 q{qq.x, qq.y, qq.z, qq.w};
            uint32_t out = (in & kEqForce) ? kEqStoreEuler : 0u;
            if ((in & kEqIntegrate) && BGE_EXPERIMENT_BASIS == 0) {
                const Q4 qn = bt_integrate_orientation(bt_quat_from_mat_sel(bt_mat_from_quat(q)), F3{aa.x, aa.y, aa.z}, p.dt);
                const bool same = !(in & kEqForce) && __float_as_uint(qn.x) == __float_as_uint(q.x) && __float_as_uint(qn.y) == __float_as_uint(q.y) &&
                                  __float_as_uint(qn.z) == __float_as_uint(q.z) && __float_as_uint(qn.w) == __float_as_uint(q.w);
                // (a re-posed or corrected body stores what the step made of its quaternion even when the bits did not move:
                //  what the inline form of this step does, and the next tick finds it settled)
                if (!same) out |= kEqStoreQuat | kEqStoreEuler;
                else if (aa.x == 0.0f && aa.y == 0.0f && aa.z == 0.0f) out |= kEqSettled;
                q = qn;
            }
            F3 e{0.0f, 0.0f, 0.0f};
            if ((out & kEqStoreEuler) && BGE_EXPERIMENT_BASIS < 2) e = bt_transform_euler_from_mat<true>(bt_mat_from_quat(q));
            eq_quat[et] = make_float4(q.x, q.y, q.z, q.w);
            eq_av[et] = make_float4(e.x, e.y, e.z, __uint_as_float(out));
        }
        __syncthreads();
        if (eq_at != kEqNone) {
            const float4 e = eq_av[eq_at];
            const uint32_t out = __float_as_uint(e.w);
            if (out & kEqStoreQuat) {
                const float4 qq = eq_quat[eq_at];
                st4(w.quat, slot, Q4{qq.x, qq.y, qq.z, qq.w});
            }
            if (out & kEqStoreEuler) {
                eul = F3{e.x, e.y, e.z};
                st3(w.euler, slot, eul);
            }
            if (out & kEqSettled) f |= kSettled;
        }
        if (XFORM) __syncthreads(); // the staging area is handed to the transform stage
    }

    if (AABB && PHYS) {
        // Broadphase partial of this wave (replaces a pass of k_bp_bounds over all AABBs): bounds, body count, widest box.
        // The extent is the expression the sort's large-body test evaluates (bge_broadphase.hip body_is_large).
        if (p.bp_partial) { // uniform
            float ext = fmaxf(fmaxf(box_mx[0] - box_mn[0], box_mx[1] - box_mn[1]), box_mx[2] - box_mn[2]);
            if (!(has_box && ext > 0.0f && ext < INFINITY)) ext = -INFINITY;
            const unsigned long long boxes = __ballot(has_box);
            float r[7];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                r[a] = wave_reduce_to_lane63<true>(box_mn[a]);
                r[3 + a] = wave_reduce_to_lane63<false>(box_mx[a]);
            }
            r[6] = wave_reduce_to_lane63<false>(ext);
            if ((tid & 63u) == 63u) {
                float4* dst = p.bp_partial + 2ull * (4ull * tile + (tid >> 6));
                dst[0] = make_float4(r[0], r[1], r[2], r[6]);
                dst[1] = make_float4(r[3], r[4], r[5], __uint_as_float(static_cast<uint32_t>(__popcll(boxes))));
            }
        }
    }

    if (XFORM) {
        float local[16];
        bx_mtx_srt(local, scl, eul, pos);
        const uint32_t level = (f & kLevelMask) >> kLevelShift;
        float4* dst = reinterpret_cast<float4*>(w.world) + 4ull * kTile * tile;
        bool keep_world = false;
        // A root that became one because its parent entity lost its Transform: the reference does not recompute it until it is
        // dirty (nothing marked it), so it keeps the stored parent * local matrix; the first dirty tick ends that.
        if ((hdr & kHdrFrozen) && valid && level == 0 && !(f & kExtParent)) { // (the header bit is uniform and almost never set)
            const uint32_t bit = 1u << (slot & 31u);
            if (w.frozen[slot >> 5] & bit) {
                if (f & kTDirty) atomicAnd(&w.frozen[slot >> 5], ~bit);
                else keep_world = true;
            }
        }

        if (hdr & kHdrWaveLocal) {
            // Every parent sits in its child's own 64-slot group (one wave64): the level loop and the write-out
            // need no workgroup barrier — DS operations of one wave execute in order.
            if (valid && level == 0) {
                if (f & kExtParent) {
                    world_mul_put(lds, w.world, w.parent[slot], tid, local);
                } else {
                    lds_put(lds, tid, local); // root: world = local (Transform.cpp:32-35)
                    if (keep_world) lds_put_stored(lds, tid, w.world, slot);
                    if (p.root_out && !(f & kHasParent)) {
                        if (keep_world) store_root_stored(p.root_out, w.root_index[slot], w.world, slot);
                        else store_root(p.root_out, w.root_index[slot], local);
                    }
                }
            }
            if (max_level != 0) {
                const uint32_t parent = (f & kParentMask) >> kParentShift;
                uint32_t wave_max = level;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) wave_max = max(wave_max, static_cast<uint32_t>(__shfl_xor(wave_max, off, 64)));
                for (uint32_t d = 1; d <= wave_max; ++d) {
                    wave_lds_sync();
                    if (valid && level == d) lds_mul_put(lds, parent, tid, local); // parent * local — the reference's order
                }
            }
            wave_lds_sync();
            // write-out of this wave's own 4 KiB: lane l stores float4 number l + 64k of the group's image
            const unsigned long long valid_mask = __ballot(valid);
            const uint32_t lane = tid & 63u;
            const uint32_t wbase = tid & ~63u;
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t qi = lane + 64u * k;      // float4 index inside the group's image
                const uint32_t nl = qi >> 2;             // node inside the group
                const uint32_t n = wbase + nl;
                const uint32_t r = (qi & 3u) ^ ((n >> 2) & 3u);
                if ((valid_mask >> nl) & 1ull) store_out(&dst[n * 4u + r], lds[wbase * 4u + qi], p.nt_out != 0);
            }
            if (NORMAL) {
                // render feed: normalMtx = transpose(inverse(world)) (Renderer.cpp:633-636), same LDS round trip
                float m[16], nm[16];
                if (valid) lds_get(lds, tid, m);
                wave_lds_sync();
                if (valid) {
                    bx_normal_matrix(nm, m);
                    lds_put(lds, tid, nm);
                }
                wave_lds_sync();
                float4* ndst = reinterpret_cast<float4*>(w.normal) + 4ull * kTile * tile;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const uint32_t qi = lane + 64u * k;
                    const uint32_t nl = qi >> 2;
                    const uint32_t n = wbase + nl;
                    const uint32_t r = (qi & 3u) ^ ((n >> 2) & 3u);
                    if ((valid_mask >> nl) & 1ull) store_out(&ndst[n * 4u + r], lds[wbase * 4u + qi], p.nt_out != 0);
                }
            }
        } else {
            // block tile (a subtree of 65..256 nodes, or the breadth-first prefix of a larger one): levels are
            // separated by workgroup barriers, parents staged in LDS
            const uint32_t parent = (f & kParentMask) >> kParentShift;
            if (valid && level == 0) {
                if (f & kExtParent) {
                    // parent resolved by an earlier launch: read its world matrix from memory
                    world_mul_put(lds, w.world, w.parent[slot], tid, local);
                } else {
                    lds_put(lds, tid, local);
                    if (keep_world) lds_put_stored(lds, tid, w.world, slot);
                    if (p.root_out && !(f & kHasParent)) {
                        if (keep_world) store_root_stored(p.root_out, w.root_index[slot], w.world, slot);
                        else store_root(p.root_out, w.root_index[slot], local);
                    }
                }
            }
            for (uint32_t d = 1; d <= max_level; ++d) {
#ifndef BGE_EXPERIMENT_NO_LEVEL_BARRIER /* timing-only A/B build: results are wrong without the barrier */
                __syncthreads();
#endif
                if (valid && level == d) lds_mul_put(lds, parent, tid, local); // parent * local — the reference's order
            }
            __syncthreads();
            // coalesced write-out: lane t stores float4 #(t + 256k) of the tile's 16 KiB image
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t qi = tid + kTile * k;
                const uint32_t n = qi >> 2;
                const uint32_t r = (qi & 3u) ^ ((n >> 2) & 3u);
                if (n < count) store_out(&dst[n * 4u + r], lds[qi], p.nt_out != 0);
            }
            if (NORMAL) {
                float m[16], nm[16];
                if (valid) lds_get(lds, tid, m);
                __syncthreads();
                if (valid) {
                    bx_normal_matrix(nm, m);
                    lds_put(lds, tid, nm);
                }
                __syncthreads();
                float4* ndst = reinterpret_cast<float4*>(w.normal) + 4ull * kTile * tile;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const uint32_t qi = tid + kTile * k;
                    const uint32_t n = qi >> 2;
                    const uint32_t r = (qi & 3u) ^ ((n >> 2) & 3u);
                    if (n < count) store_out(&ndst[n * 4u + r], lds[qi], p.nt_out != 0);
                }
            }
        }
        f &= ~kTDirty; // transform->dirty = false
    }

    if (f != f0) w.flags[slot] = f;
}

// PhysicsSystem::Update with a stepSimulation that runs no sub-step (Bullet's accumulator, src/physics/PhysicsSystem.cpp:855-863:
// m_localTime has not reached fixedStep yet).  What still happens around the step: EnsureRigidBody /
// SyncKinematicBodiesToPhysics re-pose every body whose Transform or RigidBody is dirty and zero a Dynamic one's
// velocities (:952-989); SyncRigidBodiesFromPhysics writes every Dynamic body's pose back and marks its Transform dirty
// (:916-950).  Position needs no copy (Transform::position and the body origin share storage); rotationEuler is rewritten
// when the body was re-posed, or always in Bullet's own orientation scheme (BASIS).
template <bool BASIS>
__global__ void __launch_bounds__(256) k_pose_only(WorldView w, uint64_t n_slots)
{
    const uint64_t slot64 = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (slot64 >= n_slots) return;
    const uint32_t slot = static_cast<uint32_t>(slot64);
    const uint32_t f0 = w.flags[slot];
    uint32_t f = f0;
    const uint32_t type = f & kTypeMask;
    if (!(f & kValid) || type == 0) return;
    const bool dynamic = type == 2u;
    const bool repose = (f & (kTDirty | kBDirty)) != 0;
    Q4 q{0.0f, 0.0f, 0.0f, 1.0f};
    if (repose) {
        q = bt_quat_from_transform_euler(ld3(w.euler, slot));
        st4(w.quat, slot, q);
        f &= ~kSettled;
        const F3 zero{0.0f, 0.0f, 0.0f};
        if (dynamic) st3(w.vel, slot, zero);
        if (f & kSpin) {
            st3(w.angvel, slot, zero);
            f &= ~kSpin;
        }
    }
    if (dynamic) {
        if (repose || BASIS) {
            if (!repose) q = ld4(w.quat, slot);
            st3(w.euler, slot, bt_transform_euler_from_mat(bt_mat_from_quat(q)));
        }
        f |= kTDirty; // transform->MarkDirty()
    }
    f &= ~kBDirty;
    if (f != f0) w.flags[slot] = f;
}

// ------------------------------------------------------------------ component scatter / gather (entity order <-> slots)
// stage holds `count` rows of `width` 32-bit words for entities [first, first+count).
// `index` (nullable) selects entities explicitly: row i belongs to entity index[i] instead of first + i.
__global__ void k_scatter_rows(const uint32_t* __restrict__ slot_of_entity, const uint32_t* __restrict__ index,
                               uint64_t first, uint64_t count,
                               uint32_t width, const uint32_t* __restrict__ stage, uint32_t* __restrict__ dst,
                               uint32_t* __restrict__ flags, uint32_t or_bits, uint32_t need_bits)
{
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i >= count) return;
    const uint32_t slot = slot_of_entity[index ? index[i] : first + i];
    if (slot == kNone) return;
    // need_bits = kValid: Transform data and dirty marks go to slots that hold a Transform — not to a body whose entity lost its
    // Transform (there the position array is the body's origin, and nothing marks what does not exist)
    if (need_bits && (flags[slot] & need_bits) != need_bits) return;
    if (dst) {
        for (uint32_t k = 0; k < width; ++k) dst[static_cast<uint64_t>(slot) * width + k] = stage[i * width + k];
    }
    if (flags && or_bits) flags[slot] |= or_bits;
}

__global__ void k_gather_rows(const uint32_t* __restrict__ slot_of_entity, const uint32_t* __restrict__ index,
                              uint64_t first, uint64_t count,
                              uint32_t width, const uint32_t* __restrict__ src, uint32_t* __restrict__ stage)
{
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i >= count) return;
    const uint32_t slot = slot_of_entity[index ? index[i] : first + i];
    for (uint32_t k = 0; k < width; ++k) {
        stage[i * width + k] = slot == kNone ? 0u : src[static_cast<uint64_t>(slot) * width + k];
    }
}

// Body upload: type/shape bits + dirty, inverse mass, AABB half extents, group/mask.
__global__ void k_scatter_bodies(const uint32_t* __restrict__ slot_of_entity, const uint32_t* __restrict__ index,
                                 uint64_t first, uint64_t count,
                                 const uint32_t* __restrict__ type_bits, const float* __restrict__ inv_mass,
                                 const float* __restrict__ half_extent3, const uint32_t* __restrict__ group,
                                 const uint32_t* __restrict__ mask, const uint32_t* __restrict__ filter_class, WorldView w,
                                 const float* __restrict__ cdims3, const float* __restrict__ cmass, const uint32_t* __restrict__ cbits)
{
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i >= count) return;
    const uint32_t slot = slot_of_entity[index ? index[i] : first + i];
    if (slot == kNone) return;
    // a body whose entity has no Transform is never re-created (EnsureRigidBody returns first): only its removal reaches it
    if (!(w.flags[slot] & kValid) && (type_bits[i] & kTypeMask) != 0) return;
    if (cdims3) {
        // a (re)created body starts without contacts: removeRigidBody dropped its pair with the ground and the manifold with it
        w.cshape[slot] = make_float4(cdims3[3 * i], cdims3[3 * i + 1], cdims3[3 * i + 2], 0.0f);
        w.cmass[slot] = cmass[i];
        w.cinfo[slot] = cbits[i];
    }
    w.filter_class[slot] = filter_class[i];
    uint32_t f = w.flags[slot];
    // a (re)created body is ACTIVE_TAG with its timer at zero.  kSpin stays: the body is re-created by the next physics tick
    // (EnsureRigidBody), whose re-pose rule zeroes the angular velocity record and takes the bit away
    f &= ~(kTypeMask | kBDirty | kMassMask | kDrowsy);
    f |= type_bits[i]; // body type, kBDirty and the mass class
    if ((f & kTypeMask) == 0) f &= ~kSpin;
    w.flags[slot] = f;
    w.inv_mass[slot] = inv_mass[i];
    w.half_extent[3ull * slot + 0] = half_extent3[3 * i + 0];
    w.half_extent[3ull * slot + 1] = half_extent3[3 * i + 1];
    w.half_extent[3ull * slot + 2] = half_extent3[3 * i + 2];
    w.group[slot] = group[i];
    w.mask[slot] = mask[i];
    if ((f & kTypeMask) == 0) {
        // body removed: forget its state
        w.vel[3ull * slot] = w.vel[3ull * slot + 1] = w.vel[3ull * slot + 2] = 0.0f;
        w.angvel[3ull * slot] = w.angvel[3ull * slot + 1] = w.angvel[3ull * slot + 2] = 0.0f;
    }
}

__global__ void k_scatter_velocities(const uint32_t* __restrict__ slot_of_entity, uint64_t first, uint64_t count,
                                     const float* __restrict__ lin, const float* __restrict__ ang, WorldView w)
{
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i >= count) return;
    const uint32_t slot = slot_of_entity[first + i];
    if (slot == kNone) return;
    uint32_t f = w.flags[slot];
    if ((f & kTypeMask) != 2u) return; // only Dynamic bodies carry velocity
    if (lin) {
        w.vel[3ull * slot + 0] = lin[3 * i + 0];
        w.vel[3ull * slot + 1] = lin[3 * i + 1];
        w.vel[3ull * slot + 2] = lin[3 * i + 2];
    }
    if (ang) {
        const float ax = ang[3 * i + 0], ay = ang[3 * i + 1], az = ang[3 * i + 2];
        w.angvel[3ull * slot + 0] = ax;
        w.angvel[3ull * slot + 1] = ay;
        w.angvel[3ull * slot + 2] = az;
        const uint32_t nf = (ax != 0.0f || ay != 0.0f || az != 0.0f) ? (f | kSpin) : (f & ~kSpin);
        if (nf != f) w.flags[slot] = nf;
    }
}

__global__ void k_init_slots(uint64_t n_slots, const uint32_t* __restrict__ structural_flags, WorldView w)
{
    const uint64_t s = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (s >= n_slots) return;
    const uint32_t sf = structural_flags[s];
    w.flags[s] = (sf & kValid) ? (sf | kTDirty) : sf; // Transform::Transform(): dirty = true
    w.pos[3 * s] = w.pos[3 * s + 1] = w.pos[3 * s + 2] = 0.0f;
    w.euler[3 * s] = w.euler[3 * s + 1] = w.euler[3 * s + 2] = 0.0f;
    w.scale[3 * s] = w.scale[3 * s + 1] = w.scale[3 * s + 2] = 1.0f;
    w.vel[3 * s] = w.vel[3 * s + 1] = w.vel[3 * s + 2] = 0.0f;
    w.angvel[3 * s] = w.angvel[3 * s + 1] = w.angvel[3 * s + 2] = 0.0f;
    w.quat[4 * s] = w.quat[4 * s + 1] = w.quat[4 * s + 2] = 0.0f;
    w.quat[4 * s + 3] = 1.0f;
    w.inv_mass[s] = 0.0f;
    w.deact[s] = 0u;
    w.half_extent[3 * s] = w.half_extent[3 * s + 1] = w.half_extent[3 * s + 2] = 0.5f;
    w.group[s] = 1u;
    w.mask[s] = 0xffffffffu;
    w.filter_class[s] = 0u;
    w.cshape[s] = make_float4(0.5f, 0.5f, 0.5f, 0.0f);
    w.cmass[s] = 0.0f;
    w.cfriction[s] = 0.5f; // RigidBody::friction default (src/ecs/PhysicsComponents.h:32)
    w.crestitution[s] = 0.0f; // RigidBody::restitution default (:33)
    w.cinfo[s] = kCiGroundMask;
    // mtxIdentity(local), mtxIdentity(world)
    for (int k = 0; k < 16; ++k) w.world[16 * s + k] = (k % 5 == 0) ? 1.0f : 0.0f;
    for (int k = 0; k < 6; ++k) w.aabb[6 * s + k] = 0.0f;
}

// Scene::CountDirtyTransforms: per-lane predicate -> wave ballot popcount -> one atomic per wave.
__global__ void k_count_dirty(uint64_t n_slots, const uint32_t* __restrict__ flags, unsigned long long* __restrict__ out)
{
    uint64_t s = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    unsigned long long local = 0;
    for (; s < n_slots; s += stride) {
        const uint32_t f = flags[s];
        local += ((f & kValid) && (f & kTDirty)) ? 1ull : 0ull;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63u) == 0 && local) atomicAdd(out, local);
}

__global__ void k_dirty_bytes(const uint32_t* __restrict__ slot_of_entity, uint64_t first, uint64_t count,
                              const uint32_t* __restrict__ flags, uint8_t* __restrict__ out)
{
    const uint64_t i = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    if (i >= count) return;
    const uint32_t slot = slot_of_entity[first + i];
    out[i] = (slot != kNone && (flags[slot] & kTDirty)) ? 1 : 0;
}

// Compact root rows (12 floats, see store_root): 3 lanes per root, 16 B each.
__global__ void k_pack_roots_compact(uint64_t n_roots, const uint32_t* __restrict__ root_slots, const float* __restrict__ world,
                                     float4* __restrict__ dst)
{
    const uint64_t t = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    const uint64_t r = t / 3;
    if (r >= n_roots) return;
    const uint32_t q = static_cast<uint32_t>(t - 3 * r);
    const float* m = world + 16ull * root_slots[r];
    float v[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint32_t cidx = 4u * q + k; // position in the compact row -> element of the 4x4 (skip column 3)
        v[k] = m[cidx + cidx / 3u];
    }
    dst[t] = make_float4(v[0], v[1], v[2], v[3]);
}

// Full 4x4 rows (the torch.distributed path of sharding.RootTable): 4 lanes per root, 16 B each.
__global__ void k_pack_roots(uint64_t n_roots, const uint32_t* __restrict__ root_slots, const float4* __restrict__ world,
                             float4* __restrict__ dst)
{
    const uint64_t t = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    const uint64_t r = t >> 2;
    if (r >= n_roots) return;
    dst[t] = world[4ull * root_slots[r] + (t & 3u)];
}

// ---- trigger volumes (ghost objects): AABB of the ghost at its entity's Transform, taken BEFORE the tick integrates
// (EnsureTrigger sets the ghost's pose at the start of PhysicsSystem::Update, src/physics/PhysicsSystem.cpp:575)
__global__ void k_trigger_aabb(uint32_t n_triggers, TriggerView t, WorldView w)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_triggers) return;
    const uint32_t slot = t.slot[i];
    float* bb = t.aabb + 6ull * i;
    if (slot == kFrozenGhost) return; // posed for the last time before its entity lost its Transform
    if (slot == kNone || !t.active[i]) {
        // an inactive ghost is not in the world: an empty box never overlaps
        bb[0] = bb[1] = bb[2] = INFINITY;
        bb[3] = bb[4] = bb[5] = -INFINITY;
        return;
    }
    const F3 pos = ld3(w.pos, slot);
    const F3 eul = ld3(w.euler, slot);
    const F3 he = ld3(t.half_extent, i);
    const M3 r = bt_mat_from_quat(bt_quat_from_transform_euler(eul));
    float mn[3], mx[3];
    bt_aabb_of_pose(pos, r, he, mn, mx);
    for (int a = 0; a < 3; ++a) {
        bb[a] = mn[a];
        bb[3 + a] = mx[a];
    }
}

// every body against every trigger (scenes carry a handful of triggers; n_bodies x n_triggers box tests); hits are
// ballot-compacted per wave and appended behind one atomic per wave and trigger.  With `list` only the list[0 .. *list_count)
// triggers are tested: the ones Broadphase::query_boxes left over (boxes that span too much of the grid to walk it).
__global__ void __launch_bounds__(256) k_trigger_pairs(uint64_t n_slots, uint32_t n_triggers, TriggerView t, WorldView w,
                                                       const uint32_t* __restrict__ entity_of_slot, uint32_t* __restrict__ count,
                                                       uint2* __restrict__ out, uint32_t cap, const uint32_t* __restrict__ list,
                                                       const uint32_t* __restrict__ list_count)
{
    const uint32_t n_tested = list ? *list_count : n_triggers;
    if (n_tested == 0) return;
    const uint64_t stride = static_cast<uint64_t>(gridDim.x) * blockDim.x;
    const uint64_t first = blockIdx.x * static_cast<uint64_t>(blockDim.x) + threadIdx.x;
    const uint64_t rounds = (n_slots + stride - 1) / stride; // uniform trip count: the ballot needs every lane
    const uint32_t lane = threadIdx.x & 63u;
    for (uint64_t r = 0; r < rounds; ++r) {
        const uint64_t s = first + r * stride;
        bool body = false;
        float bmn[3] = {0, 0, 0}, bmx[3] = {0, 0, 0};
        uint32_t grp = 0, msk = 0, ent = 0;
        if (s < n_slots) {
            const uint32_t f = w.flags[s];
            // every rigid body of whatever type: Bullet's pair cache pairs the ghost with Static bodies too (the reference hands
            // Bullet custom groups, PhysicsSystem.cpp:473,577; oracle/physics_ref.h has the reasoning)
            body = (f & kTypeMask) != 0u; // (with or without a Transform: an orphaned body is still in the world)
            if (body) {
                const float* b = w.aabb + 6 * s;
                for (int a = 0; a < 3; ++a) {
                    bmn[a] = b[a];
                    bmx[a] = b[3 + a];
                }
                grp = w.group[s];
                msk = w.mask[s];
                ent = entity_of_slot[s];
            }
        }
        for (uint32_t k = 0; k < n_tested; ++k) {
            const uint32_t i = list ? list[k] : k;
            const float* tb = t.aabb + 6ull * i;
            bool hit = body && ent != t.entity[i] && (t.group[i] & msk) != 0 && (grp & t.mask[i]) != 0;
            if (hit) {
                for (int a = 0; a < 3; ++a) hit = hit && tb[a] <= bmx[a] && tb[3 + a] >= bmn[a];
            }
            const unsigned long long m = __ballot(hit);
            if (m == 0) continue;
            const uint32_t leader = static_cast<uint32_t>(__ffsll(static_cast<long long>(m))) - 1u;
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(count, static_cast<uint32_t>(__popcll(m)));
            base = __shfl(base, static_cast<int>(leader), 64);
            if (hit) {
                const uint32_t at = base + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
                if (at < cap) out[at] = make_uint2(i, ent);
            }
        }
    }
}

// every ghost against every other ghost: both are registered collision objects, so the pair cache lists each in the other
// (btGhostPairCallback::addOverlappingPair serves both proxies).  Workgroup (x, y): the 256 ghosts i of tile x, one per thread,
// against the 64 ghosts j of tile y, which pass through LDS (one thread per i looping over ALL j took 352 us at 1000 ghosts: four
// workgroups, a thousand dependent trips each); a hit is (i | kGhostHit, j) — trigger INDICES, the host maps j to its entity and
// knows whether j is still in the world when i is processed (a one-shot ghost that fired earlier in ProcessTriggerEvents' loop is
// not).  An inactive ghost carries an empty box and overlaps nothing.
constexpr uint32_t kGhostTileJ = 64;
__global__ void __launch_bounds__(256) k_trigger_ghost_pairs(uint32_t n_triggers, TriggerView t, uint32_t* __restrict__ count,
                                                             uint2* __restrict__ out, uint32_t cap)
{
    __shared__ float s_box[kGhostTileJ][6];
    __shared__ uint32_t s_group[kGhostTileJ], s_mask[kGhostTileJ];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    float b[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    uint32_t grp = 0, msk = 0;
    if (i < n_triggers) {
#pragma unroll
        for (int a = 0; a < 6; ++a) b[a] = t.aabb[6ull * i + a];
        grp = t.group[i];
        msk = t.mask[i];
    }
    const uint32_t j0 = blockIdx.y * kGhostTileJ;
    const uint32_t jl = j0 + threadIdx.x;
    if (threadIdx.x < kGhostTileJ && jl < n_triggers) {
#pragma unroll
        for (int a = 0; a < 6; ++a) s_box[threadIdx.x][a] = t.aabb[6ull * jl + a];
        s_group[threadIdx.x] = t.group[jl];
        s_mask[threadIdx.x] = t.mask[jl];
    }
    __syncthreads();
    const uint32_t nj = min(kGhostTileJ, n_triggers - j0); // (uniform trip count: the ballots need every lane)
    for (uint32_t k = 0; k < nj; ++k) {
        const uint32_t j = j0 + k;
        bool hit = i < n_triggers && j != i && (grp & s_mask[k]) != 0u && (s_group[k] & msk) != 0u;
        if (hit) {
#pragma unroll
            for (int a = 0; a < 3; ++a) hit = hit && b[a] <= s_box[k][3 + a] && b[3 + a] >= s_box[k][a];
        }
        const unsigned long long m = __ballot(hit);
        if (m == 0) continue;
        const uint32_t leader = static_cast<uint32_t>(__ffsll(static_cast<long long>(m))) - 1u;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(count, static_cast<uint32_t>(__popcll(m)));
        base = __shfl(base, static_cast<int>(leader), 64);
        if (hit) {
            const uint32_t at = base + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
            if (at < cap) out[at] = make_uint2(i | kGhostHit, j);
        }
    }
}

inline dim3 grid_for(uint64_t n, uint32_t block) { return dim3(static_cast<uint32_t>((n + block - 1) / block)); }

} // namespace

// ------------------------------------------------------------------ launch wrappers
hipError_t launch_tick(hipStream_t stream, const WorldView& w, const TickParams& p, uint32_t n_tiles, uint32_t flags)
{
    if (n_tiles == 0) return hipSuccess;
    const bool phys = (flags & 1u) != 0, xform = (flags & 2u) != 0, aabb = (flags & (4u | 32u)) != 0, normal = (flags & 16u) != 0;
    const dim3 grid(n_tiles), block(kTile);
    const bool basis = phys && (flags & 64u) != 0;
#define BGE_LAUNCH(P, X, A, N)                                                                      \
    do {                                                                                            \
        if (basis) hipLaunchKernelGGL((k_tick<P, X, A, N, P>), grid, block, 0, stream, w, p);       \
        else hipLaunchKernelGGL((k_tick<P, X, A, N, false>), grid, block, 0, stream, w, p);         \
    } while (0)
    if (normal && xform) {
        if (phys && aabb) BGE_LAUNCH(true, true, true, true);
        else if (phys) BGE_LAUNCH(true, true, false, true);
        else BGE_LAUNCH(false, true, false, true);
    } else if (phys && xform && aabb) BGE_LAUNCH(true, true, true, false);
    else if (phys && xform) BGE_LAUNCH(true, true, false, false);
    else if (phys && aabb) BGE_LAUNCH(true, false, true, false);
    else if (phys) BGE_LAUNCH(true, false, false, false);
    else if (xform) BGE_LAUNCH(false, true, false, false);
#undef BGE_LAUNCH
    return hipGetLastError();
}

hipError_t launch_pose_only(hipStream_t stream, const WorldView& w, uint64_t n_slots, bool bullet_basis)
{
    if (n_slots == 0) return hipSuccess;
    if (bullet_basis) hipLaunchKernelGGL(k_pose_only<true>, grid_for(n_slots, 256), dim3(256), 0, stream, w, n_slots);
    else hipLaunchKernelGGL(k_pose_only<false>, grid_for(n_slots, 256), dim3(256), 0, stream, w, n_slots);
    return hipGetLastError();
}

hipError_t launch_scatter_rows(hipStream_t stream, const uint32_t* slot_of_entity, uint64_t first, uint64_t count,
                               uint32_t width, const void* stage, void* dst, uint32_t* flags, uint32_t or_bits,
                               const uint32_t* index, uint32_t need_bits)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_rows, grid_for(count, 256), dim3(256), 0, stream, slot_of_entity, index, first, count, width,
                       static_cast<const uint32_t*>(stage), static_cast<uint32_t*>(dst), flags, or_bits, need_bits);
    return hipGetLastError();
}

hipError_t launch_gather_rows(hipStream_t stream, const uint32_t* slot_of_entity, uint64_t first, uint64_t count,
                              uint32_t width, const void* src, void* stage, const uint32_t* index)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_rows, grid_for(count, 256), dim3(256), 0, stream, slot_of_entity, index, first, count, width,
                       static_cast<const uint32_t*>(src), static_cast<uint32_t*>(stage));
    return hipGetLastError();
}

hipError_t launch_scatter_bodies(hipStream_t stream, const uint32_t* slot_of_entity, uint64_t first, uint64_t count,
                                 const uint32_t* type_bits, const float* inv_mass, const float* half_extent3,
                                 const uint32_t* group, const uint32_t* mask, const uint32_t* filter_class, const WorldView& w,
                                 const uint32_t* index, const float* cdims3, const float* cmass, const uint32_t* cbits)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_bodies, grid_for(count, 256), dim3(256), 0, stream, slot_of_entity, index, first, count,
                       type_bits, inv_mass, half_extent3, group, mask, filter_class, w, cdims3, cmass, cbits);
    return hipGetLastError();
}

hipError_t launch_scatter_velocities(hipStream_t stream, const uint32_t* slot_of_entity, uint64_t first, uint64_t count,
                                     const float* lin, const float* ang, const WorldView& w)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_velocities, grid_for(count, 256), dim3(256), 0, stream, slot_of_entity, first, count,
                       lin, ang, w);
    return hipGetLastError();
}

hipError_t launch_init_slots(hipStream_t stream, uint64_t n_slots, const uint32_t* structural_flags, const WorldView& w)
{
    if (n_slots == 0) return hipSuccess;
    hipLaunchKernelGGL(k_init_slots, grid_for(n_slots, 256), dim3(256), 0, stream, n_slots, structural_flags, w);
    return hipGetLastError();
}

hipError_t launch_count_dirty(hipStream_t stream, uint64_t n_slots, const uint32_t* flags, unsigned long long* out)
{
    if (n_slots == 0) return hipSuccess;
    const uint64_t blocks = (n_slots + 255) / 256;
    hipLaunchKernelGGL(k_count_dirty, dim3(static_cast<uint32_t>(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream,
                       n_slots, flags, out);
    return hipGetLastError();
}

hipError_t launch_dirty_bytes(hipStream_t stream, const uint32_t* slot_of_entity, uint64_t first, uint64_t count,
                              const uint32_t* flags, uint8_t* out)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(k_dirty_bytes, grid_for(count, 256), dim3(256), 0, stream, slot_of_entity, first, count, flags, out);
    return hipGetLastError();
}

hipError_t launch_trigger_aabb(hipStream_t stream, uint32_t n_triggers, const TriggerView& t, const WorldView& w)
{
    if (n_triggers == 0) return hipSuccess;
    hipLaunchKernelGGL(k_trigger_aabb, grid_for(n_triggers, 64), dim3(64), 0, stream, n_triggers, t, w);
    return hipGetLastError();
}

hipError_t launch_trigger_pairs(hipStream_t stream, uint64_t n_slots, uint32_t n_triggers, const TriggerView& t, const WorldView& w,
                                const uint32_t* entity_of_slot, uint32_t* count, void* out_pairs, uint32_t cap, const uint32_t* list,
                                const uint32_t* list_count)
{
    if (n_triggers == 0 || n_slots == 0) return hipSuccess;
    const uint64_t blocks = (n_slots + 255) / 256;
    hipLaunchKernelGGL(k_trigger_pairs, dim3(static_cast<uint32_t>(blocks < 2048 ? blocks : 2048)), dim3(256), 0, stream, n_slots,
                       n_triggers, t, w, entity_of_slot, count, static_cast<uint2*>(out_pairs), cap, list, list_count);
    return hipGetLastError();
}

// ---- trigger overlap diff (bge_kernels.hpp TriggerDiff)
namespace {
constexpr uint32_t kTrigProbeLimit = 4096;

__device__ __forceinline__ uint32_t trig_hash(uint64_t key, uint32_t log2_cap)
{
    return static_cast<uint32_t>((key * 0x9E3779B97F4A7C15ull) >> (64u - log2_cap));
}

// 0 = was there already, 1 = inserted, 2 = no room within the probe limit
__device__ __forceinline__ int trig_insert(uint64_t* table, uint32_t log2_cap, uint64_t key)
{
    const uint32_t mask = (1u << log2_cap) - 1u;
    uint32_t h = trig_hash(key, log2_cap);
    for (uint32_t probe = 0; probe < kTrigProbeLimit; ++probe) {
        const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(table + h), static_cast<unsigned long long>(kTrigKeyEmpty),
                                                 static_cast<unsigned long long>(key));
        if (old == kTrigKeyEmpty) return 1;
        if (old == key) return 0;
        h = (h + 1u) & mask;
    }
    return 2;
}

__device__ __forceinline__ bool trig_contains(const uint64_t* table, uint32_t log2_cap, uint64_t key)
{
    const uint32_t mask = (1u << log2_cap) - 1u;
    uint32_t h = trig_hash(key, log2_cap);
    for (uint32_t probe = 0; probe <= mask; ++probe) {
        const uint64_t k = table[h];
        if (k == key) return true;
        if (k == kTrigKeyEmpty) return false;
        h = (h + 1u) & mask;
    }
    return false;
}

__device__ __forceinline__ void trig_append(const TriggerDiff& d, bool mine, uint64_t record)
{
    const unsigned long long m = __ballot(mine);
    if (m == 0ull) return;
    const int lane = static_cast<int>(threadIdx.x & 63u);
    uint32_t base = 0;
    if (lane == __ffsll(static_cast<long long>(m)) - 1) base = atomicAdd(d.header + 0, static_cast<uint32_t>(__popcll(m)));
    base = __shfl(base, __ffsll(static_cast<long long>(m)) - 1);
    if (mine) {
        const uint32_t at = base + static_cast<uint32_t>(__popcll(m & ((1ull << lane) - 1ull)));
        if (at < d.delta_cap) d.deltas[at] = record;
    }
}

// this tick's hits into the current table; the ones last tick's table does not hold are Enters
__global__ void __launch_bounds__(256) k_trigger_diff_cur(TriggerDiff d)
{
    const uint32_t hits = d.count[0];
    const uint32_t n = hits < d.pair_cap ? hits : d.pair_cap;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        d.header[3] = hits;
        d.header[4] = d.count[1];
        d.header[5] = d.count[2];
    }
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (n + stride - 1u) / stride; // every lane of a wave makes the same number of trips (ballots inside)
    for (uint32_t r = 0; r < rounds; ++r) {
        const uint32_t k = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool fresh = false, enter = false;
        uint64_t key = 0;
        if (k < n) {
            const uint2 hit = d.pairs[k];
            key = (static_cast<uint64_t>(hit.x & ~kGhostHit) << 33) | ((hit.x & kGhostHit) ? (1ull << 32) : 0ull) | hit.y;
            const int how = trig_insert(d.cur, d.log2_cap, key);
            if (how == 2) d.header[2] = 1u;
            fresh = how == 1;
            enter = fresh && !trig_contains(d.prev, d.log2_cap, key);
        }
        const unsigned long long f = __ballot(fresh);
        if ((threadIdx.x & 63u) == 0u && f) atomicAdd(d.header + 1, static_cast<uint32_t>(__popcll(f)));
        trig_append(d, enter, key);
    }
}

// last tick's keys that this tick's table does not hold are Exits
__global__ void __launch_bounds__(256) k_trigger_diff_prev(TriggerDiff d)
{
    const uint32_t cap = 1u << d.log2_cap;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t base = 0; base < cap; base += stride) {
        const uint32_t slot = base + blockIdx.x * blockDim.x + threadIdx.x;
        uint64_t key = kTrigKeyEmpty;
        if (slot < cap) key = d.prev[slot];
        const bool gone = key != kTrigKeyEmpty && !trig_contains(d.cur, d.log2_cap, key);
        trig_append(d, gone, key | kTrigKeyExit);
    }
}

__global__ void __launch_bounds__(256) k_trigger_table_build(uint64_t* table, uint32_t log2_cap, const uint64_t* keys, uint32_t n, uint32_t* header)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n && trig_insert(table, log2_cap, keys[k]) == 2) header[2] = 1u;
}
} // namespace

hipError_t launch_trigger_diff(hipStream_t stream, const TriggerDiff& d)
{
    hipLaunchKernelGGL(k_trigger_diff_cur, dim3(256), dim3(256), 0, stream, d);
    hipLaunchKernelGGL(k_trigger_diff_prev, dim3(256), dim3(256), 0, stream, d);
    return hipGetLastError();
}

hipError_t launch_trigger_table_build(hipStream_t stream, uint64_t* table, uint32_t log2_cap, const uint64_t* keys, uint32_t n, uint32_t* header)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_trigger_table_build, grid_for(n, 256), dim3(256), 0, stream, table, log2_cap, keys, n, header);
    return hipGetLastError();
}

hipError_t launch_trigger_ghost_pairs(hipStream_t stream, uint32_t n_triggers, const TriggerView& t, uint32_t* count, void* out_pairs,
                                      uint32_t cap)
{
    if (n_triggers < 2) return hipSuccess;
    hipLaunchKernelGGL(k_trigger_ghost_pairs, dim3((n_triggers + 255u) / 256u, (n_triggers + kGhostTileJ - 1u) / kGhostTileJ), dim3(256), 0, stream,
                       n_triggers, t, count, static_cast<uint2*>(out_pairs), cap);
    return hipGetLastError();
}

hipError_t launch_pack_roots(hipStream_t stream, uint64_t n_roots, const uint32_t* root_slots, const float* world, float* dst,
                             bool compact)
{
    if (n_roots == 0) return hipSuccess;
    if (compact) {
        hipLaunchKernelGGL(k_pack_roots_compact, grid_for(n_roots * 3, 256), dim3(256), 0, stream, n_roots, root_slots, world,
                           reinterpret_cast<float4*>(dst));
    } else {
        hipLaunchKernelGGL(k_pack_roots, grid_for(n_roots * 4, 256), dim3(256), 0, stream, n_roots, root_slots,
                           reinterpret_cast<const float4*>(world), reinterpret_cast<float4*>(dst));
    }
    return hipGetLastError();
}

} // namespace bge
