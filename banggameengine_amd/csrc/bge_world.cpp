// bge_world.cpp — implementation of the C ABI declared in include/bge_world.h.
//
// Host-side plumbing only: device allocation, the entity-index <-> slot maps, staging of uploads and
// downloads, and kernel launches on the world's stream.  All arithmetic of the hot path is in
// bge_kernels.hip; there is no CPU fallback — every entry point that needs the GPU fails with
// BGE_ERR_HIP when no device is usable.
#include "../../include/bge_world.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "bge_broadphase.hpp"
#include "bge_route.hpp"
#include "bge_comm.hpp"
#include "bge_flatten.hpp"
#include "bge_kernels.hpp"

namespace {

thread_local std::string g_last_error;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

// include/bge_world.h promises that no entry point throws: every `int bge_*` function is a function-try-block that ends
// in this handler list (host containers can throw std::bad_alloc; an exception must not unwind into a C caller).
#define BGE_CATCH_ALL(name)                                                                              \
    catch (const std::bad_alloc&) { return fail(BGE_ERR_OOM, "%s: host allocation failed", name); }      \
    catch (const std::exception& e) { return fail(BGE_ERR_STATE, "%s: unexpected exception: %s", name, e.what()); } \
    catch (...) { return fail(BGE_ERR_STATE, "%s: unexpected exception", name); }

#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        const hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                            \
            return fail(e_ == hipErrorOutOfMemory ? BGE_ERR_OOM : BGE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                                        \
        }                                                                                                  \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need)
    {
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        const hipError_t e = hipMalloc(&p, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// A DevBuf that frees itself: temporaries of one call (an early HIP_TRY return must not leak them).
struct TmpBuf : DevBuf {
    TmpBuf() = default;
    TmpBuf(const TmpBuf&) = delete;
    TmpBuf& operator=(const TmpBuf&) = delete;
    TmpBuf(TmpBuf&& o) noexcept
    {
        p = o.p;
        bytes = o.bytes;
        o.p = nullptr;
        o.bytes = 0;
    }
    ~TmpBuf() { release(); }
};

// Collider size -> AABB half extents in the collider's frame.
// Box: btBoxShape ctor (implicit = he - 0.04), setSafeMargin (margin = min(0.04, 0.1*min he) through
// btBoxShape::setMargin), btTransformAabb adds the margin back.  Capsule (up axis Y): (r, r+h/2, r).
// Reference clamps: src/physics/PhysicsSystem.cpp:692-703.
void collider_half_extents(uint8_t shape, const float* size, float* out)
{
    if (shape == BGE_SHAPE_CAPSULE) {
        const float radius = std::max(size[0], 0.01f);
        const float half_height = std::max(size[1], 0.0f);
        const float h = 0.5f * (half_height * 2.0f);
        out[0] = radius;
        out[1] = radius + h;
        out[2] = radius;
        return;
    }
    const float hx = std::max(size[0], 0.01f), hy = std::max(size[1], 0.01f), hz = std::max(size[2], 0.01f);
    const float m0 = 0.04f;
    float ix = hx - m0, iy = hy - m0, iz = hz - m0;
    const float min_dim = std::min(hx, std::min(hy, hz));
    const float safe = 0.1f * min_dim;
    float margin = m0;
    if (safe < margin) {
        ix = (ix + m0) - safe;
        iy = (iy + m0) - safe;
        iz = (iz + m0) - safe;
        margin = safe;
    }
    out[0] = ix + margin;
    out[1] = iy + margin;
    out[2] = iz + margin;
}

} // namespace

struct bge_world {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    uint64_t pair_capacity_req = 0;

    bge::Flattened flat;
    std::vector<uint32_t> parent_entity; // entity-level parents (Scene::m_parents) as of the last set_topology
    DevBuf frozen;                       // WorldView::frozen (allocated with the first frozen root)
    bool any_frozen = false;
    std::vector<uint8_t> body_type_host; // bge_body_type per entity index as last uploaded (BGE_BODY_NONE = no body)
    std::vector<uint8_t> orphan_host;    // 1: the entity lost its Transform while it had a body; the body kept its slot (kValid clear)
    // A body exists in the reference's world from the first PhysicsSystem::Update that sees its components AND a Transform
    // (EnsureRigidBody); an entity that loses its Transform before that has no body to keep.  body_since[e]: the value of
    // physics_updates when the body of entity e was (last) uploaded onto a slot with a Transform; it exists once a later update ran.
    std::vector<uint64_t> body_since;
    uint64_t physics_updates = 0;
    bool has_topology = false;
    bool maybe_dirty = true;
    float local_time = 0.0f; // btDiscreteDynamicsWorld::m_localTime (bge_world_step_simulation)

    // device arrays
    DevBuf flags, parent, tile_hdr, slot_of_entity, entity_of_slot, root_slots, root_index;
    DevBuf pos, euler, scale, world, vel, angvel, quat, inv_mass, half_extent, group, mask, aabb;
    DevBuf root_worlds, counter, stage, stage2, mass_palette, normal, deact, filter_class, filter_table, grav_palette;
    DevBuf cshape, cmass, cfriction, cinfo, manifold; // ground contact (bge_contact.hip); manifold allocated when the plane is switched on
    DevBuf ground_list, ground_count;                 // slots k_ground_select hands to the solver; count + ticket words
    // Dynamic boxes on the Static / Kinematic box colliders of the scene (round 3, bge_contact.hip): off by default, like the plane
    bool static_contacts = false;
    DevBuf crestitution;                              // RigidBody::restitution per slot (allocated with the layout, zero = the component default)
    DevBuf bmanifold;                                 // kBoxManifolds manifold rows per slot (allocated when the feature is switched on)
    DevBuf obstacle_slots, obstacle_gen, obstacles, obstacle_grid, box_list, box_count;
    bool obstacle_grid_on = true; // BGE_OBSTACLE_GRID=0: every body tests every obstacle (measurements)
    uint32_t n_obstacles = 0;
    bool obstacles_stale = true;                      // the compact list of Static / Kinematic box bodies must be rebuilt
    std::vector<uint8_t> body_shape_host;             // bge_shape per entity index as last uploaded
    std::vector<uint32_t> body_gen_host;              // how often the entity's body was (re)created: a re-created box is a new pair
    std::vector<std::vector<uint32_t>> trig_scratch;  // process_trigger_pairs: this tick's body overlaps per trigger (capacity kept)
    std::vector<std::vector<uint32_t>> trig_scratch_ghosts; // ... and the ghosts met, as trigger indices
    std::vector<uint32_t> trig_union;                 // ... and the union being diffed
    std::vector<uint32_t> trig_pairs_host;            // ... and the downloaded (trigger, entity) hit list
    bool ground_plane = false; // the reference's static plane y = 0 (PhysicsSystem.cpp:149-166); off: free bodies (BASELINE's workloads)
    DevBuf bp_partials; // per-wave bounds / count / widest extent written by the tick kernel for the broadphase (32 B per wave)
    float grav_cached[3] = {0.0f, 0.0f, 0.0f};
    bool grav_palette_stale = true; // the mass palette or the gravity vector changed since the table was built
    // Collision-filter palette: scenes use a handful of (layer, mask, static) combinations, so the broadphase's sorted
    // records carry an 8-bit class instead of two 32-bit words (32-byte records instead of 48).  Class 0 = (1, ~0, not
    // static), the component defaults; with more than 255 combinations the broadphase falls back to full records.
    struct FilterKey {
        uint32_t group, mask, is_static;
        bool operator==(const FilterKey& o) const { return group == o.group && mask == o.mask && is_static == o.is_static; }
    };
    std::vector<FilterKey> filter_palette{FilterKey{1u, 0xffffffffu, 0u}};
    bool filter_overflow = false;
    bool filter_table_stale = true;
    uint32_t filter_class_of(uint32_t group, uint32_t mask, bool is_static)
    {
        const FilterKey k{group, mask, is_static ? 1u : 0u};
        for (size_t c = 0; c < filter_palette.size(); ++c) {
            if (filter_palette[c] == k) return static_cast<uint32_t>(c);
        }
        if (filter_palette.size() >= 255) {
            filter_overflow = true;
            return 0;
        }
        filter_palette.push_back(k);
        filter_table_stale = true;
        return static_cast<uint32_t>(filter_palette.size() - 1);
    }
    // Bullet's defaults (btRigidBodyConstructionInfo: 0.8 / 1.0; gDeactivationTime 2 s) — the reference never changes them
    float sleep_lin = 0.8f, sleep_ang = 1.0f, sleep_time = 2.0f;
    void fill_sleep(bge::TickParams& p) const
    {
        p.sleep_lin = sleep_lin;
        p.sleep_lin2 = sleep_lin * sleep_lin;
        p.sleep_ang2 = sleep_ang * sleep_ang;
        p.sleep_time = sleep_time;
    }
    std::vector<float> palette_inv_mass;            // class -> inv_mass (class 0 = 0: Static / Kinematic)
    std::unordered_map<uint32_t, uint32_t> palette_class; // inv_mass bits -> class
    bge::Broadphase broadphase;
    // sharded broadphase (bge_route.hip): records routed to spatial slabs, pair search over what was received
    bge::ShardRouter router;
    bge::Broadphase slab_broadphase;
    // Dynamic boxes against each other (bge_island.hip): a broadphase of its own over the sub-step's fed AABBs (the tick's
    // pair list and the trigger query keep theirs), the sorted pair cache with its manifolds in two generations, per-slot scratch
    bool dynamic_contacts = false;
    bool static_contacts_ever = false; // bge_world_set_static_contacts(1) was called: bodies may hold manifolds with obstacles
    bge::Broadphase island_bp;
    DevBuf isl_slot_words, isl_counts, isl_identity, isl_gen, isl_keys_raw, isl_keys[2], isl_man[2], isl_body_keys_raw, isl_body_slot_raw,
        isl_body_keys, isl_body_slot, isl_solver_bodies, isl_rows, isl_sort_tmp, isl_big_list, isl_body_words, isl_ints;
    uint32_t* isl_counts_host = nullptr; // pinned
    int isl_cur = 0;
    uint32_t isl_n_prev = 0;
    uint64_t isl_identity_n = 0;
    bool isl_gen_stale = true;
    uint32_t isl_last_pairs = 0, isl_last_bodies = 0;
    uint32_t isl_big_points = 128; // islands with more contact points go to the workgroup solver (BGE_ISLAND_BIG_POINTS: tests force it)
    bool bp_shared = false; // this sub-step's island phases ran `broadphase` on the boxes the tick would run it on: the tick skips its run
    uint64_t isl_pair_cap = 0; // pair capacity of island_bp (grows by itself unless bge_world_create fixed pair_capacity)
    bool pairs_from_slab = false;          // bge_world_pairs reads the slab search (global ids) instead of the local one
    std::vector<uint32_t> global_id_host;  // per entity index; empty = identity
    DevBuf global_of_slot, bp_send, bp_recv, bp_small, bp_hist;
    bool global_of_slot_stale = true;
    bge::RootComm comm;
    bge::WorldView view{};

    // trigger volumes: host runtime (what PhysicsSystem keeps in m_triggerRuntime) + device mirrors
    struct Trigger {
        uint32_t entity = 0;
        uint8_t shape = 0;
        float size[3] = {0.5f, 0.5f, 0.5f};
        uint32_t layer = 4, mask = 0xffffffffu;
        bool one_shot = false;
        bool component_active = true; // TriggerVolume::active
        bool runtime_active = false;  // ghost is in the world
        bool posed = false;           // ... and a tick has posed it on the device since
        // The entity lost its Transform: EnsureTrigger returns before it touches the ghost (PhysicsSystem.cpp:530-534) and nothing
        // removes it, so the ghost stays in the world where it last was and keeps reporting overlaps.
        bool frozen = false;
        float frozen_aabb[6] = {0, 0, 0, 0, 0, 0};
        std::vector<uint32_t> overlaps;       // sorted entity indices of the previous tick: the union of the two lists below
        std::vector<uint32_t> overlap_bodies; // ... met as rigid bodies (of any type: the pair cache pairs a ghost with Static bodies too)
        std::vector<uint32_t> overlap_ghosts; // ... met as other trigger ghosts (each of two overlapping ghosts lists the other)
        void clear_overlaps()
        {
            overlaps.clear();
            overlap_bodies.clear();
            overlap_ghosts.clear();
        }
    };
    bool owns_transform(uint32_t e) const
    {
        return e < flat.slot_of_entity.size() && flat.slot_of_entity[e] != bge::kNone && !(e < orphan_host.size() && orphan_host[e]);
    }
    std::vector<Trigger> triggers; // in upload order = the order ProcessTriggerEvents' loop walks them (bge_world.h)
    std::unordered_map<uint32_t, uint32_t> trig_index_of_entity;
    std::vector<bge_trigger_event> trigger_events; // since the last bge_world_trigger_events
    bool triggers_device_stale = true;
    bool trig_list_on_device = false; // the device arrays are indexed like `triggers` (false between an upload of the list and the next sync)
    DevBuf trig_slot, trig_entity, trig_he, trig_group, trig_mask, trig_active, trig_aabb, trig_pairs, trig_count, trig_lists;
    uint32_t trigger_grid_min = 64;       // more ghosts than this: the broadphase grid answers for the small ones
    // Enter / Exit taken on the device (bge_kernels.hpp TriggerDiff): two key tables (this tick's, last tick's), header + deltas in
    // one device buffer with a page-locked copy.  The overlap sets above stay the truth; `trig_mirror_valid` says that last tick's
    // table holds exactly them — it is rebuilt from them after every tick that went the long way or changed them on the host.
    DevBuf trig_tab[2], trig_delta_dev, trig_keys_dev;
    uint32_t trig_tab_log2 = 0;
    void* trig_delta_host = nullptr;      // hipHostMalloc
    bool trig_mirror_valid = false;
    bool trig_device_diff = true;         // BGE_TRIGGER_DEVICE_DIFF=0 keeps every tick on the host's path (A/B, tests)
    bool trig_stay_events = true;         // bge_world_set_trigger_stay_events
    uint64_t trig_stay_suppressed = 0;    // Stay events not materialised since the last bge_world_trigger_events
    uint64_t trig_fast_ticks = 0, trig_slow_ticks = 0;
    double trig_wait_ms = 0.0, trig_apply_ms = 0.0; // BGE_TRIGGER_PROFILE=1: time until the deltas are on the host / spent applying them
    uint64_t trig_deltas_seen = 0;
    std::vector<uint64_t> trig_keys_host;
    std::vector<uint32_t> trig_exits, trig_touched;
    std::vector<uint8_t> trig_was;
    uint64_t trig_total_overlaps = 0;     // sum of the sets' sizes over the volumes in the world (kept by the short way, recounted by the long one)
    uint32_t trig_through_grid = 0, trig_against_all = 0; // how the last tick split them
    bge::TriggerView trigger_view() const
    {
        bge::TriggerView t{};
        t.slot = trig_slot.as<uint32_t>();
        t.entity = trig_entity.as<uint32_t>();
        t.half_extent = trig_he.as<float>();
        t.group = trig_group.as<uint32_t>();
        t.mask = trig_mask.as<uint32_t>();
        t.active = trig_active.as<uint8_t>();
        t.aabb = trig_aabb.as<float>();
        return t;
    }

    // opt-in hipGraph replay of back-to-back ticks (BGE_USE_GRAPH=1; measured slower than eager launches, see tick_many)
    static constexpr uint32_t kGraphTicks = 32;
    static constexpr uint32_t kGraphMaxTiles = 512; // ~131 k entities: above that a tick outlasts a host launch anyway
    hipGraphExec_t graph_exec = nullptr;
    uint32_t graph_flags = 0;
    float graph_dt = 0.0f, graph_g[3] = {0, 0, 0};
    bool graph_disabled = false;
    void drop_graph()
    {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        graph_exec = nullptr;
    }

    // optional event-pair timing of the tick kernels
    int profiling = 0;                   // 0 off, 1 one pair per tick_many call, 2 one pair per tick
    std::vector<hipEvent_t> prof_events; // start/stop pairs
    size_t prof_used = 0;                // events recorded since the last read
    std::vector<uint32_t> prof_ticks_pending; // ticks covered by each recorded pair
    double prof_ms_carry = 0.0;          // time of pairs folded in when the ring wrapped
    uint64_t prof_ticks_carry = 0;

    void rebuild_view()
    {
        drop_graph(); // captured launches hold the old pointers
        view.flags = flags.as<uint32_t>();
        view.parent = parent.as<uint32_t>();
        view.tile_hdr = tile_hdr.as<uint32_t>();
        view.pos = pos.as<float>();
        view.euler = euler.as<float>();
        view.scale = scale.as<float>();
        view.world = world.as<float>();
        view.vel = vel.as<float>();
        view.angvel = angvel.as<float>();
        view.quat = quat.as<float>();
        view.inv_mass = inv_mass.as<float>();
        view.half_extent = half_extent.as<float>();
        view.group = group.as<uint32_t>();
        view.mask = mask.as<uint32_t>();
        view.aabb = aabb.as<float>();
        view.mass_palette = mass_palette.as<float2>();
        view.grav_palette = grav_palette.as<float4>();
        view.deact = deact.as<uint32_t>();
        view.filter_class = filter_class.as<uint32_t>();
        view.root_index = root_index.as<uint32_t>();
        view.normal = normal.as<float>();
        view.cshape = cshape.as<float4>();
        view.cmass = cmass.as<float>();
        view.cfriction = cfriction.as<float>();
        view.cinfo = cinfo.as<uint32_t>();
        view.manifold = manifold.as<float>();
        view.crestitution = crestitution.as<float>();
        view.bmanifold = bmanifold.as<uint32_t>();
        view.frozen = frozen.as<uint32_t>();
    }
    void release_all()
    {
        for (DevBuf* b : {&flags, &parent, &tile_hdr, &slot_of_entity, &entity_of_slot, &root_slots, &root_index, &pos, &euler, &scale, &world, &vel,
                          &angvel, &quat, &inv_mass, &half_extent, &group, &mask, &aabb, &root_worlds, &counter, &stage,
                          &stage2, &mass_palette, &normal, &deact, &filter_class, &filter_table, &grav_palette, &bp_partials, &cshape, &cmass, &cfriction, &cinfo, &manifold, &crestitution, &bmanifold, &obstacle_slots, &obstacle_gen, &obstacles, &obstacle_grid, &box_list, &box_count, &frozen, &trig_slot, &trig_entity, &trig_he, &trig_group,
                          &trig_mask, &trig_active, &trig_aabb, &trig_pairs, &trig_count, &trig_lists, &ground_list, &ground_count, &trig_tab[0], &trig_tab[1],
                          &trig_delta_dev, &trig_keys_dev}) {
            b->release();
        }
        if (trig_delta_host) (void)hipHostFree(trig_delta_host);
        trig_delta_host = nullptr;
        broadphase.release();
        slab_broadphase.release();
        island_bp.release();
        for (DevBuf* b : {&isl_slot_words, &isl_counts, &isl_identity, &isl_gen, &isl_keys_raw, &isl_keys[0], &isl_keys[1], &isl_man[0], &isl_man[1], &isl_body_keys_raw,
                          &isl_body_slot_raw, &isl_body_keys, &isl_body_slot, &isl_solver_bodies, &isl_rows, &isl_sort_tmp, &isl_big_list, &isl_body_words, &isl_ints})
            b->release();
        if (isl_counts_host) (void)hipHostFree(isl_counts_host);
        isl_counts_host = nullptr;
        router.release();
        for (DevBuf* b : {&global_of_slot, &bp_send, &bp_recv, &bp_small, &bp_hist}) b->release();
        comm.destroy();
        drop_graph();
        for (hipEvent_t e : prof_events) (void)hipEventDestroy(e);
        prof_events.clear();
    }
};

namespace {

// Mass class of an inverse mass: scenes use a handful of distinct masses, so the kernel reads (inv_mass, mass)
// from a 64-entry palette instead of 4 B per body; the 63rd distinct value onwards uses the per-slot array.
uint32_t mass_class(bge_world* w, float inv_mass, bool& palette_changed)
{
    uint32_t bits;
    std::memcpy(&bits, &inv_mass, 4);
    auto it = w->palette_class.find(bits);
    if (it != w->palette_class.end()) return it->second;
    if (w->palette_inv_mass.size() >= bge::kMassClassArray) return bge::kMassClassArray;
    const uint32_t cls = static_cast<uint32_t>(w->palette_inv_mass.size());
    w->palette_inv_mass.push_back(inv_mass);
    w->palette_class[bits] = cls;
    palette_changed = true;
    return cls;
}

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int check_range(const bge_world* w, uint64_t first, uint64_t count)
{
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
    if (first > w->flat.n_entities || count > w->flat.n_entities - first) {
        return fail(BGE_ERR_INVALID, "entity range [%llu, +%llu) outside [0, %llu)", (unsigned long long)first,
                    (unsigned long long)count, (unsigned long long)w->flat.n_entities);
    }
    return BGE_OK;
}

// upload one host array of `count` rows x `width` words into a per-slot device array
// device copy of an explicit entity-index list (validated against n_entities), or null for a range call
int stage_index(bge_world* w, uint64_t count, const uint32_t* index, const uint32_t** dev)
{
    *dev = nullptr;
    if (!index) return BGE_OK;
    for (uint64_t i = 0; i < count; ++i) {
        if (index[i] >= w->flat.n_entities) {
            return fail(BGE_ERR_INVALID, "entity_index[%llu] = %u outside [0, %llu)", (unsigned long long)i, index[i],
                        (unsigned long long)w->flat.n_entities);
        }
    }
    HIP_TRY(w->stage2.ensure(count * 4));
    HIP_TRY(hipMemcpyAsync(w->stage2.p, index, count * 4, hipMemcpyHostToDevice, w->stream));
    *dev = w->stage2.as<uint32_t>();
    return BGE_OK;
}

int upload_rows(bge_world* w, uint64_t first, uint64_t count, uint32_t width, const void* host, void* dst,
                uint32_t or_bits, const uint32_t* dev_index = nullptr, uint32_t need_bits = 0)
{
    const size_t bytes = static_cast<size_t>(count) * width * 4;
    HIP_TRY(w->stage.ensure(bytes));
    HIP_TRY(hipMemcpyAsync(w->stage.p, host, bytes, hipMemcpyHostToDevice, w->stream));
    HIP_TRY(bge::launch_scatter_rows(w->stream, w->slot_of_entity.as<uint32_t>(), first, count, width, w->stage.p, dst,
                                     w->flags.as<uint32_t>(), or_bits, dev_index, need_bits));
    // the staging buffer is reused by the next call
    HIP_TRY(hipStreamSynchronize(w->stream));
    return BGE_OK;
}

int download_rows(bge_world* w, uint64_t first, uint64_t count, uint32_t width, const void* src, void* host,
                  const uint32_t* dev_index = nullptr)
{
    const size_t bytes = static_cast<size_t>(count) * width * 4;
    if (w->flat.identity && !dev_index) {
        // flat scene: slot == entity index, the rows are already contiguous in entity order
        HIP_TRY(hipMemcpyAsync(host, static_cast<const char*>(src) + first * width * 4, bytes, hipMemcpyDeviceToHost, w->stream));
        HIP_TRY(hipStreamSynchronize(w->stream));
        return BGE_OK;
    }
    HIP_TRY(w->stage.ensure(bytes));
    HIP_TRY(bge::launch_gather_rows(w->stream, w->slot_of_entity.as<uint32_t>(), first, count, width, src, w->stage.p, dev_index));
    HIP_TRY(hipMemcpyAsync(host, w->stage.p, bytes, hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    return BGE_OK;
}

} // namespace

namespace {
constexpr uint32_t kTriggerPairCap = 1u << 20;
constexpr double kNtThresholdBytes = 300.0e6; // measured crossover between 2 M (plain wins) and 4 M slots (nt wins)

// device mirrors of the trigger set (slots follow the current topology)
int sync_triggers_to_device(bge_world* w)
{
    const size_t n = w->triggers.size();
    if (n == 0) return BGE_OK;
    std::vector<uint32_t> slot(n), entity(n), group(n), mask(n);
    std::vector<float> he(3 * n);
    std::vector<uint8_t> active(n);
    for (size_t i = 0; i < n; ++i) {
        const bge_world::Trigger& t = w->triggers[i];
        entity[i] = t.entity;
        slot[i] = w->owns_transform(t.entity) ? w->flat.slot_of_entity[t.entity] : bge::kNone;
        collider_half_extents(t.shape, t.size, &he[3 * i]);
        group[i] = t.layer;
        mask[i] = t.mask;
        active[i] = t.runtime_active && slot[i] != bge::kNone ? 1 : 0;
        if (t.runtime_active && t.frozen) {
            slot[i] = bge::kFrozenGhost; // k_trigger_aabb leaves its box alone
            active[i] = 1;
        }
    }
    HIP_TRY(w->trig_slot.ensure(n * 4));
    HIP_TRY(w->trig_entity.ensure(n * 4));
    HIP_TRY(w->trig_he.ensure(n * 12));
    HIP_TRY(w->trig_group.ensure(n * 4));
    HIP_TRY(w->trig_mask.ensure(n * 4));
    HIP_TRY(w->trig_active.ensure(n));
    HIP_TRY(w->trig_aabb.ensure(n * 24));
    HIP_TRY(w->trig_pairs.ensure(static_cast<size_t>(kTriggerPairCap) * 8));
    HIP_TRY(w->trig_count.ensure(64));
    HIP_TRY(hipMemcpyAsync(w->trig_slot.p, slot.data(), n * 4, hipMemcpyHostToDevice, w->stream));
    HIP_TRY(hipMemcpyAsync(w->trig_entity.p, entity.data(), n * 4, hipMemcpyHostToDevice, w->stream));
    HIP_TRY(hipMemcpyAsync(w->trig_he.p, he.data(), n * 12, hipMemcpyHostToDevice, w->stream));
    HIP_TRY(hipMemcpyAsync(w->trig_group.p, group.data(), n * 4, hipMemcpyHostToDevice, w->stream));
    HIP_TRY(hipMemcpyAsync(w->trig_mask.p, mask.data(), n * 4, hipMemcpyHostToDevice, w->stream));
    HIP_TRY(hipMemcpyAsync(w->trig_active.p, active.data(), n, hipMemcpyHostToDevice, w->stream));
    for (size_t i = 0; i < n; ++i) {
        const bge_world::Trigger& t = w->triggers[i];
        if (t.runtime_active && t.frozen) {
            HIP_TRY(hipMemcpyAsync(static_cast<char*>(w->trig_aabb.p) + 24 * i, t.frozen_aabb, 24, hipMemcpyHostToDevice, w->stream));
        }
    }
    HIP_TRY(hipStreamSynchronize(w->stream)); // the host vectors die here
    w->triggers_device_stale = false;
    w->trig_list_on_device = true;
    return BGE_OK;
}

// EnsureTrigger's activation rule, before the step (PhysicsSystem.cpp:573-590)
void ensure_triggers(bge_world* w)
{
    for (bge_world::Trigger& t : w->triggers) {
        const bool has_tf = w->owns_transform(t.entity);
        if (!has_tf && t.runtime_active && t.frozen) continue; // EnsureTrigger returns before it touches the ghost
        if (has_tf && t.frozen) {                              // the Transform is back: posed from it again
            t.frozen = false;
            w->triggers_device_stale = true;
        }
        const bool want = t.component_active && has_tf;
        if (want != t.runtime_active) {
            t.runtime_active = want;
            t.posed = false;
            t.clear_overlaps();
            w->triggers_device_stale = true;
            w->trig_mirror_valid = false;
        }
    }
}

// One trip of ProcessTriggerEvents' loop (PhysicsSystem.cpp:1019-1073) for trigger t: `bodies` = entities met as rigid bodies
// (sorted), `ghosts` = other triggers met, as indices into w->triggers (any order).  A ghost that has left the world since the
// lists were made — a one-shot trigger that fired EARLIER in this loop (:1062-1072 removes it at once, and the pair cache takes
// it out of every other ghost's list) — no longer counts.  An entity met both ways counts once (:1026 std::unordered_set).
void diff_and_commit_trigger(bge_world* w, bge_world::Trigger& t, std::vector<uint32_t>& bodies, std::vector<uint32_t>& ghosts)
{
    size_t kept = 0;
    for (uint32_t j : ghosts) {
        if (j < w->triggers.size() && w->triggers[j].runtime_active) ghosts[kept++] = w->triggers[j].entity;
    }
    ghosts.resize(kept);
    std::sort(ghosts.begin(), ghosts.end());
    std::vector<uint32_t>& cur = w->trig_union;
    cur.resize(bodies.size() + ghosts.size());
    cur.erase(std::set_union(bodies.begin(), bodies.end(), ghosts.begin(), ghosts.end(), cur.begin()), cur.end());
    // (both sets are sorted, so Enter / Stay / Exit come out of two linear merges — with a binary search per overlap the host
    //  side of 92,000 overlaps a tick took 2.7 ms)
    const std::vector<uint32_t>& prev = t.overlaps;
    size_t at = 0;
    for (uint32_t other : cur) {
        while (at < prev.size() && prev[at] < other) ++at;
        const bool was = at < prev.size() && prev[at] == other;
        if (was && !w->trig_stay_events) {
            ++w->trig_stay_suppressed;
            continue;
        }
        w->trigger_events.push_back(bge_trigger_event{was ? 1u : 0u, t.entity, other});
    }
    at = 0;
    for (uint32_t previous : prev) {
        while (at < cur.size() && cur[at] < previous) ++at;
        if (!(at < cur.size() && cur[at] == previous)) w->trigger_events.push_back(bge_trigger_event{2u, t.entity, previous});
    }
    t.overlaps.swap(cur);
    t.overlap_bodies.swap(bodies);
    t.overlap_ghosts.swap(ghosts);
    if (t.one_shot && !t.overlaps.empty()) {
        t.component_active = false;
        t.runtime_active = false; // out of the world, and out of every list, from here on
        t.clear_overlaps();
        w->triggers_device_stale = true;
    }
}

// ProcessTriggerEvents on the hit list the device produced for this tick: (trigger, body entity) from the all-bodies pass and
// the grid look-up, (trigger | kGhostHit, other trigger) from the ghost-against-ghost pass
int process_trigger_pairs(bge_world* w)
{
    uint32_t counters[3] = {0, 0, 0};
    HIP_TRY(hipMemcpyAsync(counters, w->trig_count.p, 12, hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    const uint32_t n_pairs = counters[0];
    w->trig_against_all = counters[1];
    w->trig_through_grid = counters[2];
    if (n_pairs > kTriggerPairCap) return fail(BGE_ERR_OOM, "%u trigger overlaps in one tick exceed the buffer of %u", n_pairs, kTriggerPairCap);
    std::vector<uint32_t>& pairs = w->trig_pairs_host; // (kept between ticks: no 0.7 MB value-initialisation per tick)
    if (pairs.size() < 2 * static_cast<size_t>(n_pairs)) pairs.resize(2 * static_cast<size_t>(n_pairs));
    if (n_pairs) {
        HIP_TRY(hipMemcpyAsync(pairs.data(), w->trig_pairs.p, 2 * static_cast<size_t>(n_pairs) * 4, hipMemcpyDeviceToHost, w->stream));
        HIP_TRY(hipStreamSynchronize(w->stream));
    }
    // (the per-trigger lists keep their capacity from tick to tick)
    std::vector<std::vector<uint32_t>>& bodies = w->trig_scratch;
    std::vector<std::vector<uint32_t>>& ghosts = w->trig_scratch_ghosts;
    const size_t n_trig = w->triggers.size();
    if (bodies.size() < n_trig) bodies.resize(n_trig);
    if (ghosts.size() < n_trig) ghosts.resize(n_trig);
    for (size_t i = 0; i < n_trig; ++i) {
        bodies[i].clear();
        ghosts[i].clear();
    }
    for (uint32_t k = 0; k < n_pairs; ++k) {
        const uint32_t a = pairs[2 * k], b = pairs[2 * k + 1];
        const uint32_t i = a & ~bge::kGhostHit;
        if (i >= n_trig) continue;
        if (a & bge::kGhostHit) ghosts[i].push_back(b);
        else bodies[i].push_back(b);
    }
    w->trigger_events.reserve(w->trigger_events.size() + n_pairs + n_pairs / 4);
    for (size_t i = 0; i < n_trig; ++i) {
        bge_world::Trigger& t = w->triggers[i];
        if (!t.runtime_active) continue;
        std::sort(bodies[i].begin(), bodies[i].end());
        diff_and_commit_trigger(w, t, bodies[i], ghosts[i]);
    }
    return BGE_OK;
}

// ---- the short way: Enter / Exit come from the device (bge_kernels.hpp TriggerDiff), the sets are updated by them
constexpr uint32_t kTrigDeltaCap = 1u << 16;       // deltas fetched with the header in one copy; more than that in a tick: the long way
constexpr uint32_t kTrigHeaderWords = 8;

inline void sorted_insert(std::vector<uint32_t>& v, uint32_t e)
{
    auto it = std::lower_bound(v.begin(), v.end(), e);
    if (it == v.end() || *it != e) v.insert(it, e);
}
inline void sorted_erase(std::vector<uint32_t>& v, uint32_t e)
{
    auto it = std::lower_bound(v.begin(), v.end(), e);
    if (it != v.end() && *it == e) v.erase(it);
}
inline bool sorted_has(const std::vector<uint32_t>& v, uint32_t e) { return std::binary_search(v.begin(), v.end(), e); }

// (Re)build last tick's table from the host's sets: after a tick on the long way, an upload of the trigger list, a (de)activation
int rebuild_trigger_mirror(bge_world* w)
{
    std::vector<uint64_t>& keys = w->trig_keys_host;
    keys.clear();
    w->trig_total_overlaps = 0;
    for (size_t i = 0; i < w->triggers.size(); ++i) {
        const bge_world::Trigger& t = w->triggers[i];
        if (!t.runtime_active) continue;
        w->trig_total_overlaps += t.overlaps.size();
        for (uint32_t e : t.overlap_bodies) keys.push_back((static_cast<uint64_t>(i) << 33) | e);
        for (uint32_t e : t.overlap_ghosts) {
            auto it = w->trig_index_of_entity.find(e);
            if (it != w->trig_index_of_entity.end()) keys.push_back((static_cast<uint64_t>(i) << 33) | (1ull << 32) | it->second);
        }
    }
    uint32_t log2 = 12;
    while ((1ull << log2) < 4 * std::max<uint64_t>(keys.size(), w->trig_pairs_host.size() / 2)) ++log2;
    if (log2 > 26) return fail(BGE_ERR_OOM, "%llu remembered trigger overlaps", (unsigned long long)keys.size());
    if (log2 > w->trig_tab_log2) w->trig_tab_log2 = log2;
    const size_t bytes = sizeof(uint64_t) << w->trig_tab_log2;
    for (DevBuf& b : w->trig_tab) HIP_TRY(b.ensure(bytes));
    HIP_TRY(w->trig_delta_dev.ensure(kTrigHeaderWords * 4 + static_cast<size_t>(kTrigDeltaCap) * 8));
    if (!w->trig_delta_host) HIP_TRY(hipHostMalloc(&w->trig_delta_host, kTrigHeaderWords * 4 + static_cast<size_t>(kTrigDeltaCap) * 8, hipHostMallocDefault));
    HIP_TRY(hipMemsetAsync(w->trig_tab[0].p, 0xff, bytes, w->stream)); // this tick's table for the next diff
    HIP_TRY(hipMemsetAsync(w->trig_tab[1].p, 0xff, bytes, w->stream)); // last tick's
    HIP_TRY(hipMemsetAsync(w->trig_delta_dev.p, 0, kTrigHeaderWords * 4, w->stream));
    if (!keys.empty()) {
        HIP_TRY(w->trig_keys_dev.ensure(keys.size() * 8));
        HIP_TRY(hipMemcpyAsync(w->trig_keys_dev.p, keys.data(), keys.size() * 8, hipMemcpyHostToDevice, w->stream));
        HIP_TRY(bge::launch_trigger_table_build(w->stream, w->trig_tab[1].as<uint64_t>(), w->trig_tab_log2, w->trig_keys_dev.as<uint64_t>(),
                                                static_cast<uint32_t>(keys.size()), w->trig_delta_dev.as<uint32_t>()));
        uint32_t overflow = 0;
        HIP_TRY(hipMemcpyAsync(&overflow, w->trig_delta_dev.as<uint32_t>() + 2, 4, hipMemcpyDeviceToHost, w->stream));
        HIP_TRY(hipStreamSynchronize(w->stream)); // (`keys` is read by the copy above)
        if (overflow) return fail(BGE_ERR_HIP, "trigger mirror table overflowed at %llu keys", (unsigned long long)keys.size());
    }
    w->trig_mirror_valid = true;
    return BGE_OK;
}

// One tick's trigger events from the device's deltas.  Returns 1 when the tick has to go the long way after all (too many changes,
// a table overflow), 0 when done, a negative error otherwise.
int process_trigger_pairs_fast(bge_world* w)
{
    const auto t0 = std::chrono::steady_clock::now();
    bge::TriggerDiff d{};
    d.cur = w->trig_tab[0].as<uint64_t>();
    d.prev = w->trig_tab[1].as<uint64_t>();
    d.log2_cap = w->trig_tab_log2;
    d.header = w->trig_delta_dev.as<uint32_t>();
    d.deltas = reinterpret_cast<uint64_t*>(w->trig_delta_dev.as<uint32_t>() + kTrigHeaderWords);
    d.delta_cap = kTrigDeltaCap;
    d.pairs = static_cast<const uint2*>(w->trig_pairs.p);
    d.count = w->trig_count.as<uint32_t>();
    d.pair_cap = kTriggerPairCap;
    HIP_TRY(hipMemsetAsync(d.header, 0, kTrigHeaderWords * 4, w->stream));
    HIP_TRY(bge::launch_trigger_diff(w->stream, d));
    // the header and the first 1,024 deltas in one copy; a tick with more changes fetches the rest in a second one
    constexpr uint32_t kFirst = 1024;
    uint32_t* host = static_cast<uint32_t*>(w->trig_delta_host);
    HIP_TRY(hipMemcpyAsync(host, w->trig_delta_dev.p, kTrigHeaderWords * 4 + kFirst * 8, hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    const uint32_t n_delta = host[0], n_pairs = host[3];
    w->trig_against_all = host[4];
    w->trig_through_grid = host[5];
    if (n_pairs > kTriggerPairCap) return fail(BGE_ERR_OOM, "%u trigger overlaps in one tick exceed the buffer of %u", n_pairs, kTriggerPairCap);
    if (host[2] || n_delta > kTrigDeltaCap || (static_cast<uint64_t>(host[1]) << 1) > (1ull << w->trig_tab_log2)) {
        w->trig_mirror_valid = false; // (both tables are rebuilt, larger, after the long way)
        if (w->trig_pairs_host.size() < 2 * static_cast<size_t>(n_pairs)) w->trig_pairs_host.resize(2 * static_cast<size_t>(n_pairs));
        return 1;
    }
    if (n_delta > kFirst) {
        HIP_TRY(hipMemcpyAsync(host + kTrigHeaderWords + 2 * kFirst, reinterpret_cast<const uint64_t*>(w->trig_delta_dev.as<uint32_t>() + kTrigHeaderWords) + kFirst,
                               static_cast<size_t>(n_delta - kFirst) * 8, hipMemcpyDeviceToHost, w->stream));
        HIP_TRY(hipStreamSynchronize(w->stream));
    }
    // this tick's table becomes last tick's; the other one is cleared for the next diff (in stream order, behind the kernels)
    std::swap(w->trig_tab[0], w->trig_tab[1]);
    HIP_TRY(hipMemsetAsync(w->trig_tab[0].p, 0xff, sizeof(uint64_t) << w->trig_tab_log2, w->stream));
    const auto t1 = std::chrono::steady_clock::now();
    uint64_t* deltas = reinterpret_cast<uint64_t*>(host + kTrigHeaderWords);
    std::sort(deltas, deltas + n_delta, [](uint64_t a, uint64_t b) { return (a & ~bge::kTrigKeyExit) < (b & ~bge::kTrigKeyExit); });
    const size_t n_trig = w->triggers.size();
    std::vector<uint32_t>& enters = w->trig_union; // (scratch: the entities that entered the trigger at hand, ascending)
    std::vector<uint32_t>& exits = w->trig_exits;
    std::vector<uint32_t>& touched = w->trig_touched;
    std::vector<uint8_t>& was = w->trig_was;
    uint32_t at = 0;
    uint64_t all_enters = 0;
    // with Stay records every volume in the world is reported; without them only the volumes the deltas name are visited
    for (size_t i = 0; i < n_trig; ++i) {
        if (!w->trig_stay_events) {
            if (at >= n_delta) break;
            i = static_cast<size_t>((deltas[at] & ~bge::kTrigKeyExit) >> 33);
            if (i >= n_trig) break;
        }
        bge_world::Trigger& t = w->triggers[i];
        enters.clear();
        exits.clear();
        touched.clear();
        const uint32_t first = at;
        while (at < n_delta && ((deltas[at] & ~bge::kTrigKeyExit) >> 33) == i) ++at;
        if (!t.runtime_active) continue; // (the device skips a ghost that is not in the world: nothing can be listed for it)
        if (at > first) {
            // the entities whose membership may change, and whether they are members now
            for (uint32_t k = first; k < at; ++k) {
                const uint64_t key = deltas[k] & ~bge::kTrigKeyExit;
                const uint32_t other = static_cast<uint32_t>(key);
                const bool ghost = (key >> 32) & 1ull;
                if (ghost && other >= n_trig) continue;
                touched.push_back(ghost ? w->triggers[other].entity : other);
            }
            std::sort(touched.begin(), touched.end());
            touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
            was.resize(touched.size());
            for (size_t k = 0; k < touched.size(); ++k) was[k] = sorted_has(t.overlaps, touched[k]);
            for (uint32_t k = first; k < at; ++k) {
                const bool exit = (deltas[k] & bge::kTrigKeyExit) != 0;
                const uint64_t key = deltas[k] & ~bge::kTrigKeyExit;
                const uint32_t other = static_cast<uint32_t>(key);
                const bool ghost = (key >> 32) & 1ull;
                if (ghost && other >= n_trig) continue;
                std::vector<uint32_t>& set = ghost ? t.overlap_ghosts : t.overlap_bodies;
                const uint32_t e = ghost ? w->triggers[other].entity : other;
                if (exit) sorted_erase(set, e);
                else sorted_insert(set, e);
            }
            for (size_t k = 0; k < touched.size(); ++k) {
                const uint32_t e = touched[k];
                const bool is = sorted_has(t.overlap_bodies, e) || sorted_has(t.overlap_ghosts, e); // (met both ways counts once)
                if (is && !was[k]) {
                    sorted_insert(t.overlaps, e);
                    enters.push_back(e);
                } else if (!is && was[k]) {
                    sorted_erase(t.overlaps, e);
                    exits.push_back(e);
                }
            }
            w->trig_total_overlaps += enters.size();
            w->trig_total_overlaps -= exits.size();
            all_enters += enters.size();
        }
        // ProcessTriggerEvents' report for this trigger: Enter / Stay over the current set in ascending entity order, then Exit
        if (w->trig_stay_events) {
            size_t en = 0;
            for (uint32_t e : t.overlaps) {
                const bool entered = en < enters.size() && enters[en] == e;
                if (entered) ++en;
                w->trigger_events.push_back(bge_trigger_event{entered ? 0u : 1u, t.entity, e});
            }
        } else {
            for (uint32_t e : enters) w->trigger_events.push_back(bge_trigger_event{0u, t.entity, e});
        }
        for (uint32_t e : exits) w->trigger_events.push_back(bge_trigger_event{2u, t.entity, e});
    }
    if (!w->trig_stay_events) w->trig_stay_suppressed += w->trig_total_overlaps - all_enters;
    ++w->trig_fast_ticks;
    w->trig_wait_ms += std::chrono::duration<double, std::milli>(t1 - t0).count();
    w->trig_apply_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    w->trig_deltas_seen += n_delta;
    return 0;
}

// ... and in a call that simulates nothing: no collision detection ran, every ghost's list is last call's — minus the ghosts
// that are no longer in the world (deactivated by EnsureTrigger since, or fired as one-shot earlier in this loop); a volume
// that was made one-shot since fires on what it remembers (PhysicsSystem.cpp:1062-1072)
void process_triggers_without_a_step(bge_world* w)
{
    w->trig_mirror_valid = false; // (a ghost that left the world drops out of the others' lists here, on the host)
    std::vector<uint32_t> bodies, ghosts;
    for (bge_world::Trigger& t : w->triggers) {
        if (!t.runtime_active) continue;
        bodies = t.overlap_bodies;
        ghosts.clear();
        for (uint32_t e : t.overlap_ghosts) {
            auto it = w->trig_index_of_entity.find(e);
            if (it != w->trig_index_of_entity.end()) ghosts.push_back(it->second);
        }
        diff_and_commit_trigger(w, t, bodies, ghosts);
    }
}

// The compact list of what a Dynamic box can rest on — every Static / Kinematic body with a box collider, ascending entity index,
// with the generation of its btRigidBody — and the buffers of the box-contact path (bge_contact.hip); rebuilt after body uploads
// and re-topologies only.
int prepare_obstacles(bge_world* w, uint64_t n_slots)
{
    if (w->box_list.bytes < n_slots * 4 || !w->box_count.p) {
        HIP_TRY(w->box_list.ensure(std::max<uint64_t>(n_slots, 1) * 4));
        HIP_TRY(w->box_count.ensure(64));
        HIP_TRY(hipMemsetAsync(w->box_count.p, 0, 64, w->stream));
    }
    if (!w->obstacles_stale) return BGE_OK;
    std::vector<uint32_t> slots, gens;
    for (uint64_t e = 0; e < w->flat.n_entities && e < w->body_type_host.size(); ++e) {
        const uint8_t t = w->body_type_host[e];
        if (t != BGE_BODY_STATIC && t != BGE_BODY_KINEMATIC) continue;
        if (e < w->body_shape_host.size() && w->body_shape_host[e] == BGE_SHAPE_CAPSULE) continue;
        const uint32_t sl = w->flat.slot_of_entity[e];
        if (sl == bge::kNone) continue;
        slots.push_back(sl);
        gens.push_back(e < w->body_gen_host.size() ? w->body_gen_host[e] : 0u);
    }
    w->n_obstacles = static_cast<uint32_t>(slots.size());
    const size_t n = std::max<size_t>(slots.size(), 1);
    HIP_TRY(hipStreamSynchronize(w->stream)); // (a kernel of the previous tick may still read the old list)
    HIP_TRY(w->obstacle_slots.ensure(n * 4));
    HIP_TRY(w->obstacle_gen.ensure(n * 4));
    HIP_TRY(w->obstacles.ensure(n * sizeof(bge::ObstacleRec)));
    // (an obstacle on the grid covers at most 64 cells: 64 items each is a capacity that cannot overflow)
    if (w->n_obstacles > bge::kObstacleGridMin) HIP_TRY(w->obstacle_grid.ensure((bge::kObstacleGridItems + 64ull * n) * 4));
    if (!slots.empty()) {
        HIP_TRY(hipMemcpy(w->obstacle_slots.p, slots.data(), slots.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(w->obstacle_gen.p, gens.data(), gens.size() * 4, hipMemcpyHostToDevice));
    }
    w->obstacles_stale = false;
    return BGE_OK;
}

// the world's collision-filter palette as the broadphase takes it (uploads the table when it changed)
int filter_palette_of(bge_world* w, bge::FilterPalette* out)
{
    if (w->filter_table_stale && !w->filter_overflow) {
        std::vector<uint32_t> tab(256 * 4, 0u);
        for (size_t c = 0; c < w->filter_palette.size(); ++c) {
            tab[4 * c] = w->filter_palette[c].group;
            tab[4 * c + 1] = w->filter_palette[c].mask;
            tab[4 * c + 2] = w->filter_palette[c].is_static;
        }
        HIP_TRY(hipMemcpyAsync(w->filter_table.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, w->stream));
        HIP_TRY(hipStreamSynchronize(w->stream)); // `tab` is a local
        w->filter_table_stale = false;
    }
    *out = bge::FilterPalette{w->filter_overflow ? nullptr : w->filter_class.as<uint32_t>(), w->filter_overflow ? nullptr : w->filter_table.as<uint4>(),
                              static_cast<uint32_t>(w->filter_palette.size())};
    return BGE_OK;
}

// One sub-step's collision detection and constraint solving for the Dynamic boxes that touch each other (bge_island.hip;
// oracle/physics_ref.h CollideDynamicPairs / StepIsland).  Runs before k_ground_select; two small read-backs (pairs, island bodies)
// size the sorts between its phases.
int island_substep(bge_world* w, bge::GroundParams& gp, uint64_t n_slots, bool bullet_basis, bool later_sub_step, bool tick_wants_pairs)
{
    const uint64_t n_entities = std::max<uint64_t>(w->flat.n_entities, 1);
    HIP_TRY(w->isl_slot_words.ensure(std::max<uint64_t>(n_slots, 1) * 16));
    if (!w->isl_counts.p) {
        HIP_TRY(w->isl_counts.ensure(64));
        HIP_TRY(hipMemsetAsync(w->isl_counts.p, 0, 64, w->stream));
    }
    if (!w->isl_counts_host) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&w->isl_counts_host), 64));
    if (w->isl_identity_n < n_slots) {
        std::vector<uint32_t> iota(n_slots);
        for (uint64_t i = 0; i < n_slots; ++i) iota[i] = static_cast<uint32_t>(i);
        HIP_TRY(hipStreamSynchronize(w->stream));
        HIP_TRY(w->isl_identity.ensure(n_slots * 4));
        HIP_TRY(hipMemcpy(w->isl_identity.p, iota.data(), n_slots * 4, hipMemcpyHostToDevice));
        w->isl_identity_n = n_slots;
    }
    if (w->isl_gen_stale || w->isl_gen.bytes < n_entities * 4) {
        std::vector<uint32_t> gens(n_entities, 0u);
        for (uint64_t e = 0; e < n_entities && e < w->body_gen_host.size(); ++e) gens[e] = w->body_gen_host[e];
        HIP_TRY(hipStreamSynchronize(w->stream));
        HIP_TRY(w->isl_gen.ensure(n_entities * 4));
        HIP_TRY(hipMemcpy(w->isl_gen.p, gens.data(), n_entities * 4, hipMemcpyHostToDevice));
        w->isl_gen_stale = false;
    }
    // pair_capacity: what bge_world_create was given, or — left to the world — 8 pairs per entity to begin with, doubled whenever a
    // sub-step finds more (the pair search reports Dynamic-Static pairs too: a dropped pair would be two bodies passing through each other)
    if (w->isl_pair_cap == 0) w->isl_pair_cap = w->pair_capacity_req ? w->pair_capacity_req : std::max<uint64_t>(8 * w->flat.n_entities, 4096);
    if (!w->pair_capacity_req) w->isl_pair_cap = std::max<uint64_t>(w->isl_pair_cap, std::max<uint64_t>(8 * w->flat.n_entities, 4096));

    bge::IslandParams ip{};
    ip.dt = gp.dt;
    ip.gx = gp.gx;
    ip.gy = gp.gy;
    ip.gz = gp.gz;
    ip.n_slots = n_slots;
    ip.repose = gp.repose;
    ip.entity_of_slot = w->entity_of_slot.as<uint32_t>();
    ip.slot_of_entity = w->slot_of_entity.as<uint32_t>();
    ip.gen_of_entity = w->isl_gen.as<uint32_t>();
    ip.counts = w->isl_counts.as<uint32_t>();
    ip.parent = w->isl_slot_words.as<uint32_t>();
    ip.member = ip.parent + n_slots;
    ip.active = ip.member + n_slots;
    ip.index_of_slot = ip.active + n_slots;
    gp.entity_of_slot = w->entity_of_slot.as<uint32_t>();

    HIP_TRY(hipMemsetAsync(w->isl_counts.p, 0, 12, w->stream)); // (word 3, the error bits, stays: the solver's are read with the NEXT sub-step's counts)
    HIP_TRY(hipMemsetAsync(w->isl_counts.as<uint32_t>() + 4, 0, 16, w->stream));
    if (w->static_contacts) HIP_TRY(bge::launch_obstacles(w->stream, w->view, gp));
    gp.obstacles_ready = 1u;
    HIP_TRY(bge::launch_island_begin(w->stream, w->view, gp, ip, bullet_basis));
    gp.repose = 0u; // (done: k_ground_select must not derive the quaternions a second time from the angles k_island_begin wrote)
    bge::FilterPalette palette{};
    if (int prc = filter_palette_of(w, &palette)) return prc;
    // The tick's own broadphase (BGE_TICK_BROADPHASE: pairs for the caller, the trigger query) sees exactly these boxes after k_tick — the
    // fed AABBs of the sub-step's start — so when it is asked for, it runs HERE, once, and the tick skips its run (bp_shared).  Its
    // capacity is the world's; should it drop pairs, the island path repeats the search with its own instance, which grows.
    bool shared = tick_wants_pairs;
    w->bp_shared = false;
    for (;;) {
        bge::Broadphase& bp = shared ? w->broadphase : w->island_bp;
        const uint64_t cap = shared ? (w->pair_capacity_req ? w->pair_capacity_req : std::max<uint64_t>(8 * w->flat.n_entities, 4096)) : w->isl_pair_cap;
        int rc = bp.configure(std::max<uint64_t>(w->flat.n_slots, bge::kTile), cap);
        if (rc != BGE_OK) return fail(rc, "broadphase allocation failed: %s", bp.error());
        HIP_TRY(w->isl_keys_raw.ensure(cap * 8));
        ip.keys_raw = w->isl_keys_raw.as<uint64_t>();
        ip.pair_cap = static_cast<uint32_t>(std::min<uint64_t>(cap, 0xffffffffu));
        ip.bp_ids_are_entities = shared ? 1u : 0u;
        rc = bp.run(w->stream, w->view, n_slots, shared ? w->entity_of_slot.as<uint32_t>() : w->isl_identity.as<uint32_t>(), nullptr, &palette, nullptr);
        if (rc != BGE_OK) return fail(rc, "broadphase failed: %s", bp.error());
        const bge::PairSlices sl = bp.slices();
        ip.bp_stage = sl.stage;
        ip.bp_counts = sl.counts;
        ip.bp_shard_cap = sl.shard_cap;
        ip.bp_shards = sl.shards;
        HIP_TRY(bge::launch_island_pair_keys(w->stream, w->view, ip));
        HIP_TRY(hipMemcpyAsync(w->isl_counts_host, w->isl_counts.p, 16, hipMemcpyDeviceToHost, w->stream));
        HIP_TRY(hipStreamSynchronize(w->stream));
        if (w->isl_counts_host[3] & 5u) {
            HIP_TRY(hipMemsetAsync(w->isl_counts.p, 0, 64, w->stream));
            return fail(BGE_ERR_HIP, "internal error in the island solver of the previous sub-step (bits %#x): row pool exhausted or an obstacle record missing",
                        w->isl_counts_host[3]);
        }
        if (!(w->isl_counts_host[3] & 2u) && w->isl_counts_host[0] <= ip.pair_cap) {
            w->bp_shared = shared;
            break;
        }
        HIP_TRY(hipMemsetAsync(w->isl_counts.p, 0, 64, w->stream));
        if (shared) { // (the tick's run will report what the world's capacity does to ITS pair list; the islands need every pair)
            shared = false;
            continue;
        }
        if (w->pair_capacity_req || cap >= (1ull << 31)) {
            return fail(BGE_ERR_INVALID, "more overlapping body pairs than pair_capacity (%llu) holds: create the world with a larger pair_capacity",
                        (unsigned long long)cap);
        }
        w->isl_pair_cap = cap * 2; // (the pair search is repeated on the same boxes: nothing else of the sub-step has run yet)
    }
    const uint32_t n_pairs = w->isl_counts_host[0];
    w->isl_last_pairs = n_pairs;
    w->isl_last_bodies = 0;
    const int cur = w->isl_cur, prev = cur ^ 1;
    if (n_pairs) {
        HIP_TRY(w->isl_keys[cur].ensure(static_cast<size_t>(n_pairs) * 8));
        HIP_TRY(w->isl_man[cur].ensure(static_cast<size_t>(n_pairs) * bge::kBoxManifoldWords * 4));
        size_t tmp_bytes = 0;
        HIP_TRY(bge::island_sort_keys(w->stream, nullptr, tmp_bytes, ip.keys_raw, w->isl_keys[cur].as<uint64_t>(), n_pairs));
        HIP_TRY(w->isl_sort_tmp.ensure(std::max<size_t>(tmp_bytes, 16)));
        HIP_TRY(bge::island_sort_keys(w->stream, w->isl_sort_tmp.p, tmp_bytes, ip.keys_raw, w->isl_keys[cur].as<uint64_t>(), n_pairs));
    }
    ip.keys = w->isl_keys[cur].as<uint64_t>();
    ip.man = w->isl_man[cur].as<uint32_t>();
    ip.n_pairs = n_pairs;
    ip.prev_keys = w->isl_keys[prev].as<uint64_t>();
    ip.prev_man = w->isl_man[prev].as<uint32_t>();
    ip.n_prev = w->isl_n_prev;
    const uint64_t body_cap = std::max<uint64_t>(n_slots, 1);
    HIP_TRY(w->isl_body_keys_raw.ensure(body_cap * 8));
    HIP_TRY(w->isl_body_slot_raw.ensure(body_cap * 4));
    ip.body_keys_raw = w->isl_body_keys_raw.as<uint64_t>();
    ip.body_slot_raw = w->isl_body_slot_raw.as<uint32_t>();
    ip.body_cap = static_cast<uint32_t>(body_cap);
    w->isl_cur = prev; // (this sub-step's list is the next one's "previous", also when the rest finds nothing to do)
    w->isl_n_prev = n_pairs;
    if (n_pairs == 0 && !later_sub_step) return BGE_OK;
    HIP_TRY(bge::launch_island_build(w->stream, w->view, ip, later_sub_step));
    HIP_TRY(hipMemcpyAsync(w->isl_counts_host, w->isl_counts.p, 16, hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    const uint32_t n_bodies = w->isl_counts_host[1];
    w->isl_last_bodies = n_bodies;
    if (n_bodies == 0) return BGE_OK;
    HIP_TRY(w->isl_body_keys.ensure(static_cast<size_t>(n_bodies) * 8));
    HIP_TRY(w->isl_body_slot.ensure(static_cast<size_t>(n_bodies) * 4));
    {
        size_t tmp_bytes = 0;
        HIP_TRY(bge::island_sort_pairs(w->stream, nullptr, tmp_bytes, ip.body_keys_raw, w->isl_body_keys.as<uint64_t>(), ip.body_slot_raw,
                                       w->isl_body_slot.as<uint32_t>(), n_bodies));
        HIP_TRY(w->isl_sort_tmp.ensure(std::max<size_t>(tmp_bytes, 16)));
        HIP_TRY(bge::island_sort_pairs(w->stream, w->isl_sort_tmp.p, tmp_bytes, ip.body_keys_raw, w->isl_body_keys.as<uint64_t>(), ip.body_slot_raw,
                                       w->isl_body_slot.as<uint32_t>(), n_bodies));
    }
    ip.body_keys = w->isl_body_keys.as<uint64_t>();
    ip.body_slot = w->isl_body_slot.as<uint32_t>();
    ip.n_bodies = n_bodies;
    // every point the manifolds can hold: 4 on the plane + 4 x 4 on obstacles per body (where those are on), 4 per pair; two rows a point
    const uint64_t own_points = (w->ground_plane ? 4ull : 0ull) + (w->static_contacts_ever ? 4ull * bge::kBoxManifolds : 0ull); // (rows of obstacle manifolds outlive the switch)
    const uint64_t row_cap = 2ull * (own_points * n_bodies + 4ull * n_pairs) + 2;
    HIP_TRY(w->isl_solver_bodies.ensure(static_cast<size_t>(n_bodies) * bge::kIslBodyBytes));
    HIP_TRY(w->isl_rows.ensure(static_cast<size_t>(row_cap) * bge::kIslRowBytes + static_cast<size_t>(row_cap / 2 + 1) * bge::kIslRowColdBytes));
    ip.solver_bodies = w->isl_solver_bodies.p;
    ip.rows = w->isl_rows.p;
    ip.rows_cold = static_cast<char*>(w->isl_rows.p) + static_cast<size_t>(row_cap) * bge::kIslRowBytes;
    ip.row_cap = static_cast<uint32_t>(std::min<uint64_t>(row_cap, 0xffffffffu));
    // islands a workgroup solves level by level (k_island_solve_big): the list of them, two words per body, four per point for the levels
    const uint64_t int_cap = 2ull * row_cap + 8ull * n_bodies + 64;
    HIP_TRY(w->isl_big_list.ensure(static_cast<size_t>(n_bodies) * 16)); // (two lists: big islands, mid islands)
    HIP_TRY(w->isl_body_words.ensure(static_cast<size_t>(n_bodies) * 20 + 16)); // (two words a body for the workgroup solver, pair_first, row_count, row_first)
    const size_t scan_bytes = bge::island_scan_bytes(n_bodies);
    HIP_TRY(w->isl_sort_tmp.ensure(std::max<size_t>(scan_bytes, 16)));
    HIP_TRY(w->isl_ints.ensure(static_cast<size_t>(int_cap) * 4));
    ip.big_list = w->isl_big_list.as<uint32_t>();
    ip.mid_list = ip.big_list + 2ull * n_bodies;
    ip.body_words = w->isl_body_words.as<uint32_t>();
    ip.pair_first = ip.body_words + 2ull * n_bodies;
    ip.row_count = ip.pair_first + n_bodies;
    ip.row_first = ip.row_count + n_bodies;
    ip.scan_tmp = w->isl_sort_tmp.p;
    ip.scan_tmp_bytes = scan_bytes;
    ip.ints = w->isl_ints.as<uint32_t>();
    ip.int_cap = static_cast<uint32_t>(std::min<uint64_t>(int_cap, 0xffffffffu));
    ip.big_points = w->isl_big_points;
    ip.iterations = 10u;
    if (const char* e = std::getenv("BGE_ISLAND_ITERATIONS")) ip.iterations = static_cast<uint32_t>(std::strtoul(e, nullptr, 10)); // (measurements: what the set-up costs)
    HIP_TRY(bge::launch_island_solve(w->stream, w->view, gp, ip, bullet_basis));
    return BGE_OK;
}

// sum the recorded event pairs into the carry (synchronises the stream)
int fold_profile(bge_world* w)
{
    HIP_TRY(hipStreamSynchronize(w->stream));
    for (size_t k = 0; k + 1 < w->prof_used; k += 2) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, w->prof_events[k], w->prof_events[k + 1]));
        w->prof_ms_carry += ms;
        w->prof_ticks_carry += (k / 2 < w->prof_ticks_pending.size()) ? w->prof_ticks_pending[k / 2] : 1u;
    }
    w->prof_used = 0;
    w->prof_ticks_pending.clear();
    return BGE_OK;
}
} // namespace

extern "C" {

int bge_world_gather_roots(bge_world* w, void** table_device);

const char* bge_last_error(void) { return g_last_error.c_str(); }
uint32_t bge_version(void) { return 0x00010000u; }

int bge_world_create(const bge_world_desc* desc, bge_world** out)
try {
    if (!out) return fail(BGE_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (desc && desc->struct_size != 0 && desc->struct_size < sizeof(bge_world_desc)) {
        return fail(BGE_ERR_INVALID, "bge_world_desc.struct_size %u < %zu", desc->struct_size, sizeof(bge_world_desc));
    }
    int ndev = 0;
    {
        const hipError_t e = hipGetDeviceCount(&ndev);
        if (e != hipSuccess || ndev <= 0) {
            return fail(BGE_ERR_HIP, "no usable HIP device (hipGetDeviceCount: %s, count %d) — this library has no CPU path",
                        hipGetErrorString(e), ndev);
        }
    }
    int device = desc ? desc->device : -1;
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return fail(BGE_ERR_INVALID, "device %d out of range (count %d)", device, ndev);

    bge_world* w = new (std::nothrow) bge_world();
    if (!w) return fail(BGE_ERR_OOM, "host allocation failed");
    w->device = device;
    if (const char* e = std::getenv("BGE_TRIGGER_GRID_MIN")) w->trigger_grid_min = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
    if (const char* e = std::getenv("BGE_TRIGGER_DEVICE_DIFF")) w->trig_device_diff = std::strtoul(e, nullptr, 10) != 0;
    if (const char* e = std::getenv("BGE_OBSTACLE_GRID")) w->obstacle_grid_on = std::strtoul(e, nullptr, 10) != 0;
    DeviceGuard guard(device);
    if (desc && desc->stream) {
        w->stream = static_cast<hipStream_t>(desc->stream);
    } else {
        const hipError_t e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete w;
            return fail(BGE_ERR_HIP, "hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
        }
        w->own_stream = true;
    }
    w->pair_capacity_req = desc ? desc->pair_capacity : 0;
    *out = w;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_create")

void bge_world_destroy(bge_world* w)
{
    if (!w) return;
    DeviceGuard guard(w->device);
    (void)hipStreamSynchronize(w->stream);
    if (w->trig_fast_ticks && std::getenv("BGE_TRIGGER_PROFILE"))
        std::fprintf(stderr, "[bge] trigger diff on the device: %llu ticks, %.3f ms per tick until the deltas were on the host, %.3f ms applying %.0f of them\n",
                     (unsigned long long)w->trig_fast_ticks, w->trig_wait_ms / w->trig_fast_ticks, w->trig_apply_ms / w->trig_fast_ticks,
                     double(w->trig_deltas_seen) / w->trig_fast_ticks);
    w->release_all();
    if (w->own_stream) (void)hipStreamDestroy(w->stream);
    delete w;
}

int bge_world_set_topology(bge_world* w, uint64_t n, const uint32_t* parent, const uint8_t* has_transform)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    if (n >= 0xfffffff0ull) return fail(BGE_ERR_INVALID, "too many entities (%llu)", (unsigned long long)n);
    DeviceGuard guard(w->device);

    // An entity that loses its Transform while its body exists keeps a slot for the body (kValid clear): the reference keeps
    // stepping that Bullet body (PhysicsSystem.cpp:389-393).  To the hierarchy it is an entity without a Transform: it is
    // nobody's effective parent and has none itself.
    std::vector<uint8_t> orphan(n, 0);
    bool any_orphan = false;
    if (w->has_topology && has_transform) {
        const uint64_t lim = std::min<uint64_t>(n, w->flat.n_entities);
        for (uint64_t i = 0; i < lim; ++i) {
            const bool was_orphan = i < w->orphan_host.size() && w->orphan_host[i];
            const bool created = was_orphan || (i < w->body_since.size() && w->physics_updates > w->body_since[i]);
            if (!has_transform[i] && w->flat.slot_of_entity[i] != bge::kNone && i < w->body_type_host.size() &&
                w->body_type_host[i] != BGE_BODY_NONE && created) {
                orphan[i] = 1;
                any_orphan = true;
            }
        }
    }
    // A trigger ghost whose entity loses its Transform stays in the world where it was last posed (EnsureTrigger returns before
    // it touches the ghost, PhysicsSystem.cpp:530-534): its box is fetched now, while the device arrays still describe it.
    if (w->has_topology && has_transform && !w->triggers.empty()) {
        for (size_t k = 0; k < w->triggers.size(); ++k) {
            bge_world::Trigger& t = w->triggers[k];
            const bool keeps_transform = t.entity < n && has_transform[t.entity];
            if (keeps_transform || t.frozen || !t.runtime_active) continue;
            if (t.posed && w->trig_list_on_device && w->trig_aabb.p) {
                HIP_TRY(hipStreamSynchronize(w->stream));
                HIP_TRY(hipMemcpy(t.frozen_aabb, static_cast<const char*>(w->trig_aabb.p) + 24 * k, 24, hipMemcpyDeviceToHost));
                t.frozen = true;
            } else {
                t.runtime_active = false; // (never posed: there is no ghost to keep)
                t.clear_overlaps();
            }
        }
    }
    std::vector<uint32_t> parent_flat;
    std::vector<uint8_t> tf_flat;
    if (any_orphan) {
        parent_flat.assign(n, bge::kNone);
        tf_flat.assign(n, 0);
        for (uint64_t i = 0; i < n; ++i) {
            tf_flat[i] = (has_transform[i] || orphan[i]) ? 1 : 0;
            const uint32_t p = parent ? parent[i] : bge::kNone;
            if (!orphan[i] && p != bge::kNone && p < n && !orphan[p]) parent_flat[i] = p;
        }
    }
    bge::Flattened nf;
    try {
        bge::flatten_topology(n, any_orphan ? parent_flat.data() : parent, any_orphan ? tf_flat.data() : has_transform, nf);
    } catch (const std::bad_alloc&) {
        return fail(BGE_ERR_OOM, "host allocation failed while flattening %llu entities", (unsigned long long)n);
    }
    if (any_orphan) {
        for (uint64_t i = 0; i < n; ++i) {
            if (orphan[i]) nf.flags[nf.slot_of_entity[i]] &= ~bge::kValid;
        }
    }

    // ---- carry component state of surviving entity indices over to the new layout
    const bool carry = w->has_topology && w->flat.n_slots > 0;
    const uint64_t n_keep = carry ? std::min<uint64_t>(n, w->flat.n_entities) : 0;
    struct Carry {
        DevBuf* buf;
        uint32_t width;
        TmpBuf tmp;
    };
    std::vector<Carry> carries;
    TmpBuf old_flags_tmp;
    if (n_keep) {
        for (auto [buf, width] : std::initializer_list<std::pair<DevBuf*, uint32_t>>{
                 {&w->pos, 3}, {&w->euler, 3}, {&w->scale, 3}, {&w->world, 16}, {&w->vel, 3}, {&w->angvel, 3},
                 {&w->quat, 4}, {&w->inv_mass, 1}, {&w->deact, 1}, {&w->filter_class, 1}, {&w->half_extent, 3}, {&w->group, 1}, {&w->mask, 1}, {&w->aabb, 6},
                 {&w->cshape, 4}, {&w->cmass, 1}, {&w->cfriction, 1}, {&w->cinfo, 1}, {&w->manifold, 32}, {&w->crestitution, 1},
                 {&w->bmanifold, bge::kBoxManifolds * bge::kBoxManifoldWords}}) {
            if (buf->p) carries.push_back(Carry{buf, width, TmpBuf{}}); // (the manifold store exists only with the ground plane on)
        }
        for (Carry& c : carries) {
            HIP_TRY(c.tmp.ensure(n_keep * c.width * 4));
            HIP_TRY(bge::launch_gather_rows(w->stream, w->slot_of_entity.as<uint32_t>(), 0, n_keep, c.width, c.buf->p, c.tmp.p));
        }
        HIP_TRY(old_flags_tmp.ensure(n_keep * 4));
        HIP_TRY(bge::launch_gather_rows(w->stream, w->slot_of_entity.as<uint32_t>(), 0, n_keep, 1, w->flags.p, old_flags_tmp.p));
        HIP_TRY(hipStreamSynchronize(w->stream));
    }
    std::vector<uint32_t> old_flags(n_keep);
    if (n_keep) HIP_TRY(hipMemcpy(old_flags.data(), old_flags_tmp.p, n_keep * 4, hipMemcpyDeviceToHost));
    // roots that still keep their stored world matrix (the kernel clears a bit when its root turns dirty), per entity
    std::vector<uint8_t> was_frozen(n_keep, 0);
    if (n_keep && w->any_frozen) {
        std::vector<uint32_t> bits((w->flat.n_slots + 31) / 32);
        HIP_TRY(hipMemcpy(bits.data(), w->frozen.p, bits.size() * 4, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n_keep; ++i) {
            const uint32_t s = w->flat.slot_of_entity[i];
            if (s != bge::kNone && ((bits[s >> 5] >> (s & 31u)) & 1u)) was_frozen[i] = 1;
        }
    }

    // ---- which surviving transforms became dirty through the hierarchy.  Scene::SetParent (Scene.cpp:354-393) works on
    // ENTITIES: a changed parent — whether or not either parent owns a Transform — ends in MarkHierarchyDirty(child), which walks
    // the children lists (through Transform-less entities too) and marks every Transform on the way (Scene.cpp:535-550).  A
    // Transform that merely gained or lost an effective parent because that parent's Transform came or went is NOT marked.
    std::vector<uint8_t> keep_mask(n_keep, 0); // 1 = carried over, 2 = carried over but hierarchy-dirty
    std::vector<uint32_t> raw_new(n, bge::kNone);
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t p = parent ? parent[i] : bge::kNone;
        if (p != bge::kNone && p < n) raw_new[i] = p; // (SetParent ignores a parent that is not alive)
    }
    if (n_keep) {
        std::vector<uint32_t> order; // BFS over the new entity forest so that dirtiness flows parent -> child
        order.reserve(n);
        std::vector<uint32_t> child_begin(n + 1, 0), child_list;
        for (uint64_t i = 0; i < n; ++i) {
            if (raw_new[i] != bge::kNone) child_begin[raw_new[i] + 1]++;
        }
        for (uint64_t i = 0; i < n; ++i) child_begin[i + 1] += child_begin[i];
        child_list.resize(child_begin[n]);
        {
            std::vector<uint32_t> cur(child_begin.begin(), child_begin.end() - 1);
            for (uint64_t i = 0; i < n; ++i) {
                if (raw_new[i] != bge::kNone) child_list[cur[raw_new[i]]++] = static_cast<uint32_t>(i);
            }
        }
        std::vector<uint8_t> hdirty(n, 0);
        for (uint64_t i = 0; i < n; ++i) {
            if (raw_new[i] == bge::kNone) order.push_back(static_cast<uint32_t>(i));
        }
        for (size_t h = 0; h < order.size(); ++h) { // (entities inside a parent cycle are never reached: nothing marks them)
            const uint32_t u = order[h];
            bool d = u >= w->parent_entity.size() || w->parent_entity[u] != raw_new[u];
            if (raw_new[u] != bge::kNone && hdirty[raw_new[u]]) d = true;
            hdirty[u] = d ? 1 : 0;
            for (uint32_t c = child_begin[u]; c < child_begin[u + 1]; ++c) order.push_back(child_list[c]);
        }
        for (uint64_t i = 0; i < n_keep; ++i) {
            if (nf.slot_of_entity[i] == bge::kNone || w->flat.slot_of_entity[i] == bge::kNone) continue;
            keep_mask[i] = hdirty[i] ? 2 : 1;
        }
    }
    // A clean Transform whose parent ENTITY stays but no longer owns a Transform becomes a root without being marked dirty:
    // TransformSystem::Update will not recompute it (UpdateNode recomputes a node only when it or an ancestor is dirty,
    // TransformSystem.cpp:18-40), so its world matrix stays parent * local until something dirties it.  Such roots — and the
    // ones that were already in that state and still have no Transform above them — are listed for the kernel.
    std::vector<uint32_t> frozen_bits;
    bool any_frozen = false;
    for (uint64_t i = 0; i < n_keep; ++i) {
        if (keep_mask[i] != 1 || (old_flags[i] & bge::kTDirty)) continue;
        if (orphan[i]) continue;                    // (a body without a Transform has no world matrix to keep)
        const uint32_t p = raw_new[i];
        const bool parent_owns_transform = p != bge::kNone && nf.slot_of_entity[p] != bge::kNone && !orphan[p];
        if (p == bge::kNone || parent_owns_transform) continue; // no parent entity, or it owns a Transform (again)
        const bool parent_had_transform = p < w->flat.n_entities && w->flat.slot_of_entity[p] != bge::kNone &&
                                          !(p < w->orphan_host.size() && w->orphan_host[p]);
        if (!parent_had_transform && !was_frozen[i]) continue;
        if (frozen_bits.empty()) frozen_bits.assign((std::max<uint64_t>(nf.n_slots, bge::kTile) + 31) / 32, 0u);
        const uint32_t sl = nf.slot_of_entity[i];
        frozen_bits[sl >> 5] |= 1u << (sl & 31u);
        nf.tile_hdr[sl / bge::kTile] |= bge::kHdrFrozen;
        any_frozen = true;
    }
    w->parent_entity.swap(raw_new);

    // ---- (re)allocate for the new layout
    const uint64_t S = std::max<uint64_t>(nf.n_slots, bge::kTile);
    const uint64_t T = std::max<uint32_t>(nf.n_tiles_total, 1);
    HIP_TRY(w->flags.ensure(S * 4));
    HIP_TRY(w->parent.ensure(S * 4));
    HIP_TRY(w->tile_hdr.ensure(T * 4));
    HIP_TRY(w->slot_of_entity.ensure(std::max<uint64_t>(n, 1) * 4));
    HIP_TRY(w->entity_of_slot.ensure(S * 4));
    HIP_TRY(w->root_index.ensure(S * 4));
    HIP_TRY(w->root_slots.ensure(std::max<size_t>(nf.root_slots.size(), 1) * 4));
    HIP_TRY(w->pos.ensure(S * 12));
    HIP_TRY(w->euler.ensure(S * 12));
    HIP_TRY(w->scale.ensure(S * 12));
    HIP_TRY(w->world.ensure(S * 64));
    HIP_TRY(w->vel.ensure(S * 12));
    HIP_TRY(w->angvel.ensure(S * 12));
    HIP_TRY(w->quat.ensure(S * 16));
    HIP_TRY(w->inv_mass.ensure(S * 4));
    HIP_TRY(w->deact.ensure(S * 4));
    HIP_TRY(w->filter_class.ensure(S * 4));
    HIP_TRY(w->filter_table.ensure(256 * 16));
    HIP_TRY(w->half_extent.ensure(S * 12));
    HIP_TRY(w->group.ensure(S * 4));
    HIP_TRY(w->mask.ensure(S * 4));
    HIP_TRY(w->aabb.ensure(S * 24));
    HIP_TRY(w->cshape.ensure(S * 16));
    HIP_TRY(w->cmass.ensure(S * 4));
    HIP_TRY(w->cfriction.ensure(S * 4));
    HIP_TRY(w->cinfo.ensure(S * 4));
    // (the manifold store exists from the first time the plane is switched on and then follows every layout, plane on or off:
    //  the carry above scatters its rows to the NEW slot indices, and k_ground indexes it by slot as soon as the plane is back)
    if (w->ground_plane || w->manifold.p) {
        HIP_TRY(w->manifold.ensure(S * 128));
        HIP_TRY(hipMemsetAsync(w->manifold.p, 0, w->manifold.bytes, w->stream));
    }
    HIP_TRY(w->crestitution.ensure(S * 4)); // (k_init_slots writes the default, the carry below the surviving values)
    if (w->static_contacts || w->dynamic_contacts || w->bmanifold.p) { // (exists from the first bge_world_set_static_contacts(1) on and follows every layout, like the plane's store)
        HIP_TRY(w->bmanifold.ensure(S * bge::kBoxManifolds * bge::kBoxManifoldWords * 4));
        HIP_TRY(hipMemsetAsync(w->bmanifold.p, 0xff, w->bmanifold.bytes, w->stream));
    }
    w->obstacles_stale = true; // slots moved
    HIP_TRY(w->root_worlds.ensure(std::max<size_t>(nf.root_slots.size(), 1) * 64));
    HIP_TRY(w->counter.ensure(64));
    HIP_TRY(w->mass_palette.ensure(256 * sizeof(float2)));
    HIP_TRY(w->grav_palette.ensure(256 * sizeof(float4)));
    w->grav_palette_stale = true;
    w->rebuild_view();

    if (nf.n_slots) {
        HIP_TRY(hipMemcpyAsync(w->parent.p, nf.parent_field.data(), nf.n_slots * 4, hipMemcpyHostToDevice, w->stream));
        HIP_TRY(hipMemcpyAsync(w->entity_of_slot.p, nf.entity_of_slot.data(), nf.n_slots * 4, hipMemcpyHostToDevice, w->stream));
        HIP_TRY(hipMemcpyAsync(w->tile_hdr.p, nf.tile_hdr.data(), static_cast<size_t>(nf.n_tiles_total) * 4,
                               hipMemcpyHostToDevice, w->stream));
        // structural flags travel through the staging buffer, k_init_slots merges them
        HIP_TRY(w->stage.ensure(nf.n_slots * 4));
        HIP_TRY(hipMemcpyAsync(w->stage.p, nf.flags.data(), nf.n_slots * 4, hipMemcpyHostToDevice, w->stream));
        HIP_TRY(bge::launch_init_slots(w->stream, nf.n_slots, w->stage.as<uint32_t>(), w->view));
    }
    if (n) HIP_TRY(hipMemcpyAsync(w->slot_of_entity.p, nf.slot_of_entity.data(), n * 4, hipMemcpyHostToDevice, w->stream));
    std::vector<uint32_t> root_index_host;
    if (!nf.root_slots.empty()) {
        HIP_TRY(hipMemcpyAsync(w->root_slots.p, nf.root_slots.data(), nf.root_slots.size() * 4, hipMemcpyHostToDevice, w->stream));
        root_index_host.assign(nf.n_slots, 0);
        for (size_t k = 0; k < nf.root_slots.size(); ++k) root_index_host[nf.root_slots[k]] = static_cast<uint32_t>(k);
        HIP_TRY(hipMemcpyAsync(w->root_index.p, root_index_host.data(), nf.n_slots * 4, hipMemcpyHostToDevice, w->stream));
    }
    HIP_TRY(hipStreamSynchronize(w->stream));

    if (n_keep) {
        // entities that lost or gained their Transform restart from defaults: route them to "no slot"
        std::vector<uint32_t> carry_map(n_keep, bge::kNone);
        for (uint64_t i = 0; i < n_keep; ++i) {
            if (keep_mask[i]) carry_map[i] = nf.slot_of_entity[i];
        }
        TmpBuf map_dev;
        HIP_TRY(map_dev.ensure(n_keep * 4));
        HIP_TRY(hipMemcpyAsync(map_dev.p, carry_map.data(), n_keep * 4, hipMemcpyHostToDevice, w->stream));
        for (Carry& c : carries) {
            HIP_TRY(bge::launch_scatter_rows(w->stream, map_dev.as<uint32_t>(), 0, n_keep, c.width, c.tmp.p, c.buf->p, nullptr, 0));
        }
        // flags: keep body type / dirty / spin / shape bits of the old word, structure from the new one
        std::vector<uint32_t> merged(nf.flags);
        const uint32_t keep_bits = bge::kTypeMask | bge::kTDirty | bge::kBDirty | bge::kSpin | bge::kMassMask | bge::kDrowsy;
        for (uint64_t s = 0; s < nf.n_slots; ++s) {
            if (merged[s] & bge::kValid) merged[s] |= bge::kTDirty;
        }
        for (uint64_t i = 0; i < n_keep; ++i) {
            if (!keep_mask[i]) continue;
            const uint32_t s = nf.slot_of_entity[i];
            uint32_t f = (nf.flags[s] & ~keep_bits) | (old_flags[i] & keep_bits);
            if (keep_mask[i] == 2) f |= bge::kTDirty;
            if (orphan[i]) f &= ~(bge::kValid | bge::kTDirty); // the body's slot: no Transform here, nothing to be dirty
            merged[s] = f;
        }
        HIP_TRY(hipMemcpyAsync(w->flags.p, merged.data(), nf.n_slots * 4, hipMemcpyHostToDevice, w->stream));
        HIP_TRY(hipStreamSynchronize(w->stream));
        for (Carry& c : carries) c.tmp.release();
        old_flags_tmp.release();
        map_dev.release();
    }

    if (any_frozen) {
        HIP_TRY(w->frozen.ensure(frozen_bits.size() * 4));
        HIP_TRY(hipMemcpy(w->frozen.p, frozen_bits.data(), frozen_bits.size() * 4, hipMemcpyHostToDevice));
        w->rebuild_view();
    }
    w->any_frozen = any_frozen;
    w->flat = std::move(nf);
    w->orphan_host.swap(orphan);
    w->body_type_host.resize(n, BGE_BODY_NONE); // surviving indices keep their body, new ones have none
    for (uint64_t i = 0; i < n; ++i) {
        if (w->flat.slot_of_entity[i] == bge::kNone) w->body_type_host[i] = BGE_BODY_NONE; // no Transform, no body
    }
    w->has_topology = true;
    w->maybe_dirty = true;
    w->triggers_device_stale = true; // slots moved
    w->global_of_slot_stale = true;
    w->pairs_from_slab = false;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_set_topology")

static int upload_trs_impl(bge_world* w, uint64_t first, uint64_t count, const uint32_t* index, const float* pos3,
                           const float* euler3, const float* scale3)
{
    if (count == 0) return BGE_OK;
    DeviceGuard guard(w->device);
    const uint32_t* di = nullptr;
    if (int rc = stage_index(w, count, index, &di)) return rc;
    if (pos3) {
        if (int rc = upload_rows(w, first, count, 3, pos3, w->pos.p, 0, di, bge::kValid)) return rc;
    }
    if (euler3) {
        if (int rc = upload_rows(w, first, count, 3, euler3, w->euler.p, 0, di, bge::kValid)) return rc;
    }
    if (scale3) {
        if (int rc = upload_rows(w, first, count, 3, scale3, w->scale.p, 0, di, bge::kValid)) return rc;
    }
    HIP_TRY(bge::launch_scatter_rows(w->stream, w->slot_of_entity.as<uint32_t>(), first, count, 0, nullptr, nullptr,
                                     w->flags.as<uint32_t>(), bge::kTDirty, di, bge::kValid));
    HIP_TRY(hipStreamSynchronize(w->stream));
    w->maybe_dirty = true;
    return BGE_OK;
}

int bge_world_upload_trs(bge_world* w, uint64_t first, uint64_t count, const float* pos3, const float* euler3,
                         const float* scale3)
try {
    if (int rc = check_range(w, first, count)) return rc;
    return upload_trs_impl(w, first, count, nullptr, pos3, euler3, scale3);
}
BGE_CATCH_ALL("bge_world_upload_trs")

int bge_world_upload_trs_indexed(bge_world* w, uint64_t count, const uint32_t* entity_index, const float* pos3,
                                 const float* euler3, const float* scale3)
try {
    if (int rc = check_range(w, 0, 0)) return rc;
    if (count && !entity_index) return fail(BGE_ERR_INVALID, "entity_index is NULL");
    return upload_trs_impl(w, 0, count, entity_index, pos3, euler3, scale3);
}
BGE_CATCH_ALL("bge_world_upload_trs_indexed")

int bge_world_mark_dirty(bge_world* w, uint64_t first, uint64_t count)
try {
    if (int rc = check_range(w, first, count)) return rc;
    if (count == 0) return BGE_OK;
    DeviceGuard guard(w->device);
    HIP_TRY(bge::launch_scatter_rows(w->stream, w->slot_of_entity.as<uint32_t>(), first, count, 0, nullptr, nullptr,
                                     w->flags.as<uint32_t>(), bge::kTDirty, nullptr, bge::kValid));
    w->maybe_dirty = true;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_mark_dirty")

static int upload_bodies_impl(bge_world* w, uint64_t first, uint64_t count, const uint32_t* index, const uint8_t* type,
                              const float* mass, const uint8_t* shape, const float* size3, const uint32_t* layer,
                              const uint32_t* mask);

int bge_world_upload_bodies(bge_world* w, uint64_t first, uint64_t count, const uint8_t* type, const float* mass,
                            const uint8_t* shape, const float* size3, const uint32_t* layer, const uint32_t* mask)
try {
    if (int rc = check_range(w, first, count)) return rc;
    return upload_bodies_impl(w, first, count, nullptr, type, mass, shape, size3, layer, mask);
}
BGE_CATCH_ALL("bge_world_upload_bodies")

int bge_world_upload_bodies_indexed(bge_world* w, uint64_t count, const uint32_t* entity_index, const uint8_t* type,
                                    const float* mass, const uint8_t* shape, const float* size3, const uint32_t* layer,
                                    const uint32_t* mask)
try {
    if (int rc = check_range(w, 0, 0)) return rc;
    if (count && !entity_index) return fail(BGE_ERR_INVALID, "entity_index is NULL");
    return upload_bodies_impl(w, 0, count, entity_index, type, mass, shape, size3, layer, mask);
}
BGE_CATCH_ALL("bge_world_upload_bodies_indexed")

static int upload_bodies_impl(bge_world* w, uint64_t first, uint64_t count, const uint32_t* index, const uint8_t* type,
                              const float* mass, const uint8_t* shape, const float* size3, const uint32_t* layer,
                              const uint32_t* mask)
{
    if (!type) return fail(BGE_ERR_INVALID, "type is NULL");
    if (count == 0) return BGE_OK;
    DeviceGuard guard(w->device);
    const uint32_t* di = nullptr;
    if (int rc = stage_index(w, count, index, &di)) return rc;

    // host: component values -> device parameters (PhysicsSystem.cpp:398-477, 686-707)
    bool palette_changed = w->palette_inv_mass.empty();
    if (palette_changed) {
        w->palette_inv_mass.push_back(0.0f);
        w->palette_class[0u] = 0;
    }
    std::vector<uint32_t> words(count * 13);
    float* cdims = reinterpret_cast<float*>(words.data() + 8 * count); // collider as Bullet holds it (ground contact)
    float* cmass = reinterpret_cast<float*>(words.data() + 11 * count);
    uint32_t* cbits = words.data() + 12 * count;
    uint32_t* type_bits = words.data();
    float* inv_mass = reinterpret_cast<float*>(words.data() + count);
    float* he = reinterpret_cast<float*>(words.data() + 2 * count);
    uint32_t* group = words.data() + 5 * count;
    uint32_t* msk = words.data() + 6 * count;
    uint32_t* fclass = words.data() + 7 * count;
    for (uint64_t i = 0; i < count; ++i) {
        const uint8_t t = type[i];
        if (t != BGE_BODY_NONE && t > BGE_BODY_KINEMATIC) return fail(BGE_ERR_INVALID, "type[%llu] = %u", (unsigned long long)i, t);
        const uint8_t sh = shape ? shape[i] : BGE_SHAPE_BOX;
        const float default_size[3] = {0.5f, 0.5f, 0.5f};
        const float* sz = size3 ? size3 + 3 * i : default_size;
        if (t == BGE_BODY_NONE) {
            type_bits[i] = 0;
            inv_mass[i] = 0.0f;
        } else {
            float m = 0.0f;
            if (t == BGE_BODY_DYNAMIC) m = std::max(mass ? mass[i] : 1.0f, 0.01f);
            inv_mass[i] = m != 0.0f ? 1.0f / m : 0.0f;
            type_bits[i] = (static_cast<uint32_t>(t) + 1u) | bge::kBDirty | (mass_class(w, inv_mass[i], palette_changed) << bge::kMassShift);
        }
        {
            const uint64_t e = index ? index[i] : first + i;
            const bool orphan = e < w->orphan_host.size() && w->orphan_host[e];
            if (e < w->body_type_host.size() && !(orphan && t != BGE_BODY_NONE)) {
                const uint8_t now = w->flat.slot_of_entity[e] == bge::kNone ? BGE_BODY_NONE : t;
                if (w->body_type_host[e] == BGE_BODY_NONE && now != BGE_BODY_NONE) {
                    if (w->body_since.size() < w->body_type_host.size()) w->body_since.resize(w->body_type_host.size(), 0);
                    w->body_since[e] = w->physics_updates; // not in the reference's world before the next update
                }
                w->body_type_host[e] = now;
                if (w->body_shape_host.size() < w->body_type_host.size()) w->body_shape_host.resize(w->body_type_host.size(), BGE_SHAPE_BOX);
                if (w->body_gen_host.size() < w->body_type_host.size()) w->body_gen_host.resize(w->body_type_host.size(), 0u);
                w->body_shape_host[e] = sh;
                w->body_gen_host[e] += 1u; // EnsureRigidBody re-creates the btRigidBody: its pairs and their manifolds are gone
                w->isl_gen_stale = true;
                w->obstacles_stale = true;
            }
        }
        collider_half_extents(sh, sz, he + 3 * i);
        const uint32_t l = layer ? layer[i] : 1u;
        group[i] = l ? l : 1u;
        msk[i] = mask ? mask[i] : 0xffffffffu;
        if (sh == BGE_SHAPE_CAPSULE) {
            // btCapsuleShape(radius, 2 * halfHeight): m_implicitShapeDimensions = (radius, 0.5 * height, radius)
            cdims[3 * i] = std::max(sz[0], 0.01f);
            cdims[3 * i + 1] = 0.5f * (std::max(sz[1], 0.0f) * 2.0f);
            cdims[3 * i + 2] = cdims[3 * i];
        } else {
            std::memcpy(cdims + 3 * i, he + 3 * i, 12); // btBoxShape::getHalfExtentsWithMargin() is what the AABB uses too
        }
        cmass[i] = t == BGE_BODY_DYNAMIC ? std::max(mass ? mass[i] : 1.0f, 0.01f) : 0.0f;
        // the ground is in group StaticFilter (2) with mask AllFilter: it reaches the bodies whose mask has bit 1
        cbits[i] = (sh == BGE_SHAPE_CAPSULE ? bge::kCiCapsule : 0u) | ((msk[i] & 2u) ? bge::kCiGroundMask : 0u);
        fclass[i] = t == BGE_BODY_NONE ? 0u : w->filter_class_of(group[i], msk[i], t == BGE_BODY_STATIC);
    }
    if (palette_changed) {
        std::vector<float2> pal(256, float2{0.0f, 0.0f});
        for (size_t k = 0; k < w->palette_inv_mass.size(); ++k) {
            const float im = w->palette_inv_mass[k];
            pal[k] = float2{im, im != 0.0f ? 1.0f / im : 0.0f}; // same IEEE divide the kernel's fallback path performs
        }
        w->grav_palette_stale = true;
        HIP_TRY(hipMemcpyAsync(w->mass_palette.p, pal.data(), pal.size() * sizeof(float2), hipMemcpyHostToDevice, w->stream));
        HIP_TRY(hipStreamSynchronize(w->stream));
    }
    {
        // tile headers: "every valid slot carries a Dynamic body" lets the tick kernel load velocities without waiting for flags
        std::vector<uint32_t> dyn_in_tile(w->flat.n_tiles_total, 0);
        for (uint64_t e = 0; e < w->flat.n_entities; ++e) {
            const uint32_t sl = w->flat.slot_of_entity[e];
            if (sl != bge::kNone && w->body_type_host[e] == BGE_BODY_DYNAMIC) dyn_in_tile[sl / bge::kTile]++;
        }
        bool changed = false;
        for (uint32_t t = 0; t < w->flat.n_tiles_total; ++t) {
            const uint32_t count = (w->flat.tile_hdr[t] >> bge::kHdrCountShift) & bge::kHdrCountMask;
            const uint32_t want = (count != 0 && dyn_in_tile[t] == count) ? bge::kHdrAllDynamic : 0u;
            if ((w->flat.tile_hdr[t] & bge::kHdrAllDynamic) != want) {
                w->flat.tile_hdr[t] = (w->flat.tile_hdr[t] & ~bge::kHdrAllDynamic) | want;
                changed = true;
            }
        }
        if (changed) {
            HIP_TRY(hipStreamSynchronize(w->stream));
            HIP_TRY(hipMemcpy(w->tile_hdr.p, w->flat.tile_hdr.data(), static_cast<size_t>(w->flat.n_tiles_total) * 4, hipMemcpyHostToDevice));
            w->drop_graph();
        }
    }
    const size_t bytes = words.size() * 4;
    HIP_TRY(w->stage.ensure(bytes));
    HIP_TRY(hipMemcpyAsync(w->stage.p, words.data(), bytes, hipMemcpyHostToDevice, w->stream));
    const uint32_t* d = w->stage.as<uint32_t>();
    HIP_TRY(bge::launch_scatter_bodies(w->stream, w->slot_of_entity.as<uint32_t>(), first, count, d,
                                       reinterpret_cast<const float*>(d + count), reinterpret_cast<const float*>(d + 2 * count),
                                       d + 5 * count, d + 6 * count, d + 7 * count, w->view, di,
                                       reinterpret_cast<const float*>(d + 8 * count), reinterpret_cast<const float*>(d + 11 * count),
                                       d + 12 * count));
    HIP_TRY(hipStreamSynchronize(w->stream));
    w->maybe_dirty = true;
    return BGE_OK;
}

int bge_world_set_velocities(bge_world* w, uint64_t first, uint64_t count, const float* linvel3, const float* angvel3)
try {
    if (int rc = check_range(w, first, count)) return rc;
    if (count == 0 || (!linvel3 && !angvel3)) return BGE_OK;
    DeviceGuard guard(w->device);
    const size_t rows = static_cast<size_t>(count) * 12;
    HIP_TRY(w->stage.ensure(rows * 2));
    float* dl = w->stage.as<float>();
    float* da = reinterpret_cast<float*>(static_cast<char*>(w->stage.p) + rows);
    if (linvel3) HIP_TRY(hipMemcpyAsync(dl, linvel3, rows, hipMemcpyHostToDevice, w->stream));
    if (angvel3) HIP_TRY(hipMemcpyAsync(da, angvel3, rows, hipMemcpyHostToDevice, w->stream));
    HIP_TRY(bge::launch_scatter_velocities(w->stream, w->slot_of_entity.as<uint32_t>(), first, count, linvel3 ? dl : nullptr,
                                           angvel3 ? da : nullptr, w->view));
    HIP_TRY(hipStreamSynchronize(w->stream));
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_set_velocities")

namespace {
// How one enqueued tick relates to PhysicsSystem::Update's stepSimulation call (bge_world_step_simulation):
struct SubStep {
    bool no_repose = false;    // a later sub-step of the same call: the teleport rule ran before the first one
    bool ghosts_posed = false; // the trigger ghosts were posed (and their activation rule applied) before the first sub-step
};
int tick_impl(bge_world* w, uint32_t ticks, float dt, const float gravity[3], uint32_t flags, SubStep sub);
} // namespace

int bge_world_tick_many(bge_world* w, uint32_t ticks, float dt, const float gravity[3], uint32_t flags)
try {
    return tick_impl(w, ticks, dt, gravity, flags, SubStep{});
}
BGE_CATCH_ALL("bge_world_tick_many")

namespace {
int tick_impl(bge_world* w, uint32_t ticks, float dt, const float gravity[3], uint32_t flags, SubStep sub)
{
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
    if ((flags & (BGE_TICK_PHYSICS | BGE_TICK_TRANSFORMS)) == 0) return fail(BGE_ERR_INVALID, "tick flags select nothing");
    if ((flags & (BGE_TICK_BROADPHASE | BGE_TICK_AABBS)) && !(flags & BGE_TICK_PHYSICS)) {
        return fail(BGE_ERR_INVALID, "BGE_TICK_BROADPHASE / BGE_TICK_AABBS need BGE_TICK_PHYSICS (the AABBs come from the physics step)");
    }
    if ((flags & BGE_TICK_BULLET_BASIS) && !(flags & BGE_TICK_PHYSICS)) {
        return fail(BGE_ERR_INVALID, "BGE_TICK_BULLET_BASIS needs BGE_TICK_PHYSICS (it selects how the physics step carries orientations)");
    }
    if ((flags & BGE_TICK_PHYSICS) && !gravity) return fail(BGE_ERR_INVALID, "gravity is NULL");
    if ((flags & BGE_TICK_NORMAL_MATRICES) && !(flags & BGE_TICK_TRANSFORMS)) {
        return fail(BGE_ERR_INVALID, "BGE_TICK_NORMAL_MATRICES needs BGE_TICK_TRANSFORMS (they are derived from the new world matrices)");
    }
    DeviceGuard guard(w->device);
    if ((flags & BGE_TICK_NORMAL_MATRICES) && w->normal.bytes < std::max<uint64_t>(w->flat.n_slots, bge::kTile) * 64) {
        HIP_TRY(hipStreamSynchronize(w->stream));
        HIP_TRY(w->normal.ensure(std::max<uint64_t>(w->flat.n_slots, bge::kTile) * 64));
        HIP_TRY(hipMemsetAsync(w->normal.p, 0, w->normal.bytes, w->stream));
        w->rebuild_view();
    }
    const bool phys = (flags & BGE_TICK_PHYSICS) != 0;
    const bool xform = (flags & BGE_TICK_TRANSFORMS) != 0;
    if (phys) w->physics_updates += ticks; // (bodies uploaded before this call are in the world from now on)
    if (phys && (w->grav_palette_stale || gravity[0] != w->grav_cached[0] || gravity[1] != w->grav_cached[1] ||
                 gravity[2] != w->grav_cached[2] || std::memcmp(gravity, w->grav_cached, 12) != 0)) {
        // btRigidBody::setGravity: m_gravity = acceleration / m_inverseMass, one IEEE division per component (the same
        // correctly rounded binary32 division on the host as the kernel's fallback path performs on the device)
        std::vector<float> tab(256 * 4, 0.0f);
        for (size_t k = 0; k < w->palette_inv_mass.size() && k < 256; ++k) {
            const float im = w->palette_inv_mass[k];
            if (im != 0.0f) {
                tab[4 * k] = gravity[0] / im;
                tab[4 * k + 1] = gravity[1] / im;
                tab[4 * k + 2] = gravity[2] / im;
            }
            tab[4 * k + 3] = im;
        }
        HIP_TRY(hipStreamSynchronize(w->stream)); // ticks in flight still read the old table
        HIP_TRY(hipMemcpy(w->grav_palette.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
        std::memcpy(w->grav_cached, gravity, 12);
        w->grav_palette_stale = false;
        w->drop_graph();
    }
    if (w->profiling) {
        // room for every pair of this call, so that no mid-run synchronisation is needed
        const size_t need = w->prof_used + 2 * static_cast<size_t>(w->profiling == 2 ? ticks : 1);
        while (w->prof_events.size() < need && w->prof_events.size() < (1u << 20)) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            w->prof_events.push_back(e);
        }
    }
    // Optional (BGE_USE_GRAPH=1): replay a captured hipGraph of 32 ticks instead of issuing every launch from the host.
    // Same kernels, same order, same results — but MEASURED SLOWER on ROCm 7.2 / MI355X: 10 k entities 5.9 us per tick
    // against 3.2 us eager, 100 k 7.4 against 4.5, 1 M equal (graph kernel nodes cost more than back-to-back eager
    // launches on one stream), so it is off by default.
    uint32_t first_eager = 0;
    const bool use_graph = std::getenv("BGE_USE_GRAPH") != nullptr;
    if (use_graph && !sub.no_repose && !w->ground_plane && !w->dynamic_contacts && !w->graph_disabled && phys && ticks >= 2 * bge_world::kGraphTicks && w->profiling != 2 &&
        !(flags & (BGE_TICK_BROADPHASE | BGE_TICK_AABBS | BGE_TICK_GATHER_ROOTS)) && w->flat.n_tiles_ticked <= bge_world::kGraphMaxTiles &&
        w->flat.n_tiles_ticked > 0) {
        const bool same = w->graph_exec && w->graph_flags == flags && w->graph_dt == dt && w->graph_g[0] == gravity[0] &&
                          w->graph_g[1] == gravity[1] && w->graph_g[2] == gravity[2];
        if (!same) {
            w->drop_graph();
            hipGraph_t graph = nullptr;
            bool ok = hipStreamBeginCapture(w->stream, hipStreamCaptureModeRelaxed) == hipSuccess;
            if (ok) {
                bge::TickParams p{};
                p.dt = dt;
                p.gx = gravity[0];
                p.gy = gravity[1];
                p.gz = gravity[2];
                p.nt_out = 0; // graph replay is limited to small scenes
                w->fill_sleep(p);
                for (uint32_t t = 0; ok && t < bge_world::kGraphTicks; ++t) {
                    for (size_t pass = 0; ok && pass + 1 < w->flat.pass_tile_begin.size(); ++pass) {
                        p.tile_begin = w->flat.pass_tile_begin[pass];
                        ok = bge::launch_tick(w->stream, w->view, p, w->flat.pass_tile_begin[pass + 1] - p.tile_begin, flags) == hipSuccess;
                    }
                }
                ok = (hipStreamEndCapture(w->stream, &graph) == hipSuccess) && ok && graph != nullptr;
            }
            if (ok) ok = hipGraphInstantiate(&w->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
            if (!ok) {
                (void)hipGetLastError();
                w->drop_graph();
                w->graph_disabled = true; // capture is not available here: keep issuing launches eagerly
            } else {
                w->graph_flags = flags;
                w->graph_dt = dt;
                w->graph_g[0] = gravity[0];
                w->graph_g[1] = gravity[1];
                w->graph_g[2] = gravity[2];
            }
        }
        if (w->graph_exec) {
            const uint32_t chunks = ticks / bge_world::kGraphTicks;
            if (w->profiling == 1) {
                if (w->prof_used + 2 > w->prof_events.size()) {
                    if (int rc = fold_profile(w)) return rc;
                }
                HIP_TRY(hipEventRecord(w->prof_events[w->prof_used], w->stream));
            }
            for (uint32_t c = 0; c < chunks; ++c) HIP_TRY(hipGraphLaunch(w->graph_exec, w->stream));
            first_eager = chunks * bge_world::kGraphTicks;
            w->maybe_dirty = phys && !xform;
            if (w->profiling == 1 && first_eager == ticks) {
                HIP_TRY(hipEventRecord(w->prof_events[w->prof_used + 1], w->stream));
                w->prof_used += 2;
                w->prof_ticks_pending.push_back(ticks);
            }
        }
    }
    // Output stores: non-temporal once the tick's working set (~140 B per slot, 204 B with normal matrices) no longer
    // fits the 256 MiB Infinity Cache; BGE_NT_STORES=0/1 overrides (experiments)
    bool nt_out = static_cast<double>(w->flat.n_slots) * ((flags & BGE_TICK_NORMAL_MATRICES) ? 204.0 : 140.0) > kNtThresholdBytes;
    if (const char* e = std::getenv("BGE_NT_STORES")) nt_out = std::atoi(e) != 0;
    for (uint32_t t = first_eager; t < ticks; ++t) {
        if (!phys && !w->maybe_dirty && !(flags & BGE_TICK_NORMAL_MATRICES)) {
            // TransformSystem::Update with nothing dirty: a no-op scan — only the kernel launches are skipped.  The
            // collective is NOT: a peer rank may have dirty transforms and issue its gather, and a rank that stayed
            // away would leave it hanging (and let the ranks' ring frame counters drift apart).  The unchanged roots are
            // packed from the world array and gathered like any other frame.
            if (flags & BGE_TICK_GATHER_ROOTS) {
                if (int rc = bge_world_gather_roots(w, nullptr)) return rc;
            }
            continue;
        }
        bge::TickParams p{};
        p.dt = dt;
        p.gx = gravity ? gravity[0] : 0.0f;
        p.gy = gravity ? gravity[1] : 0.0f;
        p.gz = gravity ? gravity[2] : 0.0f;
        p.nt_out = nt_out ? 1u : 0u;
        p.no_repose = sub.no_repose ? 1u : 0u;
        w->fill_sleep(p);
        const bool with_triggers = (flags & BGE_TICK_BROADPHASE) && !w->triggers.empty();
        if (with_triggers && !sub.ghosts_posed) {
            ensure_triggers(w);
            if (w->triggers_device_stale) {
                if (int rc = sync_triggers_to_device(w)) return rc;
            }
            // ghost boxes from the Transforms as they are before the step
            HIP_TRY(bge::launch_trigger_aabb(w->stream, static_cast<uint32_t>(w->triggers.size()), w->trigger_view(), w->view));
            for (bge_world::Trigger& tg : w->triggers) tg.posed = tg.posed || (tg.runtime_active && !tg.frozen);
        }
        if (flags & BGE_TICK_BROADPHASE) {
            // the broadphase takes its grid from per-wave partials the tick kernel writes beside the AABBs
            const size_t need = std::max<size_t>(w->flat.n_tiles_total, 1) * 4 * 32;
            if (w->bp_partials.bytes < need) {
                HIP_TRY(hipStreamSynchronize(w->stream));
                HIP_TRY(w->bp_partials.ensure(need));
            }
            p.bp_partial = w->bp_partials.as<float4>();
        }
        if (phys && (w->ground_plane || w->static_contacts || w->dynamic_contacts)) {
            // Ground plane on.  Bullet's order inside PhysicsSystem::Update: teleport dirty bodies (before stepSimulation), then per
            // sub-step collision detection + solver, then integrateTransforms — so k_ground_select re-poses them (once per
            // stepSimulation call) and picks the bodies at the ground, k_ground collides and solves those, and the tick kernel
            // integrates.
            const uint64_t n_slots = static_cast<uint64_t>(w->flat.n_tiles_ticked) * bge::kTile;
            // the solver's work list: slots + its count and ticket words (left at zero by every k_ground)
            const uint64_t shard_cap = bge::ground_shard_cap(n_slots);
            const size_t list_bytes = shard_cap * bge::kGroundShards * 4, count_bytes = (bge::kGroundShards + 1) * 64;
            if (w->ground_list.bytes < list_bytes || !w->ground_count.p) {
                HIP_TRY(w->ground_list.ensure(list_bytes));
                HIP_TRY(w->ground_count.ensure(count_bytes));
                HIP_TRY(hipMemsetAsync(w->ground_count.p, 0, count_bytes, w->stream));
                w->drop_graph();
            }
            bge::GroundParams gp{};
            gp.repose = sub.no_repose ? 0u : 1u; // (the teleport rule is part of k_ground_select)
            gp.list = w->ground_list.as<uint32_t>();
            gp.list_count = w->ground_count.as<uint32_t>();
            gp.shard_cap = shard_cap;
            gp.dt = dt;
            gp.gx = gravity[0];
            gp.gy = gravity[1];
            gp.gz = gravity[2];
            gp.n_slots = n_slots;
            gp.want_aabb = (flags & (BGE_TICK_BROADPHASE | BGE_TICK_AABBS)) ? 1u : 0u;
            gp.plane = w->ground_plane ? 1u : 0u;
            if (w->static_contacts) {
                if (int rc = prepare_obstacles(w, n_slots)) return rc;
                gp.obstacle_slots = w->obstacle_slots.as<uint32_t>();
                gp.obstacle_gen = w->obstacle_gen.as<uint32_t>();
                gp.obstacles = w->obstacles.as<bge::ObstacleRec>();
                gp.n_obstacles = w->n_obstacles;
                if (w->obstacle_grid_on && w->n_obstacles > bge::kObstacleGridMin) {
                    gp.obstacle_grid = w->obstacle_grid.as<uint32_t>();
                    gp.obstacle_grid_cap = 64u * w->n_obstacles;
                }
                gp.entity_of_slot = w->entity_of_slot.as<uint32_t>();
                gp.box_list = w->box_list.as<uint32_t>();
                gp.box_count = w->box_count.as<uint32_t>();
            }
            if (w->dynamic_contacts) {
                if (int rc = island_substep(w, gp, n_slots, (flags & BGE_TICK_BULLET_BASIS) != 0, sub.no_repose != 0, (flags & BGE_TICK_BROADPHASE) != 0)) return rc;
            }
            HIP_TRY(bge::launch_ground(w->stream, w->view, gp, (flags & BGE_TICK_BULLET_BASIS) != 0));
            p.no_repose = 1u;
            p.cinfo_in = w->cinfo.as<uint32_t>();
        }
        const size_t n_passes = w->flat.pass_tile_begin.size() - 1;
        // BGE_TICK_GATHER_ROOTS with a transform pass: the roots write the all-gather's send buffer themselves
        const bool fused_gather = (flags & BGE_TICK_GATHER_ROOTS) && xform;
        if (flags & BGE_TICK_GATHER_ROOTS) {
            if (!w->comm.ready()) return fail(BGE_ERR_STATE, "BGE_TICK_GATHER_ROOTS needs bge_world_comm_init");
            if (w->flat.root_slots.size() > w->comm.rows_per_rank()) {
                return fail(BGE_ERR_INVALID, "%zu roots but the communicator was sized for %llu rows per rank",
                            w->flat.root_slots.size(), (unsigned long long)w->comm.rows_per_rank());
            }
        }
        if (fused_gather) {
            float* send = nullptr;
            if (w->comm.begin_frame(w->stream, &send) != BGE_OK) return fail(BGE_ERR_HIP, "%s", w->comm.error());
            p.root_out = send;
        }
        const bool pair_begins = w->profiling == 2 || (w->profiling == 1 && t == 0); // (t == 0 never happens after graph chunks: their start event is already recorded)
        const bool pair_ends = w->profiling == 2 || (w->profiling == 1 && t + 1 == ticks);
        if (pair_begins) {
            if (w->prof_used + 2 > w->prof_events.size()) {
                if (int rc = fold_profile(w)) return rc;
            }
            HIP_TRY(hipEventRecord(w->prof_events[w->prof_used], w->stream));
        }
        for (size_t pass = 0; pass < n_passes; ++pass) {
            p.tile_begin = w->flat.pass_tile_begin[pass];
            const uint32_t n_tiles = w->flat.pass_tile_begin[pass + 1] - p.tile_begin;
            HIP_TRY(bge::launch_tick(w->stream, w->view, p, n_tiles, flags));
        }
        if (pair_ends) {
            HIP_TRY(hipEventRecord(w->prof_events[w->prof_used + 1], w->stream));
            w->prof_used += 2;
            w->prof_ticks_pending.push_back(w->profiling == 2 ? 1u : ticks);
        }
        if (flags & BGE_TICK_BROADPHASE) {
            // buffers are sized on first use: a world that never asks for pairs does not pay for them
            const uint64_t cap = w->pair_capacity_req ? w->pair_capacity_req : std::max<uint64_t>(8 * w->flat.n_entities, 4096);
            int rc = BGE_OK;
            if (!(w->dynamic_contacts && w->bp_shared)) rc = w->broadphase.configure(std::max<uint64_t>(w->flat.n_slots, bge::kTile), cap);
            if (rc != BGE_OK) return fail(rc, "broadphase allocation failed: %s", w->broadphase.error());
            if (w->filter_table_stale && !w->filter_overflow) {
                std::vector<uint32_t> tab(256 * 4, 0u);
                for (size_t c = 0; c < w->filter_palette.size(); ++c) {
                    tab[4 * c] = w->filter_palette[c].group;
                    tab[4 * c + 1] = w->filter_palette[c].mask;
                    tab[4 * c + 2] = w->filter_palette[c].is_static;
                }
                HIP_TRY(hipMemcpyAsync(w->filter_table.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, w->stream));
                HIP_TRY(hipStreamSynchronize(w->stream)); // `tab` is a local
                w->filter_table_stale = false;
            }
            const bge::FilterPalette palette{w->filter_overflow ? nullptr : w->filter_class.as<uint32_t>(),
                                             w->filter_overflow ? nullptr : w->filter_table.as<uint4>(),
                                             static_cast<uint32_t>(w->filter_palette.size())};
            if (!(w->dynamic_contacts && w->bp_shared)) { // (else: island_substep ran it on these very boxes)
                rc = w->broadphase.run(w->stream, w->view, static_cast<uint64_t>(w->flat.n_tiles_ticked) * bge::kTile,
                                       w->entity_of_slot.as<uint32_t>(), nullptr, &palette, w->bp_partials.as<float4>());
                if (rc != BGE_OK) return fail(rc, "broadphase failed: %s", w->broadphase.error());
            }
            w->bp_shared = false;
            w->pairs_from_slab = false;
            if (with_triggers) {
                // counters: [0] overlaps found, [1] ghosts left to the all-bodies pass, [2] ghosts walked through the grid
                HIP_TRY(hipMemsetAsync(w->trig_count.p, 0, 12, w->stream));
                const uint32_t n_trig = static_cast<uint32_t>(w->triggers.size());
                const uint32_t* big_list = nullptr;
                if (n_trig > w->trigger_grid_min) {
                    // many ghosts: the ones that cover few cells look their bodies up in the broadphase's sorted grid
                    // (n_bodies x n_triggers box tests otherwise: 4 M bodies x 1000 ghosts = 4 G tests a tick)
                    HIP_TRY(w->trig_lists.ensure(static_cast<size_t>(n_trig) * 8));
                    const bge::TriggerView tv = w->trigger_view();
                    const bge::BoxQuery q{n_trig, tv.aabb, tv.group, tv.mask, tv.entity, w->trig_count.as<uint32_t>(),
                                          w->trig_lists.as<uint32_t>(), w->trig_lists.as<uint32_t>() + n_trig,
                                          static_cast<uint2*>(w->trig_pairs.p), kTriggerPairCap};
                    rc = w->broadphase.query_boxes(w->stream, w->view, w->entity_of_slot.as<uint32_t>(), &palette, q);
                    if (rc != BGE_OK) return fail(rc, "trigger query failed: %s", w->broadphase.error());
                    big_list = w->trig_lists.as<uint32_t>();
                }
                HIP_TRY(bge::launch_trigger_pairs(w->stream, static_cast<uint64_t>(w->flat.n_tiles_ticked) * bge::kTile, n_trig,
                                                  w->trigger_view(), w->view, w->entity_of_slot.as<uint32_t>(), w->trig_count.as<uint32_t>(),
                                                  w->trig_pairs.p, kTriggerPairCap, big_list,
                                                  big_list ? w->trig_count.as<uint32_t>() + 1 : nullptr));
                HIP_TRY(bge::launch_trigger_ghost_pairs(w->stream, n_trig, w->trigger_view(), w->trig_count.as<uint32_t>(), w->trig_pairs.p,
                                                        kTriggerPairCap));
                // The short way needs last tick's table to hold exactly the host's sets and no one-shot volume in the world (one that
                // fires leaves the world in the middle of ProcessTriggerEvents' loop and takes itself out of the later ghosts' lists:
                // that order dependence stays on the host)
                bool short_way = w->trig_device_diff && w->trig_mirror_valid;
                for (const bge_world::Trigger& t : w->triggers) short_way = short_way && !(t.one_shot && t.runtime_active);
                int how = 1;
                if (short_way) {
                    how = process_trigger_pairs_fast(w);
                    if (how < 0) return how;
                }
                if (how == 1) {
                    if (int rc2 = process_trigger_pairs(w)) return rc2;
                    ++w->trig_slow_ticks;
                    w->trig_mirror_valid = false;
                    bool one_shot = false;
                    for (const bge_world::Trigger& t : w->triggers) one_shot = one_shot || (t.one_shot && t.runtime_active);
                    if (w->trig_device_diff && !one_shot) {
                        if (int rc2 = rebuild_trigger_mirror(w)) return rc2;
                    }
                }
            }
        }
        w->maybe_dirty = phys && !xform;
        if (fused_gather) {
            if (w->comm.gather(w->stream, nullptr) != BGE_OK) return fail(BGE_ERR_HIP, "%s", w->comm.error());
        } else if (flags & BGE_TICK_GATHER_ROOTS) {
            if (int rc = bge_world_gather_roots(w, nullptr)) return rc;
        }
    }
    return BGE_OK;
}
} // namespace

// Bullet's btDiscreteDynamicsWorld::stepSimulation(timeStep, maxSubSteps, fixedTimeStep) around the world's ticks, as
// PhysicsSystem::StepSimulation calls it (src/physics/PhysicsSystem.cpp:855-863): the clock m_localTime accumulates
// timeStep; floor(m_localTime / fixedTimeStep) sub-steps are due and ALL of them are taken off the clock, but at most
// maxSubSteps are simulated; maxSubSteps == 0 is Bullet's variable-step mode (one step of timeStep, none when it is ~0).
// All arithmetic in binary32, as in the reference's single-precision Bullet build.
int bge_world_step_simulation(bge_world* w, double dt, int max_sub_steps, float fixed_step, const float gravity[3],
                              uint32_t flags, int* sub_steps)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    if (!(flags & BGE_TICK_PHYSICS)) return fail(BGE_ERR_INVALID, "bge_world_step_simulation needs BGE_TICK_PHYSICS");
    if (max_sub_steps < 0) return fail(BGE_ERR_INVALID, "max_sub_steps %d < 0", max_sub_steps);
    if (max_sub_steps > 0 && !(fixed_step > 0.0f)) return fail(BGE_ERR_INVALID, "fixed_step must be positive");
    const float time_step = static_cast<float>(dt); // stepSimulation(static_cast<btScalar>(dt), ...), PhysicsSystem.cpp:863
    int due = 0;
    float step = fixed_step;
    if (max_sub_steps > 0) {
        w->local_time = w->local_time + time_step;
        if (w->local_time >= fixed_step) {
            due = static_cast<int>(w->local_time / fixed_step);
            w->local_time = w->local_time - static_cast<float>(due) * fixed_step;
        }
    } else {
        // variable time step: m_localTime = m_latencyMotionStateInterpolation ? 0 : timeStep (not observable here);
        // btFuzzyZero(timeStep) ? 0 : 1 sub-step of timeStep
        step = time_step;
        due = std::fabs(time_step) < 1.1920928955078125e-07f ? 0 : 1;
        max_sub_steps = 1;
    }
    if (sub_steps) *sub_steps = due; // what stepSimulation returns (and LogStats prints as "substeps")
    const int run = std::min(due, max_sub_steps);
    const bool triggers = (flags & BGE_TICK_BROADPHASE) && !w->triggers.empty();
    SubStep sub{};
    if (triggers) {
        if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
        // EnsureTrigger / SyncTriggersToPhysics pose the ghosts from the Transforms as they are BEFORE stepSimulation
        DeviceGuard guard(w->device);
        ensure_triggers(w);
        if (w->triggers_device_stale) {
            if (int rc = sync_triggers_to_device(w)) return rc;
        }
        HIP_TRY(bge::launch_trigger_aabb(w->stream, static_cast<uint32_t>(w->triggers.size()), w->trigger_view(), w->view));
        for (bge_world::Trigger& tg : w->triggers) tg.posed = tg.posed || (tg.runtime_active && !tg.frozen);
        sub.ghosts_posed = true;
    }
    if (run == 0) {
        w->physics_updates += 1; // EnsureRigidBody runs in every PhysicsSystem::Update, sub-steps or not
        // no sub-step: no collision detection, no integration — but the calls around stepSimulation still run
        if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
        DeviceGuard guard(w->device);
        HIP_TRY(bge::launch_pose_only(w->stream, w->view, static_cast<uint64_t>(w->flat.n_tiles_ticked) * bge::kTile,
                                      (flags & BGE_TICK_BULLET_BASIS) != 0));
        w->maybe_dirty = true;
        if (triggers) {
            process_triggers_without_a_step(w);
        }
        const uint32_t rest = flags & (BGE_TICK_TRANSFORMS | BGE_TICK_NORMAL_MATRICES | BGE_TICK_GATHER_ROOTS);
        if (rest & BGE_TICK_TRANSFORMS) return tick_impl(w, 1, step, gravity, rest, sub);
        if (rest & BGE_TICK_GATHER_ROOTS) return bge_world_gather_roots(w, nullptr); // the collective is never skipped
        return BGE_OK;
    }
    // sub-steps before the last one: physics only (AABBs, pairs, world matrices and the gather are observable only after
    // the call); the teleport rule belongs to the first
    const uint32_t inner = flags & (BGE_TICK_PHYSICS | BGE_TICK_BULLET_BASIS);
    if (run >= 2) {
        if (int rc = tick_impl(w, 1, step, gravity, inner, sub)) return rc;
        sub.no_repose = true;
        if (run > 2) {
            if (int rc = tick_impl(w, static_cast<uint32_t>(run - 2), step, gravity, inner, sub)) return rc;
        }
    }
    return tick_impl(w, 1, step, gravity, flags, sub);
}
BGE_CATCH_ALL("bge_world_step_simulation")

int bge_world_reset_clock(bge_world* w)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    w->local_time = 0.0f;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_reset_clock")

int bge_world_set_ground_plane(bge_world* w, int enabled)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    DeviceGuard guard(w->device);
    const bool on = enabled != 0;
    const uint64_t manifold_bytes = std::max<uint64_t>(w->flat.n_slots, bge::kTile) * 128;
    if (on && w->has_topology && w->manifold.bytes < manifold_bytes) {
        // (first use; bge_world_set_topology keeps an existing store sized for the layout, so a smaller one cannot survive —
        //  checked by size all the same: k_ground loads and stores manifold[32 * slot ..] for every slot of the layout)
        HIP_TRY(hipStreamSynchronize(w->stream));
        HIP_TRY(w->manifold.ensure(manifold_bytes));
        HIP_TRY(hipMemsetAsync(w->manifold.p, 0, w->manifold.bytes, w->stream));
        w->rebuild_view();
    }
    w->ground_plane = on;
    w->drop_graph();
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_set_ground_plane")

int bge_world_set_static_contacts(bge_world* w, int enabled)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    DeviceGuard guard(w->device);
    const bool on = enabled != 0;
    const uint64_t bytes = std::max<uint64_t>(w->flat.n_slots, bge::kTile) * bge::kBoxManifolds * bge::kBoxManifoldWords * 4;
    if (on && w->has_topology && w->bmanifold.bytes < bytes) {
        HIP_TRY(hipStreamSynchronize(w->stream));
        HIP_TRY(w->bmanifold.ensure(bytes));
        HIP_TRY(hipMemsetAsync(w->bmanifold.p, 0xff, w->bmanifold.bytes, w->stream)); // every row free (bge::kBoxNone)
        w->rebuild_view();
    }
    w->static_contacts = on;
    w->static_contacts_ever = w->static_contacts_ever || on;
    w->obstacles_stale = true;
    w->drop_graph();
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_set_static_contacts")

int bge_world_set_dynamic_contacts(bge_world* w, int enabled)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    DeviceGuard guard(w->device);
    HIP_TRY(hipStreamSynchronize(w->stream));
    // (a body of an island collides its own pairs through contact_body, which keeps the obstacle manifold rows of its slot tidy
    //  whether or not there are obstacles: the store exists from here on, as after bge_world_set_static_contacts(1))
    const uint64_t bytes = std::max<uint64_t>(w->flat.n_slots, bge::kTile) * bge::kBoxManifolds * bge::kBoxManifoldWords * 4;
    if (enabled && w->has_topology && w->bmanifold.bytes < bytes) {
        HIP_TRY(w->bmanifold.ensure(bytes));
        HIP_TRY(hipMemsetAsync(w->bmanifold.p, 0xff, w->bmanifold.bytes, w->stream)); // every row free (bge::kBoxNone)
        w->rebuild_view();
    }
    if (const char* e = std::getenv("BGE_ISLAND_BIG_POINTS")) w->isl_big_points = static_cast<uint32_t>(std::strtoul(e, nullptr, 10));
    w->dynamic_contacts = enabled != 0;
    w->isl_n_prev = 0; // (off and on again: the pair cache starts empty)
    w->isl_gen_stale = true;
    w->drop_graph();
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_set_dynamic_contacts")

int bge_world_download_dynamic_pairs(bge_world* w, uint64_t cap, uint32_t* header3, float* points48, uint64_t* total)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    if (!total) return fail(BGE_ERR_INVALID, "total is NULL");
    DeviceGuard guard(w->device);
    HIP_TRY(hipStreamSynchronize(w->stream));
    const uint32_t n = w->dynamic_contacts ? w->isl_n_prev : 0u;
    *total = n;
    const uint64_t take = std::min<uint64_t>(n, cap);
    if (take == 0 || (!header3 && !points48)) return BGE_OK;
    const int at = w->isl_cur ^ 1; // (island_substep flipped the generations)
    std::vector<uint64_t> keys(take);
    std::vector<uint32_t> man(take * bge::kBoxManifoldWords);
    HIP_TRY(hipMemcpy(keys.data(), w->isl_keys[at].p, take * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(man.data(), w->isl_man[at].p, take * bge::kBoxManifoldWords * 4, hipMemcpyDeviceToHost));
    for (uint64_t k = 0; k < take; ++k) {
        const uint32_t* m = man.data() + k * bge::kBoxManifoldWords;
        if (header3) {
            header3[3 * k] = static_cast<uint32_t>(keys[k] >> 32);
            header3[3 * k + 1] = static_cast<uint32_t>(keys[k]);
            header3[3 * k + 2] = m[0];
        }
        if (points48) {
            float* o = points48 + 48 * k;
            for (uint32_t j = 0; j < 4; ++j) {
                for (uint32_t q = 0; q < 12; ++q) o[12 * j + q] = j < m[0] ? reinterpret_cast<const float*>(m + 4)[12 * j + q] : 0.0f;
            }
        }
    }
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_download_dynamic_pairs")

static int upload_restitution_impl(bge_world* w, uint64_t first, uint64_t count, const uint32_t* index, const float* restitution)
{
    if (count == 0) return BGE_OK;
    if (!restitution) return fail(BGE_ERR_INVALID, "restitution is NULL");
    DeviceGuard guard(w->device);
    const uint32_t* di = nullptr;
    if (int rc = stage_index(w, count, index, &di)) return rc;
    return upload_rows(w, first, count, 1, restitution, w->crestitution.p, 0, di);
}

int bge_world_upload_restitution(bge_world* w, uint64_t first, uint64_t count, const float* restitution)
try {
    if (int rc = check_range(w, first, count)) return rc;
    return upload_restitution_impl(w, first, count, nullptr, restitution);
}
BGE_CATCH_ALL("bge_world_upload_restitution")

int bge_world_upload_restitution_indexed(bge_world* w, uint64_t count, const uint32_t* entity_index, const float* restitution)
try {
    if (int rc = check_range(w, 0, 0)) return rc;
    if (count && !entity_index) return fail(BGE_ERR_INVALID, "entity_index is NULL");
    return upload_restitution_impl(w, 0, count, entity_index, restitution);
}
BGE_CATCH_ALL("bge_world_upload_restitution_indexed")

int bge_world_download_box_contacts(bge_world* w, uint64_t first, uint64_t count, uint8_t* n_manifolds, uint32_t* header8, float* points192)
try {
    if (int rc = check_range(w, first, count)) return rc;
    if (count == 0) return BGE_OK;
    DeviceGuard guard(w->device);
    std::vector<uint32_t> ci(count);
    if (int rc = download_rows(w, first, count, 1, w->cinfo.p, ci.data())) return rc;
    constexpr uint32_t kWords = bge::kBoxManifolds * bge::kBoxManifoldWords;
    std::vector<uint32_t> rows;
    if (w->bmanifold.p) {
        rows.resize(count * kWords);
        if (int rc = download_rows(w, first, count, kWords, w->bmanifold.p, rows.data())) return rc;
    }
    for (uint64_t i = 0; i < count; ++i) {
        // the device keeps a body's rows in no particular order: ascending entity of the other box here
        uint32_t order[bge::kBoxManifolds];
        uint32_t n = 0;
        if (w->bmanifold.p && (ci[i] & bge::kCiBoxes)) {
            for (uint32_t e = 0; e < bge::kBoxManifolds; ++e) {
                if (rows[i * kWords + e * bge::kBoxManifoldWords] != bge::kBoxNone) order[n++] = e;
            }
            std::sort(order, order + n, [&](uint32_t a, uint32_t b) {
                return rows[i * kWords + a * bge::kBoxManifoldWords] < rows[i * kWords + b * bge::kBoxManifoldWords];
            });
        }
        if (n_manifolds) n_manifolds[i] = static_cast<uint8_t>(n);
        for (uint32_t k = 0; k < bge::kBoxManifolds; ++k) {
            const uint32_t* src = k < n ? &rows[i * kWords + order[k] * bge::kBoxManifoldWords] : nullptr;
            if (header8) {
                header8[8 * i + 2 * k] = src ? src[0] : bge::kBoxNone;
                header8[8 * i + 2 * k + 1] = src ? src[1] : 0u;
            }
            if (points192) {
                float* dst = points192 + 192 * i + 48 * k;
                if (src) {
                    std::memcpy(dst, src + 4, 48 * 4);
                    for (uint32_t j = src[1]; j < 4; ++j) std::memset(dst + 12 * j, 0, 48); // (points beyond the count: whatever the row held)
                } else {
                    std::memset(dst, 0, 48 * 4);
                }
            }
        }
    }
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_download_box_contacts")

static int upload_friction_impl(bge_world* w, uint64_t first, uint64_t count, const uint32_t* index, const float* friction)
{
    if (count == 0) return BGE_OK;
    if (!friction) return fail(BGE_ERR_INVALID, "friction is NULL");
    DeviceGuard guard(w->device);
    const uint32_t* di = nullptr;
    if (int rc = stage_index(w, count, index, &di)) return rc;
    return upload_rows(w, first, count, 1, friction, w->cfriction.p, 0, di);
}

int bge_world_upload_friction(bge_world* w, uint64_t first, uint64_t count, const float* friction)
try {
    if (int rc = check_range(w, first, count)) return rc;
    return upload_friction_impl(w, first, count, nullptr, friction);
}
BGE_CATCH_ALL("bge_world_upload_friction")

int bge_world_upload_friction_indexed(bge_world* w, uint64_t count, const uint32_t* entity_index, const float* friction)
try {
    if (int rc = check_range(w, 0, 0)) return rc;
    if (count && !entity_index) return fail(BGE_ERR_INVALID, "entity_index is NULL");
    return upload_friction_impl(w, 0, count, entity_index, friction);
}
BGE_CATCH_ALL("bge_world_upload_friction_indexed")

int bge_world_download_contacts(bge_world* w, uint64_t first, uint64_t count, uint8_t* n_points, float* points32)
try {
    if (int rc = check_range(w, first, count)) return rc;
    if (count == 0) return BGE_OK;
    DeviceGuard guard(w->device);
    std::vector<uint32_t> ci(count);
    if (int rc = download_rows(w, first, count, 1, w->cinfo.p, ci.data())) return rc;
    if (n_points) {
        for (uint64_t i = 0; i < count; ++i) n_points[i] = w->manifold.p ? static_cast<uint8_t>((ci[i] >> bge::kCiCountShift) & 7u) : 0;
    }
    if (points32) {
        if (!w->manifold.p) {
            std::memset(points32, 0, count * 128);
        } else if (int rc = download_rows(w, first, count, 32, w->manifold.p, points32)) {
            return rc;
        }
    }
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_download_contacts")

int bge_world_tick(bge_world* w, float dt, const float gravity[3], uint32_t flags)
try {
    return bge_world_tick_many(w, 1, dt, gravity, flags);
}
BGE_CATCH_ALL("bge_world_tick")

int bge_world_profile_enable(bge_world* w, int enable)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    DeviceGuard guard(w->device);
    if (enable && w->prof_events.empty()) {
        w->prof_events.resize(2048);
        for (hipEvent_t& e : w->prof_events) {
            e = nullptr;
            HIP_TRY(hipEventCreate(&e));
        }
    }
    if (int rc = fold_profile(w)) return rc;
    w->prof_ms_carry = 0.0;
    w->prof_ticks_carry = 0;
    w->profiling = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_profile_enable")

int bge_world_profile_read(bge_world* w, double* tick_kernel_ms, uint64_t* ticks)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    DeviceGuard guard(w->device);
    if (int rc = fold_profile(w)) return rc;
    if (tick_kernel_ms) *tick_kernel_ms = w->prof_ms_carry;
    if (ticks) *ticks = w->prof_ticks_carry;
    w->prof_ms_carry = 0.0;
    w->prof_ticks_carry = 0;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_profile_read")

int bge_world_sync(bge_world* w)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    DeviceGuard guard(w->device);
    HIP_TRY(hipStreamSynchronize(w->stream));
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_sync")

int bge_world_download_world(bge_world* w, uint64_t first, uint64_t count, float* out16)
try {
    if (int rc = check_range(w, first, count)) return rc;
    if (!out16) return fail(BGE_ERR_INVALID, "out16 is NULL");
    if (count == 0) return BGE_OK;
    DeviceGuard guard(w->device);
    return download_rows(w, first, count, 16, w->world.p, out16);
}
BGE_CATCH_ALL("bge_world_download_world")

int bge_world_download_world_indexed(bge_world* w, uint64_t count, const uint32_t* entity_index, float* out16)
try {
    if (int rc = check_range(w, 0, 0)) return rc;
    if (count == 0) return BGE_OK;
    if (!entity_index || !out16) return fail(BGE_ERR_INVALID, "NULL argument");
    DeviceGuard guard(w->device);
    const uint32_t* di = nullptr;
    if (int rc = stage_index(w, count, entity_index, &di)) return rc;
    return download_rows(w, 0, count, 16, w->world.p, out16, di);
}
BGE_CATCH_ALL("bge_world_download_world_indexed")

int bge_world_download_pose_indexed(bge_world* w, uint64_t count, const uint32_t* entity_index, float* pos3, float* euler3)
try {
    if (int rc = check_range(w, 0, 0)) return rc;
    if (count == 0) return BGE_OK;
    if (!entity_index) return fail(BGE_ERR_INVALID, "entity_index is NULL");
    DeviceGuard guard(w->device);
    const uint32_t* di = nullptr;
    if (int rc = stage_index(w, count, entity_index, &di)) return rc;
    if (pos3) {
        if (int rc = download_rows(w, 0, count, 3, w->pos.p, pos3, di)) return rc;
    }
    if (euler3) {
        if (int rc = download_rows(w, 0, count, 3, w->euler.p, euler3, di)) return rc;
    }
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_download_pose_indexed")

int bge_world_download_normal(bge_world* w, uint64_t first, uint64_t count, float* out16)
try {
    if (int rc = check_range(w, first, count)) return rc;
    if (!out16) return fail(BGE_ERR_INVALID, "out16 is NULL");
    if (!w->normal.p) return fail(BGE_ERR_STATE, "no tick with BGE_TICK_NORMAL_MATRICES has run");
    if (count == 0) return BGE_OK;
    DeviceGuard guard(w->device);
    return download_rows(w, first, count, 16, w->normal.p, out16);
}
BGE_CATCH_ALL("bge_world_download_normal")

int bge_world_download_pose(bge_world* w, uint64_t first, uint64_t count, float* pos3, float* euler3)
try {
    if (int rc = check_range(w, first, count)) return rc;
    if (count == 0) return BGE_OK;
    DeviceGuard guard(w->device);
    if (pos3) {
        if (int rc = download_rows(w, first, count, 3, w->pos.p, pos3)) return rc;
    }
    if (euler3) {
        if (int rc = download_rows(w, first, count, 3, w->euler.p, euler3)) return rc;
    }
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_download_pose")

int bge_world_download_bodies(bge_world* w, uint64_t first, uint64_t count, float* linvel3, float* angvel3, float* quat4,
                              float* aabb6)
try {
    if (int rc = check_range(w, first, count)) return rc;
    if (count == 0) return BGE_OK;
    DeviceGuard guard(w->device);
    if (linvel3) {
        if (int rc = download_rows(w, first, count, 3, w->vel.p, linvel3)) return rc;
    }
    if (angvel3) {
        if (int rc = download_rows(w, first, count, 3, w->angvel.p, angvel3)) return rc;
    }
    if (quat4) {
        if (int rc = download_rows(w, first, count, 4, w->quat.p, quat4)) return rc;
    }
    if (aabb6) {
        if (int rc = download_rows(w, first, count, 6, w->aabb.p, aabb6)) return rc;
    }
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_download_bodies")

int bge_world_download_activation(bge_world* w, uint64_t first, uint64_t count, uint8_t* state, float* time)
try {
    if (int rc = check_range(w, first, count)) return rc;
    if (count == 0) return BGE_OK;
    DeviceGuard guard(w->device);
    std::vector<uint32_t> rec(count), fl(count);
    if (int rc = download_rows(w, first, count, 1, w->deact.p, rec.data())) return rc;
    if (int rc = download_rows(w, first, count, 1, w->flags.p, fl.data())) return rc;
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t type = fl[i] & bge::kTypeMask; // download_rows zero-fills entities without a slot
        const uint32_t r = (fl[i] & bge::kDrowsy) ? rec[i] : 0u;
        uint8_t s = BGE_ACTIVATION_NONE;
        float t = 0.0f;
        if (type == 1u) s = BGE_ISLAND_SLEEPING;           // addRigidBody puts static objects to sleep
        else if (type == 3u) s = BGE_DISABLE_DEACTIVATION; // PhysicsSystem.cpp:457
        else if (type == 2u) {
            if (r == bge::kDeactSleeping) s = BGE_ISLAND_SLEEPING;
            else if (r == bge::kDeactWants) s = BGE_WANTS_DEACTIVATION;
            else {
                s = BGE_ACTIVE_TAG;
                std::memcpy(&t, &r, 4);
            }
        }
        if (state) state[i] = s;
        if (time) time[i] = t;
    }
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_download_activation")

int bge_world_set_sleeping(bge_world* w, float linear_threshold, float angular_threshold, float seconds)
try {
    if (!w) return fail(BGE_ERR_INVALID, "NULL argument");
    if (!(linear_threshold >= 0.0f) || !(angular_threshold >= 0.0f) || !(seconds >= 0.0f))
        return fail(BGE_ERR_INVALID, "sleeping thresholds must be >= 0");
    w->sleep_lin = linear_threshold;
    w->sleep_ang = angular_threshold;
    w->sleep_time = seconds;
    w->drop_graph();
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_set_sleeping")

// Page-locked host memory: transfers to and from it run at the full PCIe rate (~55 GB/s against ~10 GB/s pageable).
int bge_host_alloc(uint64_t bytes, void** out)
try {
    if (!out) return fail(BGE_ERR_INVALID, "NULL argument");
    *out = nullptr;
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? BGE_ERR_OOM : BGE_ERR_HIP, "hipHostMalloc(%llu): %s", (unsigned long long)bytes,
                                     hipGetErrorString(e));
    *out = p;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_host_alloc")

int bge_host_free(void* p)
try {
    if (!p) return BGE_OK;
    const hipError_t e = hipHostFree(p);
    if (e != hipSuccess) return fail(BGE_ERR_HIP, "hipHostFree: %s", hipGetErrorString(e));
    return BGE_OK;
}
BGE_CATCH_ALL("bge_host_free")

int bge_world_download_dirty(bge_world* w, uint64_t first, uint64_t count, uint8_t* dirty)
try {
    if (int rc = check_range(w, first, count)) return rc;
    if (!dirty) return fail(BGE_ERR_INVALID, "dirty is NULL");
    if (count == 0) return BGE_OK;
    DeviceGuard guard(w->device);
    HIP_TRY(w->stage.ensure(count));
    HIP_TRY(bge::launch_dirty_bytes(w->stream, w->slot_of_entity.as<uint32_t>(), first, count, w->flags.as<uint32_t>(),
                                    w->stage.as<uint8_t>()));
    HIP_TRY(hipMemcpyAsync(dirty, w->stage.p, count, hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_download_dirty")

int bge_world_dirty_count(bge_world* w, uint64_t* out)
try {
    if (!w || !out) return fail(BGE_ERR_INVALID, "NULL argument");
    if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
    DeviceGuard guard(w->device);
    HIP_TRY(hipMemsetAsync(w->counter.p, 0, 8, w->stream));
    HIP_TRY(bge::launch_count_dirty(w->stream, w->flat.n_slots, w->flags.as<uint32_t>(), w->counter.as<unsigned long long>()));
    unsigned long long v = 0;
    HIP_TRY(hipMemcpyAsync(&v, w->counter.p, 8, hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    *out = v;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_dirty_count")

int bge_world_pairs(bge_world* w, uint32_t* pairs2, uint64_t cap, uint64_t* total)
try {
    if (!w || !total) return fail(BGE_ERR_INVALID, "NULL argument");
    if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
    DeviceGuard guard(w->device);
    bge::Broadphase& bp = w->pairs_from_slab ? w->slab_broadphase : w->broadphase;
    const int rc = bp.download(w->stream, pairs2, cap, total);
    if (rc != BGE_OK) return fail(rc, "pair download failed: %s", bp.error());
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_pairs")

// ---------------------------------------------------------------- sharded broadphase (bge_route.hip)
namespace {
uint64_t ticked_slots(const bge_world* w) { return static_cast<uint64_t>(w->flat.n_tiles_ticked) * bge::kTile; }

int refresh_global_of_slot(bge_world* w)
{
    if (!w->global_of_slot_stale && w->global_of_slot.p) return BGE_OK;
    const uint64_t S = std::max<uint64_t>(w->flat.n_slots, 1);
    std::vector<uint32_t> g(S, bge::kNone);
    for (uint64_t s = 0; s < w->flat.n_slots; ++s) {
        const uint32_t e = w->flat.entity_of_slot[s];
        if (e != bge::kNone) g[s] = e < w->global_id_host.size() ? w->global_id_host[e] : e;
    }
    HIP_TRY(w->global_of_slot.ensure(S * 4));
    HIP_TRY(hipMemcpyAsync(w->global_of_slot.p, g.data(), S * 4, hipMemcpyHostToDevice, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    w->global_of_slot_stale = false;
    return BGE_OK;
}
} // namespace

int bge_world_set_global_ids(bge_world* w, uint64_t first, uint64_t count, const uint32_t* ids)
try {
    if (int rc = check_range(w, first, count)) return rc;
    if (count && !ids) return fail(BGE_ERR_INVALID, "ids is NULL");
    if (w->global_id_host.size() < w->flat.n_entities) {
        const size_t old = w->global_id_host.size();
        w->global_id_host.resize(w->flat.n_entities);
        for (size_t i = old; i < w->global_id_host.size(); ++i) w->global_id_host[i] = static_cast<uint32_t>(i);
    }
    for (uint64_t i = 0; i < count; ++i) w->global_id_host[first + i] = ids[i];
    w->global_of_slot_stale = true;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_set_global_ids")

int bge_world_aabb_bounds(bge_world* w, float mn[3], float mx[3], uint64_t* n_bodies)
try {
    if (!w || !mn || !mx) return fail(BGE_ERR_INVALID, "NULL argument");
    if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
    DeviceGuard guard(w->device);
    const int rc = w->router.bounds(w->stream, w->view, ticked_slots(w), mn, mx, n_bodies);
    if (rc != BGE_OK) return fail(rc, "%s", w->router.error());
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_aabb_bounds")

int bge_world_axis_histogram(bge_world* w, uint32_t axis, float lo, float hi, uint32_t bins, uint64_t* hist)
try {
    if (!w || !hist) return fail(BGE_ERR_INVALID, "NULL argument");
    if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
    DeviceGuard guard(w->device);
    const int rc = w->router.histogram(w->stream, w->view, ticked_slots(w), axis, lo, hi, bins, hist);
    if (rc != BGE_OK) return fail(rc, "%s", w->router.error());
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_axis_histogram")

int bge_balanced_cuts(const uint64_t* hist, uint32_t bins, float lo, float hi, uint32_t nranks, float* cuts)
try {
    if (!hist || !cuts || bins == 0 || nranks == 0) return fail(BGE_ERR_INVALID, "bad argument");
    uint64_t total = 0;
    for (uint32_t b = 0; b < bins; ++b) total += hist[b];
    const double width = (static_cast<double>(hi) - static_cast<double>(lo)) / bins;
    cuts[0] = lo;
    cuts[nranks] = hi;
    uint64_t cum = 0;
    uint32_t b = 0;
    for (uint32_t k = 1; k < nranks; ++k) {
        // smallest bin whose inclusive prefix reaches k/nranks of the bodies; the cut is that bin's upper edge
        const uint64_t target = (total * k + nranks - 1) / nranks;
        while (b < bins && cum + hist[b] < target) cum += hist[b++];
        const uint32_t edge = b < bins ? b + 1 : bins;
        cuts[k] = (hi >= lo) ? static_cast<float>(static_cast<double>(lo) + width * edge) : lo;
        if (k > 1 && cuts[k] < cuts[k - 1]) cuts[k] = cuts[k - 1];
    }
    return BGE_OK;
}
BGE_CATCH_ALL("bge_balanced_cuts")

int bge_world_bp_route(bge_world* w, uint32_t axis, uint32_t nranks, const float* cuts, uint64_t* counts)
try {
    if (!w || !counts) return fail(BGE_ERR_INVALID, "NULL argument");
    if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
    if (nranks > 1 && !cuts) return fail(BGE_ERR_INVALID, "cuts is NULL");
    DeviceGuard guard(w->device);
    const float none[2] = {0.0f, 0.0f};
    const int rc = w->router.count(w->stream, w->view, ticked_slots(w), axis, nranks, cuts ? cuts : none, counts);
    if (rc != BGE_OK) return fail(rc, "%s", w->router.error());
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_bp_route")

int bge_world_bp_pack(bge_world* w, void* send_device)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
    DeviceGuard guard(w->device);
    if (int rc = refresh_global_of_slot(w)) return rc;
    const int rc = w->router.pack(w->stream, w->view, ticked_slots(w), w->global_of_slot.as<uint32_t>(), send_device);
    if (rc != BGE_OK) return fail(rc, "%s", w->router.error());
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_bp_pack")

int bge_world_bp_find(bge_world* w, const void* records_device, uint64_t n_records, uint32_t axis, float window_lo, float window_hi)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    if (n_records && !records_device) return fail(BGE_ERR_INVALID, "records_device is NULL");
    if (axis > 2) return fail(BGE_ERR_INVALID, "axis %u", axis);
    if (n_records > 0x7fff0000ull) return fail(BGE_ERR_INVALID, "%llu records exceed the 32-bit record index", (unsigned long long)n_records);
    DeviceGuard guard(w->device);
    bge::WorldView view{};
    const uint32_t* ids = nullptr;
    int rc = w->router.unpack(w->stream, records_device, n_records, &view, &ids);
    if (rc != BGE_OK) return fail(rc, "%s", w->router.error());
    // the number of records a slab receives changes from tick to tick: grow with 25 % headroom, never shrink
    uint64_t want_slots = std::max<uint64_t>(n_records, bge::kTile);
    if (want_slots > w->slab_broadphase.configured_slots()) want_slots += want_slots / 4;
    const uint64_t cap = std::max<uint64_t>(w->pair_capacity_req ? w->pair_capacity_req : std::max<uint64_t>(8 * want_slots, 4096),
                                            w->slab_broadphase.capacity());
    rc = w->slab_broadphase.configure(std::max<uint64_t>(want_slots, w->slab_broadphase.configured_slots()), cap);
    if (rc != BGE_OK) return fail(rc, "broadphase allocation failed: %s", w->slab_broadphase.error());
    const bge::PairWindow win{axis, window_lo, window_hi};
    rc = w->slab_broadphase.run(w->stream, view, n_records, ids, &win);
    if (rc != BGE_OK) return fail(rc, "broadphase failed: %s", w->slab_broadphase.error());
    w->pairs_from_slab = true;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_bp_find")

// The whole exchange over the world's RCCL communicator: common bounds (one all-reduce), uniform cuts along `axis`,
// counts (one all-gather), records (one grouped send/recv per peer), slab search.
int bge_world_bp_exchange(bge_world* w, uint32_t axis)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
    if (!w->comm.ready()) return fail(BGE_ERR_STATE, "bge_world_comm_init has not been called");
    if (axis > 2) return fail(BGE_ERR_INVALID, "axis %u", axis);
    const uint32_t N = static_cast<uint32_t>(w->comm.nranks());
    const uint32_t me = static_cast<uint32_t>(w->comm.rank());
    if (N > bge::kMaxSlabs) return fail(BGE_ERR_UNSUPPORTED, "at most %u ranks", bge::kMaxSlabs);
    DeviceGuard guard(w->device);
    float mn[3], mx[3];
    int rc = w->router.bounds(w->stream, w->view, ticked_slots(w), mn, mx, nullptr);
    if (rc != BGE_OK) return fail(rc, "%s", w->router.error());
    // max-reduce of (-min, max) along the axis
    HIP_TRY(w->bp_small.ensure(8 * static_cast<size_t>(N) * (N + 1) + 64));
    float ext[2] = {-mn[axis], mx[axis]};
    HIP_TRY(hipMemcpyAsync(w->bp_small.p, ext, sizeof ext, hipMemcpyHostToDevice, w->stream));
    if (w->comm.all_reduce_max(w->stream, w->bp_small.as<float>(), 2) != BGE_OK) return fail(BGE_ERR_HIP, "%s", w->comm.error());
    HIP_TRY(hipMemcpyAsync(ext, w->bp_small.p, sizeof ext, hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    // balanced cuts: all-reduced histogram of the min corners -> quantiles; every rank evaluates the same code on the
    // same reduced data, so the cuts are identical everywhere
    const float lo = -ext[0], hi = ext[1];
    std::vector<float> cuts(N + 1, 0.0f);
    if (hi >= lo) {
        std::vector<uint64_t> hist(bge::kMaxHistBins, 0);
        rc = w->router.histogram(w->stream, w->view, ticked_slots(w), axis, lo, hi, bge::kMaxHistBins, hist.data());
        if (rc != BGE_OK) return fail(rc, "%s", w->router.error());
        HIP_TRY(w->bp_hist.ensure(bge::kMaxHistBins * 8));
        HIP_TRY(hipMemcpyAsync(w->bp_hist.p, hist.data(), bge::kMaxHistBins * 8, hipMemcpyHostToDevice, w->stream));
        if (w->comm.all_reduce_sum_u64(w->stream, w->bp_hist.as<uint64_t>(), bge::kMaxHistBins) != BGE_OK)
            return fail(BGE_ERR_HIP, "%s", w->comm.error());
        HIP_TRY(hipMemcpyAsync(hist.data(), w->bp_hist.p, bge::kMaxHistBins * 8, hipMemcpyDeviceToHost, w->stream));
        HIP_TRY(hipStreamSynchronize(w->stream));
        if (int rc2 = bge_balanced_cuts(hist.data(), bge::kMaxHistBins, lo, hi, N, cuts.data())) return rc2;
    }
    std::vector<uint64_t> send_counts(N, 0), table(static_cast<size_t>(N) * N, 0), recv_counts(N, 0);
    rc = w->router.count(w->stream, w->view, ticked_slots(w), axis, N, cuts.data(), send_counts.data());
    if (rc != BGE_OK) return fail(rc, "%s", w->router.error());
    uint64_t* dev_counts = reinterpret_cast<uint64_t*>(w->bp_small.as<char>() + 64);
    HIP_TRY(hipMemcpyAsync(dev_counts, send_counts.data(), 8ull * N, hipMemcpyHostToDevice, w->stream));
    if (w->comm.all_gather_bytes(w->stream, dev_counts, dev_counts + N, 8ull * N) != BGE_OK) return fail(BGE_ERR_HIP, "%s", w->comm.error());
    HIP_TRY(hipMemcpyAsync(table.data(), dev_counts + N, 8ull * N * N, hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    uint64_t n_send = 0, n_recv = 0;
    for (uint32_t p = 0; p < N; ++p) {
        recv_counts[p] = table[static_cast<size_t>(p) * N + me]; // what rank p routes to my slab
        n_send += send_counts[p];
        n_recv += recv_counts[p];
    }
    HIP_TRY(w->bp_send.ensure(std::max<uint64_t>(n_send, 1) * bge::kRecordBytes));
    HIP_TRY(w->bp_recv.ensure(std::max<uint64_t>(n_recv, 1) * bge::kRecordBytes));
    if (int rc2 = refresh_global_of_slot(w)) return rc2;
    rc = w->router.pack(w->stream, w->view, ticked_slots(w), w->global_of_slot.as<uint32_t>(), w->bp_send.p);
    if (rc != BGE_OK) return fail(rc, "%s", w->router.error());
    if (w->comm.all_to_all_v(w->stream, w->bp_send.p, send_counts.data(), w->bp_recv.p, recv_counts.data(), bge::kRecordBytes) != BGE_OK)
        return fail(BGE_ERR_HIP, "%s", w->comm.error());
    const float wlo = me == 0 ? -INFINITY : cuts[me];
    const float whi = me + 1 == N ? INFINITY : cuts[me + 1];
    return bge_world_bp_find(w, w->bp_recv.p, n_recv, axis, wlo, whi);
}
BGE_CATCH_ALL("bge_world_bp_exchange")

int bge_world_upload_triggers(bge_world* w, uint64_t count, const uint32_t* entity_index, const uint8_t* shape, const float* size3,
                              const uint32_t* layer, const uint32_t* mask, const uint8_t* one_shot, const uint8_t* active)
try {
    if (int rc = check_range(w, 0, 0)) return rc;
    if (count && !entity_index) return fail(BGE_ERR_INVALID, "entity_index is NULL");
    std::unordered_map<uint32_t, size_t> old_index;
    for (size_t i = 0; i < w->triggers.size(); ++i) old_index[w->triggers[i].entity] = i;
    std::vector<bge_world::Trigger> next(count);
    std::unordered_map<uint32_t, bool> seen;
    for (uint64_t i = 0; i < count; ++i) {
        const uint32_t e = entity_index[i];
        if (e >= w->flat.n_entities) return fail(BGE_ERR_INVALID, "entity_index[%llu] = %u outside [0, %llu)", (unsigned long long)i, e,
                                                  (unsigned long long)w->flat.n_entities);
        if (seen.count(e)) return fail(BGE_ERR_INVALID, "entity %u carries two triggers", e);
        seen[e] = true;
        bge_world::Trigger& t = next[i];
        t.entity = e;
        t.shape = shape ? shape[i] : 0;
        if (size3) std::memcpy(t.size, size3 + 3 * i, 12);
        const uint32_t l = layer ? layer[i] : 0u;
        t.layer = l ? l : 4u; // kDefaultTriggerLayer = 1 << 2 (PhysicsSystem.cpp:38, 557)
        t.mask = mask ? mask[i] : 0xffffffffu;
        t.one_shot = one_shot && one_shot[i];
        t.component_active = !active || active[i];
        auto it = old_index.find(e);
        if (it != old_index.end()) {
            bge_world::Trigger& o = w->triggers[it->second];
            // a layer/mask change re-adds the ghost (PhysicsSystem.cpp:560-571): remembered overlaps are dropped
            if (o.layer == t.layer && o.mask == t.mask) {
                t.runtime_active = o.runtime_active;
                t.overlaps.swap(o.overlaps);
                t.overlap_bodies.swap(o.overlap_bodies);
                t.overlap_ghosts.swap(o.overlap_ghosts);
            }
            // (a ghost whose entity has no Transform keeps its place whatever is uploaded: EnsureTrigger does not reach it)
            t.posed = false;
            if (o.frozen && o.runtime_active) {
                t.frozen = true;
                t.runtime_active = true;
                std::memcpy(t.frozen_aabb, o.frozen_aabb, 24);
                t.layer = o.layer;
                t.mask = o.mask;
                if (t.overlaps.empty()) {
                    t.overlaps.swap(o.overlaps);
                    t.overlap_bodies.swap(o.overlap_bodies);
                    t.overlap_ghosts.swap(o.overlap_ghosts);
                }
            }
        }
    }
    w->triggers.swap(next);
    w->trig_index_of_entity.clear();
    for (size_t i = 0; i < w->triggers.size(); ++i) w->trig_index_of_entity[w->triggers[i].entity] = static_cast<uint32_t>(i);
    w->triggers_device_stale = true;
    w->trig_list_on_device = false;
    w->trig_mirror_valid = false; // (the keys carry trigger indices)
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_upload_triggers")

int bge_world_trigger_events(bge_world* w, bge_trigger_event* out, uint64_t cap, uint64_t* total)
try {
    if (!w || !total) return fail(BGE_ERR_INVALID, "NULL argument");
    *total = w->trigger_events.size();
    if (out) {
        const uint64_t take = std::min<uint64_t>(cap, w->trigger_events.size());
        if (take) std::memcpy(out, w->trigger_events.data(), take * sizeof(bge_trigger_event));
        w->trigger_events.clear();
    }
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_trigger_events")

int bge_world_set_trigger_stay_events(bge_world* w, int enabled)
try {
    if (!w) return fail(BGE_ERR_INVALID, "NULL world");
    w->trig_stay_events = enabled != 0;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_set_trigger_stay_events")

int bge_world_trigger_diff_stats(bge_world* w, uint64_t* device_ticks, uint64_t* host_ticks, uint64_t* stay_suppressed)
try {
    if (!w) return fail(BGE_ERR_INVALID, "NULL world");
    if (device_ticks) *device_ticks = w->trig_fast_ticks;
    if (host_ticks) *host_ticks = w->trig_slow_ticks;
    if (stay_suppressed) *stay_suppressed = w->trig_stay_suppressed;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_trigger_diff_stats")

int bge_world_trigger_active(bge_world* w, uint64_t count, const uint32_t* entity_index, uint8_t* active)
try {
    if (!w || (count && (!entity_index || !active))) return fail(BGE_ERR_INVALID, "NULL argument");
    std::unordered_map<uint32_t, bool> state;
    for (const bge_world::Trigger& t : w->triggers) state[t.entity] = t.component_active;
    for (uint64_t i = 0; i < count; ++i) {
        auto it = state.find(entity_index[i]);
        active[i] = it != state.end() && it->second ? 1 : 0;
    }
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_trigger_active")

int bge_world_trigger_query_stats(bge_world* w, uint32_t* through_grid, uint32_t* against_all_bodies)
try {
    if (!w) return fail(BGE_ERR_INVALID, "NULL world");
    if (through_grid) *through_grid = w->trig_through_grid;
    if (against_all_bodies) *against_all_bodies = w->trig_against_all;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_trigger_query_stats")

int bge_world_pack_roots(bge_world* w, void* dst_device)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
    DeviceGuard guard(w->device);
    float* dst = dst_device ? static_cast<float*>(dst_device) : w->root_worlds.as<float>();
    HIP_TRY(bge::launch_pack_roots(w->stream, w->flat.root_slots.size(), w->root_slots.as<uint32_t>(), w->world.as<float>(), dst));
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_pack_roots")

int bge_world_device_array(bge_world* w, int which, void** device_ptr, uint64_t* elements)
try {
    if (!w || !device_ptr) return fail(BGE_ERR_INVALID, "NULL argument");
    if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
    uint64_t n = 0;
    void* p = nullptr;
    switch (which) {
    case BGE_ARRAY_WORLD: p = w->world.p; n = w->flat.n_slots; break;
    case BGE_ARRAY_ROOT_WORLDS: p = w->root_worlds.p; n = w->flat.root_slots.size(); break;
    case BGE_ARRAY_SLOT_OF_ENTITY: p = w->slot_of_entity.p; n = w->flat.n_entities; break;
    case BGE_ARRAY_POSITION: p = w->pos.p; n = w->flat.n_slots; break;
    case BGE_ARRAY_PAIRS: {
        bge::Broadphase& bp = w->pairs_from_slab ? w->slab_broadphase : w->broadphase;
        DeviceGuard guard(w->device);
        if (bp.compact(w->stream) != BGE_OK) return fail(BGE_ERR_HIP, "%s", bp.error());
        p = bp.pairs_device();
        n = bp.capacity();
        break;
    }
    case BGE_ARRAY_NORMAL: p = w->normal.p; n = w->normal.p ? w->flat.n_slots : 0; break;
    default: return fail(BGE_ERR_INVALID, "unknown device array %d", which);
    }
    *device_ptr = p;
    if (elements) *elements = n;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_device_array")

int bge_comm_unique_id(void* out128)
try {
    if (!out128) return fail(BGE_ERR_INVALID, "out128 is NULL");
    std::string err;
    const int rc = bge::RootComm::unique_id(out128, err);
    if (rc != BGE_OK) return fail(rc, "%s", err.c_str());
    return BGE_OK;
}
BGE_CATCH_ALL("bge_comm_unique_id")

int bge_world_comm_init(bge_world* w, int nranks, int rank, const void* id128, uint64_t rows_per_rank)
try {
    if (!w || !id128) return fail(BGE_ERR_INVALID, "NULL argument");
    if (nranks <= 0 || rank < 0 || rank >= nranks) return fail(BGE_ERR_INVALID, "rank %d of %d", rank, nranks);
    DeviceGuard guard(w->device);
    const int rc = w->comm.init(nranks, rank, id128, rows_per_rank);
    if (rc != BGE_OK) return fail(rc, "%s", w->comm.error());
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_comm_init")

int bge_world_gather_roots(bge_world* w, void** table_device)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
    if (!w->comm.ready()) return fail(BGE_ERR_STATE, "bge_world_comm_init has not been called");
    if (w->flat.root_slots.size() > w->comm.rows_per_rank()) {
        return fail(BGE_ERR_INVALID, "%zu roots but the communicator was sized for %llu rows per rank", w->flat.root_slots.size(),
                    (unsigned long long)w->comm.rows_per_rank());
    }
    DeviceGuard guard(w->device);
    float* send = nullptr;
    int rc = w->comm.begin_frame(w->stream, &send);
    if (rc != BGE_OK) return fail(rc, "%s", w->comm.error());
    HIP_TRY(bge::launch_pack_roots(w->stream, w->flat.root_slots.size(), w->root_slots.as<uint32_t>(), w->world.as<float>(), send,
                                   /*compact=*/true));
    rc = w->comm.gather(w->stream, table_device);
    if (rc != BGE_OK) return fail(rc, "%s", w->comm.error());
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_gather_roots")

int bge_world_download_gathered(bge_world* w, float* out, uint64_t floats)
try {
    if (!w || !out) return fail(BGE_ERR_INVALID, "NULL argument");
    if (!w->comm.ready()) return fail(BGE_ERR_STATE, "bge_world_comm_init has not been called");
    DeviceGuard guard(w->device);
    if (w->comm.wait(w->stream) != BGE_OK) return fail(BGE_ERR_HIP, "%s", w->comm.error());
    const void* table = w->comm.last_table();
    if (!table) return fail(BGE_ERR_STATE, "nothing has been gathered yet");
    // the table holds compact rows (12 floats); the caller gets full 4x4 matrices
    const uint64_t rows = w->comm.rows_per_rank() * static_cast<uint64_t>(w->comm.nranks());
    std::vector<float> compact(rows * bge::RootComm::kRowFloats);
    HIP_TRY(hipMemcpyAsync(compact.data(), table, compact.size() * 4, hipMemcpyDeviceToHost, w->stream));
    HIP_TRY(hipStreamSynchronize(w->stream));
    const uint64_t take = std::min<uint64_t>(floats / 16, rows);
    for (uint64_t r = 0; r < take; ++r) {
        const float* c = &compact[r * bge::RootComm::kRowFloats];
        float* m = out + 16 * r;
        for (int row = 0; row < 4; ++row) {
            m[4 * row] = c[3 * row];
            m[4 * row + 1] = c[3 * row + 1];
            m[4 * row + 2] = c[3 * row + 2];
            m[4 * row + 3] = row == 3 ? 1.0f : 0.0f;
        }
    }
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_download_gathered")

int bge_world_comm_set_mode(bge_world* w, int mode)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    if (mode != BGE_GATHER_ALLGATHER && mode != BGE_GATHER_DIRECT) return fail(BGE_ERR_INVALID, "unknown gather mode %d", mode);
    DeviceGuard guard(w->device);
    if (w->comm.ready() && w->comm.wait(w->stream) != BGE_OK) return fail(BGE_ERR_HIP, "%s", w->comm.error());
    w->comm.set_mode(mode);
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_comm_set_mode")

int bge_world_comm_wait(bge_world* w)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    DeviceGuard guard(w->device);
    const int rc = w->comm.wait(w->stream);
    if (rc != BGE_OK) return fail(rc, "%s", w->comm.error());
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_comm_wait")

int bge_world_comm_destroy(bge_world* w)
try {
    if (!w) return fail(BGE_ERR_INVALID, "world is NULL");
    DeviceGuard guard(w->device);
    (void)hipStreamSynchronize(w->stream);
    w->comm.destroy();
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_comm_destroy")

static void fill_info(const bge::Flattened& f, bge_world_info* info)
{
    info->n_entities = f.n_entities;
    info->n_transforms = f.n_transforms;
    info->n_slots = f.n_slots;
    info->n_tiles = f.n_tiles_ticked;
    info->n_passes = f.pass_tile_begin.size() - 1;
    info->n_roots = f.root_slots.size();
    info->n_limbo = f.n_limbo;
    info->n_bodies = 0;
    info->max_depth = f.max_depth;
}

int bge_world_get_info(bge_world* w, bge_world_info* info)
try {
    if (!w || !info) return fail(BGE_ERR_INVALID, "NULL argument");
    if (!w->has_topology) return fail(BGE_ERR_STATE, "bge_world_set_topology has not been called");
    fill_info(w->flat, info);
    for (uint8_t t : w->body_type_host) info->n_bodies += t != BGE_BODY_NONE ? 1 : 0;
    return BGE_OK;
}
BGE_CATCH_ALL("bge_world_get_info")

int bge_flatten_topology(uint64_t n, const uint32_t* parent, const uint8_t* has_transform, uint32_t* slot_of_entity,
                         uint8_t* level_of_entity, uint32_t* pass_of_entity, bge_world_info* info)
try {
    if (n >= 0xfffffff0ull) return fail(BGE_ERR_INVALID, "too many entities");
    bge::Flattened f;
    try {
        bge::flatten_topology(n, parent, has_transform, f);
    } catch (const std::bad_alloc&) {
        return fail(BGE_ERR_OOM, "host allocation failed");
    }
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t s = f.slot_of_entity[i];
        if (slot_of_entity) slot_of_entity[i] = s;
        if (level_of_entity) level_of_entity[i] = s == bge::kNone ? 0 : static_cast<uint8_t>((f.flags[s] & bge::kLevelMask) >> bge::kLevelShift);
        if (pass_of_entity) pass_of_entity[i] = f.pass_of_entity[i];
    }
    if (info) fill_info(f, info);
    return BGE_OK;
}
BGE_CATCH_ALL("bge_flatten_topology")

int bge_partition_subtrees(uint64_t n, const uint32_t* parent, const uint8_t* has_transform, uint32_t nranks,
                           uint32_t* rank_of_entity, uint64_t* nodes_per_rank)
try {
    if (!rank_of_entity || nranks == 0) return fail(BGE_ERR_INVALID, "bad arguments");
    try {
        bge::partition_subtrees(n, parent, has_transform, nranks, rank_of_entity, nodes_per_rank);
    } catch (const std::bad_alloc&) {
        return fail(BGE_ERR_OOM, "host allocation failed");
    }
    return BGE_OK;
}
BGE_CATCH_ALL("bge_partition_subtrees")

} // extern "C"
