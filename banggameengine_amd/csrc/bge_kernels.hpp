// bge_kernels.hpp — device data layout and kernel launch entry points (see bge_kernels.hip).
#pragma once

#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>

#include <cstdint>

namespace bge {

// Structure-of-arrays image of the Scene's components in HBM.  Every array is indexed by SLOT
// (bge_flatten.hpp); 256 consecutive slots form the tile one workgroup ticks.
struct WorldView {
    uint32_t* flags;          // [slots] body type, dirty bits, level, ...
    const uint32_t* parent;   // [slots] in-tile index / global slot of the parent
    const uint32_t* tile_hdr; // [tiles]
    // Transform (src/ecs/Transform.h:14-16)
    float* pos;               // [slots][3]  Transform::position == rigid body origin
    float* euler;             // [slots][3]  Transform::rotationEuler
    float* scale;             // [slots][3]
    // world matrices (Transform::world); `local` is never materialised
    float* world;             // [slots][16]
    // rigid body state (Bullet's btRigidBody) and parameters
    float* vel;               // [slots][3]
    float* angvel;            // [slots][3]
    float* quat;              // [slots][4]
    float* inv_mass;          // [slots]     only read for mass class 63 (more than 62 distinct masses)
    const float2* mass_palette; // [128]     (inv_mass, 1/inv_mass) per mass class (host bookkeeping; the kernel reads grav_palette)
    const float4* grav_palette; // [256]     (gravity / inv_mass, inv_mass) per mass class for the gravity vector of the tick
    uint32_t* deact;          // [slots]     deactivation record (bge_flatten.hpp), only touched while a body is slow or asleep
    float* half_extent;       // [slots][3]  AABB half extents of the collider in its own frame
    uint32_t* group;          // [slots]     collision filter group (layer)
    uint32_t* mask;           // [slots]
    uint32_t* filter_class;   // [slots]     index of the body's (group, mask, static) triple in the world's filter palette
    float* aabb;              // [slots][6]  min xyz, max xyz fed to the broadphase
    float* normal;            // [slots][16] transpose(inverse(world)); allocated on first use
    const uint32_t* root_index; // [slots]   position of a root in the root table (read by roots only, when packing)
    // ground contact (bge_contact.hip): the collider as Bullet holds it and the body's manifold with the plane y = 0
    float4* cshape;           // [slots]     box: half extents with margin; capsule: (radius, half height, radius); w unused
    float* cmass;             // [slots]     the body's mass as handed to btRigidBody (inertia = f(mass), not f(1 / inv_mass))
    float* cfriction;         // [slots]     RigidBody::friction
    uint32_t* cinfo;          // [slots]     kCi* bits below
    float* manifold;          // [slots][32] four contact points x (localA.xyz, appliedImpulse, localB.x, distance, localB.z, appliedImpulseLateral1);
                              //             allocated when the ground plane is switched on
    // contacts with the Static / Kinematic box colliders of the scene (round 3; bge_contact.hip, allocated when they are switched on)
    float* crestitution;      // [slots]     RigidBody::restitution (live on box contacts only: the plane's is 0, and the two multiply)
    uint32_t* bmanifold;      // [slots][kBoxManifolds][kBoxManifoldWords] a Dynamic box's manifolds with boxes, in no particular order:
                              //             words 0..3 = other entity, points, the other body's generation, 0; then four points x 12 floats
                              //             (localA.xyz, localB.xyz, normalWorldOnB.xyz, distance, appliedImpulse, appliedImpulseLateral1)
    uint32_t* frozen;         // [slots / 32] bit per slot, or null: a root whose parent entity lost its Transform keeps the world matrix it had
                              //             (parent * local) until something marks it dirty — TransformSystem::Update recomputes a node only when
                              //             it or an ancestor is dirty, and Scene::RemoveTransform marks nobody (tiles with kHdrFrozen look here)
};

// per-slot contact word (the flag word has no bit left)
constexpr uint32_t kCiCapsule = 1u;      // collider is a capsule (else a box)
constexpr uint32_t kCiGroundMask = 2u;   // the body's mask contains the ground's group (btBroadphaseProxy::StaticFilter = 2)
constexpr uint32_t kCiSolved = 4u;       // k_ground ran the solver for this body in this sub-step: velocities are final, gravity included
constexpr uint32_t kCiMoved = 8u;        // ... and the split impulse corrected the pose (rotationEuler must be rewritten)
constexpr uint32_t kCiCountShift = 4;    // bits 4..6: contact points in the manifold (0..4)
constexpr uint32_t kCiBoxes = 0x80u;     // the body holds at least one manifold with a box (its bmanifold rows are live)
constexpr uint32_t kCiIsland = 0x100u;   // Dynamic boxes collide with each other: this sub-step the body is in a simulation island of several
                                         // bodies that stays awake — the island kernels collide and solve it (k_ground_select leaves it alone),
                                         // k_tick does not put it to sleep on WANTS_DEACTIVATION; consumed by k_tick like kCiSolved
constexpr uint32_t kCiNoGravity = 0x200u; // Dynamic-against-Dynamic contacts on: the body was asleep when this stepSimulation call applied
                                         // gravity (k_island_begin of its first sub-step): no gravity for it until the call ends, whatever
                                         // wakes it in between

// A Dynamic box keeps at most this many manifolds with Static / Kinematic boxes (the lowest entity ids; Bullet has no limit —
// oracle/boxbox_ref.h kMaxBoxManifolds, a stated specification choice)
constexpr uint32_t kBoxManifolds = 4;
constexpr uint32_t kBoxManifoldWords = 52;
constexpr uint32_t kBoxNone = 0xffffffffu; // `other` of a free manifold row

// What a Dynamic box can rest on: one record per Static / Kinematic body with a box collider, ascending entity index, rebuilt by
// k_obstacles at the head of every sub-step (pose as Bullet holds it after SyncKinematicBodiesToPhysics, fed AABB, material)
struct ObstacleRec {
    float origin[3];
    float half[3];     // btBoxShape::getHalfExtentsWithMargin()
    float basis[9];
    float aabb[6];     // min xyz, max xyz: shape AABB at the pose + 0.02 (updateSingleAabb)
    float friction, restitution, breaking;
    uint32_t entity, group, mask, generation;
    uint32_t live;     // 0: the slot no longer carries such a body
    uint32_t pad[3];
};
static_assert(sizeof(ObstacleRec) == 128, "one record = two float4 x 4 lines");

struct GroundParams {
    float dt;
    float gx, gy, gz;
    uint64_t n_slots;
    uint32_t want_aabb; // the tick feeds AABBs (broadphase / triggers): solved bodies' boxes are written here, before the solve
    uint32_t repose;    // first sub-step of a PhysicsSystem::Update: dirty bodies are re-posed from their Transforms first
    uint32_t* list;            // [kGroundShards][shard_cap] slots k_ground_select hands to the solver
    uint32_t* list_count;      // [16 * shard] entries of a shard's segment, [16 * shard + 1] the shard's ticket counter in k_ground; all
                               // zero between sub-steps (the last workgroup of each shard sees to it)
    uint64_t shard_cap;        // ground_shard_cap(n_slots)
    uint32_t plane;            // the static plane y = 0 is in the world (bge_world_set_ground_plane)
    // Static / Kinematic box colliders (bge_world_set_static_contacts): null / 0 when off
    const uint32_t* obstacle_slots; // [n_obstacles] ascending entity index
    const uint32_t* obstacle_gen;   // [n_obstacles] generation of each body (bumped by every re-creation)
    ObstacleRec* obstacles;         // [n_obstacles] written by k_obstacles
    uint32_t n_obstacles;
    const uint32_t* entity_of_slot;
    // A grid over the obstacles' fed AABBs in x and z (k_obstacle_grid, rebuilt every sub-step behind k_obstacles when there are more
    // than kObstacleGridMin of them): the candidate test of a body walks the few cells its own box covers instead of all obstacles.
    //   words 0..7: [0] 1 = the grid is valid (0: walk all obstacles), [1] cells per axis, [2] wide obstacles (they cover more than 64
    //               cells and are always tested), [4..7] floats min x, min z, cells per unit in x, in z;  8..39: the wide ones;
    //   kObstacleGridStart..: start of each cell's items (cells + 1 words);  kObstacleGridItems..: obstacle numbers, cell by cell
    uint32_t* obstacle_grid;   // null when there are few obstacles
    uint32_t obstacle_grid_cap; // item capacity
    uint32_t* box_list;        // [n_slots] slots k_ground_select hands to k_contact_boxes
    uint32_t* box_count;       // [0] entries of box_list, [1] workgroups of k_contact_boxes that are done (both zero between sub-steps)
    uint32_t obstacles_ready;  // the island phases of this sub-step ran k_obstacles (launch_obstacles) already
};
// Dynamic boxes against each other (round 3, bge_island.hip): the pair cache of Dynamic boxes with a persistent manifold per
// pair, simulation islands by union-find over the pairs, one solver thread per island.  All arrays are device memory of the world.
struct IslandParams {
    float dt, gx, gy, gz;
    uint64_t n_slots;
    uint32_t repose;                // first sub-step of a stepSimulation call
    const uint32_t* entity_of_slot;
    const uint32_t* slot_of_entity;
    const uint32_t* gen_of_entity;  // how often the entity's body was (re)created
    uint32_t* counts;               // [0] pairs of Dynamic boxes found, [1] bodies in islands, [2] unused, [3] error bits,
                                    // [4] big islands listed, [5] their ticket, [6] ints handed out, [7] mid islands listed
    // pairs: key = lower entity << 32 | higher entity
    const uint2* bp_stage;          // the broadphase's pair list (slot, slot) in its shard slices (Broadphase::slices)
    const unsigned long long* bp_counts;
    uint64_t bp_shard_cap;
    uint32_t bp_shards;
    uint32_t bp_ids_are_entities;   // the list holds entity indices (the tick's own broadphase instance, shared) instead of slots
    uint64_t* keys_raw;             // [pair_cap] unsorted
    const uint64_t* keys;           // [n_pairs] ascending
    uint32_t pair_cap, n_pairs;
    uint32_t* man;                  // [n_pairs][kBoxManifoldWords]: points, generation of A, of B, 0; then 4 x 12 floats as in bmanifold
    const uint64_t* prev_keys;      // last sub-step's pairs and manifolds
    const uint32_t* prev_man;
    uint32_t n_prev;
    // islands
    uint32_t* parent;               // [n_slots] union-find over slots
    uint32_t* member;               // [n_slots] 1: listed as an island body this sub-step
    uint32_t* active;               // [n_slots] by root: the island holds a body that is ACTIVE_TAG
    uint32_t* index_of_slot;        // [n_slots] position in the sorted body list
    uint64_t* body_keys_raw;        // [body_cap] root slot << 32 | entity
    uint32_t* body_slot_raw;
    const uint64_t* body_keys;      // sorted
    const uint32_t* body_slot;
    uint32_t body_cap, n_bodies;
    void* solver_bodies;            // [n_bodies] IslBody
    void* rows;                     // [row_cap] IslRow: the contact rows in [0, row_cap / 2), the friction rows behind them
    void* rows_cold;                // [row_cap / 2] IslRowCold: what only the split-impulse sweeps and the write-back need of a contact row
    uint32_t row_cap;
    uint32_t iterations;            // 10 (btContactSolverInfo::m_numIterations); BGE_ISLAND_ITERATIONS overrides it for MEASUREMENTS only
    // islands too big for one thread's LDS column: k_island_solve lists them, k_island_solve_big takes a workgroup to each
    uint32_t* big_list;             // [n_bodies][2] first body, end (counts[4] of them; counts[5] is the workgroups' ticket)
    uint32_t* mid_list;             // [n_bodies][2] likewise, islands of 5 .. 16 bodies (counts[7] of them): k_island_solve<.., true>
    uint32_t* row_count;            // [n_bodies] contact points body i brings into its island's row list (k_island_bodies; 0 off the path)
    uint32_t* row_first;            // [n_bodies] exclusive sum of row_count: the island's rows are rows row_first[first] .. of the two arrays
    void* scan_tmp;                 // hipcub's temporary storage for that sum (island_scan_bytes)
    uint64_t scan_tmp_bytes;
    uint32_t* pair_first;           // [n_bodies] first pair of the sorted list that body i owns (k_island_flags)
    uint32_t* body_words;           // [n_bodies][2] per body of a big island: rows before it / the level of its last row
    uint32_t* ints;                 // [int_cap] per-row level, rows in level order, level starts (counts[6] handed out)
    uint32_t int_cap;
    uint32_t big_points;            // islands with more contact points than this go to k_island_solve_big (128; BGE_ISLAND_BIG_POINTS for tests)
};
constexpr uint32_t kIslBodyBytes = 160, kIslRowBytes = 96, kIslRowColdBytes = 32;

constexpr uint32_t kGroundShards = 64;
constexpr uint32_t kObstacleGridMin = 64, kObstacleGridAxis = 64, kObstacleGridWide = 32, kObstacleGridStart = 40,
                   kObstacleGridItems = kObstacleGridStart + kObstacleGridAxis * kObstacleGridAxis + 1;
// slots the select workgroups b = shard, shard + 64, ... (256 slots each) can send at most
inline uint64_t ground_shard_cap(uint64_t n_slots) { return ((n_slots + 255) / 256 + kGroundShards - 1) / kGroundShards * 256; }

// Trigger volumes (ghost objects), indexed by trigger number
struct TriggerView {
    const uint32_t* slot;      // [triggers] slot of the trigger's entity (kNone: entity has no Transform)
    const uint32_t* entity;    // [triggers] entity index
    const float* half_extent;  // [triggers][3]
    const uint32_t* group;     // [triggers] layer (0 -> 4, kDefaultTriggerLayer)
    const uint32_t* mask;      // [triggers]
    const uint8_t* active;     // [triggers]
    float* aabb;               // [triggers][6]
};

struct TickParams {
    float dt;
    float gx, gy, gz;
    float sleep_lin, sleep_lin2, sleep_ang2, sleep_time; // Bullet's sleeping thresholds (linear also squared, angular squared) and gDeactivationTime; time 0 = never sleep
    uint32_t tile_begin;
    uint32_t nt_out; // non-temporal stores for world / normal matrices (working set larger than the Infinity Cache)
    float* root_out; // when non-null: roots also write their world matrix (compact, 12 floats) to root_out[root_index] (send buffer of the gather)
    float4* bp_partial; // when non-null (AABB variants): every wave also writes the bounds of its bodies' fed AABBs, their
                        // number and the widest one — 2 x float4 at bp_partial[2 * (tile * 4 + wave)]: (min.xyz, widest extent),
                        // (max.xyz, count bits) — so that the broadphase needs no pass of its own over the AABBs for the grid
    const uint32_t* cinfo_in; // ground plane on: k_ground ran before this kernel; bodies it solved carry kCiSolved (null: no ground)
    uint32_t no_repose; // this tick is the 2nd..nth sub-step of ONE stepSimulation call (bge_world_step_simulation): dirty flags
                        // do not re-pose bodies — SyncKinematicBodiesToPhysics ran once, before the first sub-step
};

// flags: bit0 physics, bit1 transforms, bit2 / bit5 aabb, bit4 normal matrices (bge_tick_flags)
hipError_t launch_tick(hipStream_t stream, const WorldView& w, const TickParams& p, uint32_t n_tiles, uint32_t flags);

// PhysicsSystem::Update around a stepSimulation call that runs ZERO sub-steps (accumulated time < fixedStep): re-pose dirty
// bodies, write Dynamic poses back, mark them dirty — no integration (bge_world_step_simulation)
hipError_t launch_pose_only(hipStream_t stream, const WorldView& w, uint64_t n_slots, bool bullet_basis);

// Ground plane y = 0: collide every Dynamic body with it and run Bullet's solver for the bodies in contact (bge_contact.hip);
// launched before launch_tick of the same sub-step.
// the phases of a sub-step with Dynamic-against-Dynamic contacts (the host reads counts[] back between them: the sorts need them)
hipError_t launch_obstacles(hipStream_t stream, const WorldView& w, const GroundParams& g);
hipError_t launch_island_begin(hipStream_t stream, const WorldView& w, const GroundParams& g, const IslandParams& ip, bool bullet_basis);
hipError_t launch_island_pair_keys(hipStream_t stream, const WorldView& w, const IslandParams& ip);
hipError_t island_sort_keys(hipStream_t stream, void* tmp, size_t& tmp_bytes, const uint64_t* in, uint64_t* out, uint32_t n);
hipError_t island_sort_pairs(hipStream_t stream, void* tmp, size_t& tmp_bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, uint32_t n);
hipError_t launch_island_build(hipStream_t stream, const WorldView& w, const IslandParams& ip, bool orphans);
size_t island_scan_bytes(uint32_t n);
hipError_t launch_island_solve(hipStream_t stream, const WorldView& w, const GroundParams& g, const IslandParams& ip, bool bullet_basis);
hipError_t launch_ground(hipStream_t stream, const WorldView& w, const GroundParams& g, bool bullet_basis);

hipError_t launch_scatter_rows(hipStream_t stream, const uint32_t* slot_of_entity, uint64_t first, uint64_t count,
                               uint32_t width, const void* stage, void* dst, uint32_t* flags, uint32_t or_bits,
                               const uint32_t* index = nullptr, uint32_t need_bits = 0);
hipError_t launch_gather_rows(hipStream_t stream, const uint32_t* slot_of_entity, uint64_t first, uint64_t count,
                              uint32_t width, const void* src, void* stage, const uint32_t* index = nullptr);
hipError_t launch_scatter_bodies(hipStream_t stream, const uint32_t* slot_of_entity, uint64_t first, uint64_t count,
                                 const uint32_t* type_bits, const float* inv_mass, const float* half_extent3,
                                 const uint32_t* group, const uint32_t* mask, const uint32_t* filter_class, const WorldView& w,
                                 const uint32_t* index = nullptr, const float* cdims3 = nullptr, const float* cmass = nullptr,
                                 const uint32_t* cbits = nullptr);
hipError_t launch_scatter_velocities(hipStream_t stream, const uint32_t* slot_of_entity, uint64_t first, uint64_t count,
                                     const float* lin, const float* ang, const WorldView& w);
hipError_t launch_init_slots(hipStream_t stream, uint64_t n_slots, const uint32_t* structural_flags, const WorldView& w);
hipError_t launch_count_dirty(hipStream_t stream, uint64_t n_slots, const uint32_t* flags, unsigned long long* out);
hipError_t launch_dirty_bytes(hipStream_t stream, const uint32_t* slot_of_entity, uint64_t first, uint64_t count,
                              const uint32_t* flags, uint8_t* out);
hipError_t launch_trigger_aabb(hipStream_t stream, uint32_t n_triggers, const TriggerView& t, const WorldView& w);
hipError_t launch_trigger_pairs(hipStream_t stream, uint64_t n_slots, uint32_t n_triggers, const TriggerView& t, const WorldView& w,
                                const uint32_t* entity_of_slot, uint32_t* count, void* out_pairs, uint32_t cap,
                                const uint32_t* list = nullptr, const uint32_t* list_count = nullptr);
// ghost against ghost (both directions): appends (i | kGhostHit, j), trigger indices, to the same hit list
constexpr uint32_t kGhostHit = 0x80000000u;
hipError_t launch_trigger_ghost_pairs(hipStream_t stream, uint32_t n_triggers, const TriggerView& t, uint32_t* count, void* out_pairs,
                                      uint32_t cap);

// ---- Enter / Exit of the trigger overlaps on the device.  The hit list of a tick is turned into keys
//   (trigger index << 33) | (kGhostHit ? 1 << 32 : 0) | (body entity or other trigger's index)
// and put into an open-addressing table; a key that is not in LAST tick's table is an Enter, a key of last tick's table that is
// not in this one an Exit.  Only those — as keys, Exit marked by bit 63 — and five counters travel to the host, which owns the
// overlap sets (bge_world.cpp apply_trigger_deltas) and rebuilds the mirror table whenever it changed them itself.
constexpr uint64_t kTrigKeyEmpty = ~0ull;
constexpr uint64_t kTrigKeyExit = 1ull << 63;
struct TriggerDiff {
    uint64_t* cur;        // [1 << log2_cap] filled with kTrigKeyEmpty before the launch
    uint64_t* prev;       // last tick's table
    uint32_t log2_cap;
    uint32_t* header;     // [0] deltas appended, [1] distinct keys this tick, [2] a table overflowed, [3] hits, [4] [5] query counters
    uint64_t* deltas;     // behind the header in one buffer, so that one copy fetches both
    uint32_t delta_cap;
    const uint2* pairs;
    const uint32_t* count; // the hit list's counters ([0] hits, [1] [2] how the ghosts were split)
    uint32_t pair_cap;
};
hipError_t launch_trigger_diff(hipStream_t stream, const TriggerDiff& d);
// inserts `n` keys into a table (mirror rebuild from the host's sets); header[2] reports an overflow
hipError_t launch_trigger_table_build(hipStream_t stream, uint64_t* table, uint32_t log2_cap, const uint64_t* keys, uint32_t n, uint32_t* header);

// compact: 12 floats per root (4x3, the constant fourth column dropped) instead of 16
hipError_t launch_pack_roots(hipStream_t stream, uint64_t n_roots, const uint32_t* root_slots, const float* world, float* dst,
                             bool compact = false);

} // namespace bge
