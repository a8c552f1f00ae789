/*
 * bge_world.h — C ABI of the MI355X-native ECS world tick (libbge_world.so).
 *
 * The reference (Skeletus/BangGameEngine) has no plugin/FFI layer: the per-frame world update is
 * three C++ call sites inside Application::Update (src/core/Application.cpp:256, :283-285).  This
 * header is the boundary a maintainer binds instead (INTEGRATION.md shows the C++ adapter that keeps
 * the reference's call shapes on top of it).  Each entry point cites the reference code it replaces.
 *
 * Model
 *   - A `bge_world` is a device-resident mirror of one reference `Scene` (src/ecs/Scene.h:97-105):
 *     Transform / RigidBody / Collider components as structure-of-arrays in HBM, plus the parent→child
 *     hierarchy flattened into 256-slot tiles ordered by depth.
 *   - Entities are addressed by a dense ENTITY INDEX in [0, n) chosen by the caller (the adapter maps
 *     the reference's sparse EntityId — src/ecs/Entity.h:4-5 — onto it).
 *   - Matrices are 16 floats, row-major, row-vector convention, translation in elements 12..14, exactly
 *     as `Transform::world` (src/ecs/Transform.h:18-19).
 *   - Not thread-safe: one host thread per world (the reference is single-threaded).
 *   - Every function returns BGE_OK (0) or a negative error code and never throws; the message of the
 *     last failure on the calling thread is available from bge_last_error().
 *   - All work is enqueued on the world's HIP stream (bge_world_desc.stream, or a private stream);
 *     download functions synchronise that stream before returning.
 */
#ifndef BGE_WORLD_H
#define BGE_WORLD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BGE_API __attribute__((visibility("default")))

#define BGE_NO_PARENT 0xffffffffu

enum bge_status {
    BGE_OK = 0,
    BGE_ERR_INVALID = -1,     /* bad argument */
    BGE_ERR_HIP = -2,         /* a HIP runtime call failed */
    BGE_ERR_OOM = -3,         /* host or device allocation failed */
    BGE_ERR_STATE = -4,       /* call order (e.g. upload before set_topology) */
    BGE_ERR_UNSUPPORTED = -5  /* feature not built */
};

/* RigidBodyType (src/ecs/PhysicsComponents.h:20-25) + "entity has no rigid body" */
enum bge_body_type { BGE_BODY_STATIC = 0, BGE_BODY_DYNAMIC = 1, BGE_BODY_KINEMATIC = 2, BGE_BODY_NONE = 255 };
/* Bullet's activation states (btCollisionObject.h), as the reference's bodies carry them:
 * Dynamic bodies are created ACTIVE_TAG, Kinematic ones DISABLE_DEACTIVATION (src/physics/PhysicsSystem.cpp:454-463). */
enum bge_activation {
    BGE_ACTIVATION_NONE = 0,
    BGE_ACTIVE_TAG = 1,
    BGE_ISLAND_SLEEPING = 2,
    BGE_WANTS_DEACTIVATION = 3,
    BGE_DISABLE_DEACTIVATION = 4
};
/* ColliderShape (src/ecs/PhysicsComponents.h:7-11) */
enum bge_shape { BGE_SHAPE_BOX = 0, BGE_SHAPE_CAPSULE = 1 };

/* What one bge_world_tick does. */
enum bge_tick_flags {
    BGE_TICK_PHYSICS = 1u,    /* rigid-body slice of PhysicsSystem::Update (src/physics/PhysicsSystem.cpp:1208-1328) */
    BGE_TICK_TRANSFORMS = 2u, /* TransformSystem::Update (src/ecs/TransformSystem.cpp:40-46) */
    BGE_TICK_BROADPHASE = 4u, /* AABB update + overlapping pairs (Bullet updateAabbs/calculateOverlappingPairs) */
    BGE_TICK_ALL = 3u,        /* Application::Update's physics + transform steps (src/core/Application.cpp:256,284) */
    BGE_TICK_GATHER_ROOTS = 8u, /* after the tick: bge_world_gather_roots (needs bge_world_comm_init) */
    BGE_TICK_NORMAL_MATRICES = 16u, /* with TRANSFORMS: also write transpose(inverse(world)) per entity, the normalMtx
                                      Renderer::BeginFrame computes on the CPU per mesh (src/render/Renderer.cpp:633-636) */
    BGE_TICK_AABBS = 32u,     /* with PHYSICS: update the AABBs (Bullet updateAabbs) without the local pair search —
                                 the sharded broadphase (bge_world_bp_*) searches them across ranks instead */
    BGE_TICK_BULLET_BASIS = 64u /* with PHYSICS: Bullet's own orientation scheme for every Dynamic body, spinning or not —
                                 each step its basis goes through btMatrix3x3::getRotation, btTransformUtil::integrateTransform's
                                 exponential map + safeNormalize and setRotation, and SyncRigidBodiesFromPhysics
                                 (src/physics/PhysicsSystem.cpp:925-950) rewrites Transform::rotationEuler from it.  Without the
                                 flag a body whose angular velocity is exactly zero keeps its orientation and euler triple bit
                                 for bit (within 1e-5 relative of this mode; 44 bytes per body and tick cheaper).
                                 Use one mode for the whole life of a world. */
};

enum bge_device_array {
    BGE_ARRAY_WORLD = 0,          /* float[n_slots][16], slot order */
    BGE_ARRAY_ROOT_WORLDS = 1,    /* float[n_roots][16], filled by bge_world_pack_roots */
    BGE_ARRAY_SLOT_OF_ENTITY = 2, /* uint32[n_entities], BGE_NO_PARENT where the entity has no Transform */
    BGE_ARRAY_POSITION = 3,       /* float[n_slots][3] */
    BGE_ARRAY_PAIRS = 4,          /* uint32[pair_capacity][2], entity indices */
    BGE_ARRAY_NORMAL = 5          /* float[n_slots][16] normal matrices, slot order (after a NORMAL_MATRICES tick) */
};

typedef struct bge_world bge_world;

typedef struct bge_world_desc {
    uint32_t struct_size; /* sizeof(bge_world_desc) */
    int32_t device;       /* HIP device ordinal, or -1 for the current device */
    void* stream;         /* hipStream_t to enqueue on, or NULL for a private non-blocking stream */
    uint64_t pair_capacity; /* max overlapping pairs kept per tick (0 = 8 x entities); emission is split over 64 shards */
} bge_world_desc;

typedef struct bge_world_info {
    uint64_t n_entities;
    uint64_t n_transforms; /* entities that own a Transform */
    uint64_t n_slots;      /* n_tiles * 256 */
    uint64_t n_tiles;      /* tiles that are ticked */
    uint64_t n_passes;     /* dependent launches per transform pass (1 unless a subtree exceeds a tile) */
    uint64_t n_roots;
    uint64_t n_limbo;      /* transforms inside a parent cycle: never updated, stay dirty */
    uint64_t n_bodies;
    uint64_t max_depth;
} bge_world_info;

BGE_API const char* bge_last_error(void);
BGE_API uint32_t bge_version(void);

/* Scene lifetime: `Application` owning a `Scene` + `PhysicsSystem` (src/core/Application.h:39-42). */
BGE_API int bge_world_create(const bge_world_desc* desc, bge_world** out);
BGE_API void bge_world_destroy(bge_world* world);

/*
 * Hierarchy: replaces Scene::m_parents / m_children and the root scan of
 * Scene::ForEachRootTransform (src/ecs/Scene.cpp:354-393, 523-533).
 *   parent[i]         entity index of i's parent, or BGE_NO_PARENT
 *   has_transform[i]  0/1, NULL = all 1.  An entity whose parent has no Transform is a root
 *                     (Scene.cpp:528); an entity without a Transform gets no slot.
 * Entities in a parent cycle are never reached by the reference's DFS; they are kept, never updated,
 * and stay dirty.  Calling this again re-flattens and keeps all component state of surviving indices.
 * All Transforms start dirty with TRS = (0, 0, 1) as a default-constructed Transform (Transform.h:14-16).
 * On a live world the call also does what the reference's scene edits do to what stays:
 *   - a changed parent ENTITY marks the child's subtree dirty, through Transform-less entities too (Scene::SetParent ->
 *     MarkHierarchyDirty, Scene.cpp:354-393, 535-550).  "Changed" is read off the two arrays: a child moved away and back
 *     between two calls (two SetParent calls in the reference, each marking it) shows nothing here — mark it with
 *     bge_world_mark_dirty, as an adapter that forwards Transform::dirty does anyway;
 *   - a clean Transform whose parent entity stays but lost its Transform becomes a root WITHOUT being marked, and keeps its
 *     world matrix until something marks it (Scene::RemoveTransform marks nobody; TransformSystem::Update recomputes a node
 *     only when it or an ancestor is dirty);
 *   - an entity that loses its Transform while its body is in the world (uploaded, and a physics tick has run since) keeps
 *     the body: it is stepped, collides and enters trigger volumes as before, is never re-posed or re-created, nothing is
 *     written back, and uploads of Transform or body data to it are ignored — except BGE_BODY_NONE, which removes it
 *     (PhysicsSystem::EnsureRigidBody returns before it looks at the runtime of an entity without a Transform,
 *     PhysicsSystem.cpp:389-393, and only the RigidBody component's removal takes the body out).  A caller that drops an
 *     entity altogether uploads BGE_BODY_NONE for it before the call;
 *   - a trigger volume whose entity loses its Transform stays in the world with the box it was last posed to (EnsureTrigger
 *     returns early as well) until the trigger is removed from the uploaded set or the Transform returns.
 */
BGE_API int bge_world_set_topology(bge_world* world, uint64_t n, const uint32_t* parent, const uint8_t* has_transform);

/*
 * Transform components: position / rotationEuler (radians) / scale as packed xyz triples
 * (src/ecs/Transform.h:14-16).  Any array may be NULL (left unchanged).  Marks the range dirty
 * (Transform::MarkDirty, src/ecs/Transform.cpp:13-16).
 */
BGE_API int bge_world_upload_trs(bge_world* world, uint64_t first, uint64_t count, const float* pos3,
                                 const float* euler3, const float* scale3);
BGE_API int bge_world_mark_dirty(bge_world* world, uint64_t first, uint64_t count);
/* Sparse form: row i belongs to entity entity_index[i] (what an adapter uploads after scanning for Transform::dirty). */
BGE_API int bge_world_upload_trs_indexed(bge_world* world, uint64_t count, const uint32_t* entity_index,
                                         const float* pos3, const float* euler3, const float* scale3);

/*
 * RigidBody + Collider components (src/ecs/PhysicsComponents.h:13-37) for a range of entities.
 *   type[i]   bge_body_type; BGE_BODY_NONE removes the body.  A body exists only where the entity also has
 *             a Transform (PhysicsSystem.cpp:389-393).
 *   mass      RigidBody::mass (Dynamic: max(mass, 0.01), else 0 — PhysicsSystem.cpp:426-429); NULL = 1
 *   shape     bge_shape, NULL = box;   size3: Collider::size, NULL = (.5,.5,.5)
 *   layer / mask   NULL = 1 / 0xffffffff; layer 0 is treated as 1 (PhysicsSystem.cpp:407)
 * Marks RigidBody.dirty and Collider.dirty: on the next physics tick the body is (re)created from its
 * Transform with zero velocity (PhysicsSystem.cpp:398-477).
 */
BGE_API int bge_world_upload_bodies(bge_world* world, uint64_t first, uint64_t count, const uint8_t* type,
                                    const float* mass, const uint8_t* shape, const float* size3,
                                    const uint32_t* layer, const uint32_t* mask);
BGE_API int bge_world_upload_bodies_indexed(bge_world* world, uint64_t count, const uint32_t* entity_index,
                                            const uint8_t* type, const float* mass, const uint8_t* shape,
                                            const float* size3, const uint32_t* layer, const uint32_t* mask);
/*
 * Extension (no reference API): overwrite linear / angular velocity without touching dirty flags.
 * The reference's bodies only ever gain velocity from gravity and contacts; synthetic workloads seed it.
 * An angular velocity then stays constant: Bullet's gyroscopic impulse (zero at rest, hence zero for every body the
 * reference can create without contacts) is not part of the path.
 */
BGE_API int bge_world_set_velocities(bge_world* world, uint64_t first, uint64_t count, const float* linvel3,
                                     const float* angvel3);

/*
 * One fixed-step tick = Application::Update's data-parallel part (src/core/Application.cpp:256, 284):
 *   BGE_TICK_PHYSICS     re-pose dirty bodies from their Transform (zeroing Dynamic velocities,
 *                        PhysicsSystem.cpp:952-989), one Bullet sub-step for free bodies
 *                        (v += g*dt; x += v*dt; exponential-map rotation), write position/rotationEuler back
 *                        and mark Dynamic transforms dirty (PhysicsSystem.cpp:916-950)
 *   BGE_TICK_TRANSFORMS  local = mtxSRT(S,R,T); world = parentWorld * local (root: world = local); clear dirty
 *   BGE_TICK_BROADPHASE  (with PHYSICS) feed AABBs and collect overlapping pairs
 * dt is seconds (the reference narrows its double once, PhysicsSystem.cpp:863); gravity is 3 floats
 * (the reference uses (0, config.gravity, 0), PhysicsSystem.cpp:130).
 */
BGE_API int bge_world_tick(bge_world* world, float dt, const float gravity[3], uint32_t flags);
/* Enqueue `ticks` identical ticks back to back (the catch-up loop of Application::Run, Application.cpp:96-101). */
BGE_API int bge_world_tick_many(bge_world* world, uint32_t ticks, float dt, const float gravity[3], uint32_t flags);
/*
 * PhysicsSystem::StepSimulation (src/physics/PhysicsSystem.cpp:848-875) = Bullet's
 *     m_world->stepSimulation(static_cast<btScalar>(dt), 4, max(config.fixedStep, 1/240))
 * around the world's ticks: the world keeps Bullet's clock m_localTime (binary32).  Every call adds dt to it;
 * n = int(m_localTime / fixed_step) sub-steps of fixed_step are due (0 while the clock is below fixed_step), all n are
 * taken off the clock, min(n, max_sub_steps) are simulated; *sub_steps (may be NULL) receives n, stepSimulation's return
 * value.  With dt == fixed_step — what Application::Run passes (src/core/Application.cpp:96-101, 326) — that is exactly
 * one sub-step per call and identical to bge_world_tick(world, fixed_step, gravity, flags).  max_sub_steps == 0 is Bullet's
 * variable-step mode: one step of dt.
 *   flags   as for bge_world_tick; BGE_TICK_PHYSICS is required.  The teleport rule (re-pose of dirty bodies) is applied
 *           once, before the first sub-step (SyncKinematicBodiesToPhysics runs before stepSimulation); AABBs / pairs /
 *           trigger events / world matrices / the root gather are produced once, from the state after the last sub-step
 *           (trigger ghosts are posed from the Transforms as they were before the first).
 *   n == 0  nothing is simulated, but what PhysicsSystem::Update does around the step still happens: dirty bodies are
 *           re-posed, every Dynamic body's Transform is marked dirty again (SyncRigidBodiesFromPhysics), and every
 *           remembered trigger overlap is reported as Stay (the ghosts' pair caches did not change).
 * bge_world_reset_clock zeroes m_localTime (a new btDiscreteDynamicsWorld, e.g. after OnSceneReloaded of a new world).
 */
BGE_API int bge_world_step_simulation(bge_world* world, double dt, int max_sub_steps, float fixed_step, const float gravity[3],
                                      uint32_t flags, int* sub_steps);
BGE_API int bge_world_reset_clock(bge_world* world);
BGE_API int bge_world_sync(bge_world* world);
/*
 * Kernel timing with HIP events on the world's stream (the reference times its step with chrono around
 * stepSimulation, src/physics/PhysicsSystem.cpp:862-866).
 *   enable = 1  one event pair around each bge_world_tick_many call (all its launches, gaps included) — no
 *               markers between back-to-back kernels, so it does not perturb what it measures
 *   enable = 2  one pair around every tick's tick-kernel launches (excludes broadphase / collective work that
 *               is interleaved on the stream; costs a few microseconds per tick)
 *   enable = 0  off
 * bge_world_profile_read synchronises the stream and returns the summed time and the number of ticks covered
 * since the last read.
 */
BGE_API int bge_world_profile_enable(bge_world* world, int enable);
BGE_API int bge_world_profile_read(bge_world* world, double* tick_kernel_ms, uint64_t* ticks);

/* Page-locked host buffers for the upload / download entry points: copies to and from them run at the PCIe rate
 * (pageable memory: roughly a fifth of it).  Optional — every entry point accepts any host pointer. */
BGE_API int bge_host_alloc(uint64_t bytes, void** out);
BGE_API int bge_host_free(void* p);

/* Results.  `Transform::world` after TransformSystem::Update; position/rotationEuler after PhysicsSystem::Update. */
BGE_API int bge_world_download_world(bge_world* world, uint64_t first, uint64_t count, float* out16);
BGE_API int bge_world_download_pose(bge_world* world, uint64_t first, uint64_t count, float* pos3, float* euler3);
/* normalMtx per entity of the last BGE_TICK_NORMAL_MATRICES tick (16 floats, bx::mtxTranspose(bx::mtxInverse(world))). */
BGE_API int bge_world_download_normal(bge_world* world, uint64_t first, uint64_t count, float* out16);
BGE_API int bge_world_download_world_indexed(bge_world* world, uint64_t count, const uint32_t* entity_index, float* out16);
BGE_API int bge_world_download_pose_indexed(bge_world* world, uint64_t count, const uint32_t* entity_index, float* pos3,
                                            float* euler3);
BGE_API int bge_world_download_bodies(bge_world* world, uint64_t first, uint64_t count, float* linvel3,
                                      float* angvel3, float* quat4, float* aabb6);
BGE_API int bge_world_download_dirty(bge_world* world, uint64_t first, uint64_t count, uint8_t* dirty);
/* Deactivation ("sleeping") of free bodies, inside stepSimulation (src/physics/PhysicsSystem.cpp:863): a Dynamic body
 * whose |v| stays below 0.8 and |w| below 1.0 for more than 2 s goes WANTS_DEACTIVATION at the end of that step and
 * ISLAND_SLEEPING in the next one (a free body is an island of its own); asleep it takes no gravity, is not integrated,
 * and its velocities are zeroed every step.  Only body (re)creation wakes it (the reference never calls activate()).
 * state[i] receives a bge_activation; time[i] btCollisionObject::m_deactivationTime while the body is ACTIVE_TAG and 0
 * otherwise (Bullet keeps a stale value there that nothing reads).  Either pointer may be NULL. */
BGE_API int bge_world_download_activation(bge_world* world, uint64_t first, uint64_t count, uint8_t* state, float* time);
/* Thresholds of the above (Bullet's defaults 0.8, 1.0, 2.0 — the reference never changes them).
 * seconds == 0 disables sleeping (Bullet: gDeactivationTime == 0). */
BGE_API int bge_world_set_sleeping(bge_world* world, float linear_threshold, float angular_threshold, float seconds);
/*
 * The static ground plane of the reference's physics world, with Bullet's contact handling for it (SURVEY.md section 8(f)
 * rank 4).  PhysicsSystem::EnsureGround (src/physics/PhysicsSystem.cpp:149-166) adds btStaticPlaneShape((0,1,0), 0) with
 * friction 1 and restitution 0 to every world, in group StaticFilter (2) with mask AllFilter; stepSimulation (:863) then runs
 * btConvexPlaneCollisionAlgorithm and btSequentialImpulseConstraintSolver for every Dynamic body at the plane:
 *   per sub-step ONE contact at the collider's support vertex, kept in a 4-point persistent manifold; contact + friction
 *   rows with warm starting, split impulse below -0.04, 10 iterations; the implicit gyroscopic impulse of a spinning body.
 * A body dropped on the plane comes to rest on it and falls asleep (bge_world_download_activation).  Bodies still do not
 * collide with EACH OTHER on this path (no convex-convex narrowphase): a body is an island of its own.
 *   bge_world_set_ground_plane   0 (default) = free bodies, what BASELINE's workloads are; 1 = the reference's world.
 *                                A body whose mask lacks bit 1 (value 2) passes through, as in Bullet.
 *   bge_world_upload_friction    RigidBody::friction (default 0.5, src/ecs/PhysicsComponents.h:32); the plane's is 1.
 *   bge_world_download_contacts  n_points[i] in 0..4 and, per body, 4 x (localA.xyz, appliedImpulse, localB.x, distance,
 *                                localB.z, appliedImpulseLateral1; localB.y is exactly 0) — inspection (the reference exposes no contacts).
 */
BGE_API int bge_world_set_ground_plane(bge_world* world, int enabled);
BGE_API int bge_world_upload_friction(bge_world* world, uint64_t first, uint64_t count, const float* friction);
BGE_API int bge_world_upload_friction_indexed(bge_world* world, uint64_t count, const uint32_t* entity_index, const float* friction);
BGE_API int bge_world_download_contacts(bge_world* world, uint64_t first, uint64_t count, uint8_t* n_points, float* points32);

/*
 * Dynamic boxes on the Static / Kinematic BOX colliders of the scene (round 3).  The reference
 * creates every RigidBody — Static ones too — as a Bullet body with its btBoxShape / btCapsuleShape
 * (src/physics/PhysicsSystem.cpp:421-474, CreateShape :686-707) and its default dispatcher collides Dynamic bodies with them
 * (:122-128, 863): a body dropped over assets/scenes/demo.json:67-91's "Ground", a 50 x 1 x 50 box, rests on the box's top.
 *   bge_world_set_static_contacts  0 (default) = off, as BASELINE's free bodies need; 1 = every Dynamic body with a BOX collider
 *       collides with every Static / Kinematic body with a BOX collider whose fed AABB overlaps its own and whose layer / mask
 *       pass both ways: Bullet's btBoxBoxDetector (15-axis separating-axis test, face clipping, the four-point cull) into a
 *       4-point persistent manifold per pair, all of a body's manifolds (the plane's too) in ONE sequential-impulse island per
 *       body — a static body merges no islands.  Combined friction = product of both frictions clamped to +-10, combined
 *       restitution = product of both restitutions (btManifoldResult), so RigidBody::restitution (:438) is live here.
 *       Not built: capsules against boxes (GJK / EPA); Dynamic against Dynamic is bge_world_set_dynamic_contacts.  A body holds at most 4 such manifolds (lowest
 *       entity indices); the Dynamic body is always the pair's body A — Bullet orders a pair by proxy creation, which the
 *       reference leaves to an unordered_map's iteration order (oracle/boxbox_ref.h states every such choice).
 *   bge_world_upload_restitution   RigidBody::restitution per entity (default 0, src/ecs/PhysicsComponents.h:33).
 *   bge_world_download_box_contacts  per queried entity: n_manifolds[i] in 0..4, header8 = 4 x (other entity index or
 *       0xffffffff, points 0..4) in ascending entity index, points192 = 4 manifolds x 4 points x (localA.xyz, localB.xyz,
 *       normalWorldOnB.xyz, distance, appliedImpulse, appliedImpulseLateral1).  Any of the three may be NULL.
 */
BGE_API int bge_world_set_static_contacts(bge_world* world, int enabled);
BGE_API int bge_world_upload_restitution(bge_world* world, uint64_t first, uint64_t count, const float* restitution);
BGE_API int bge_world_upload_restitution_indexed(bge_world* world, uint64_t count, const uint32_t* entity_index, const float* restitution);
BGE_API int bge_world_download_box_contacts(bge_world* world, uint64_t first, uint64_t count, uint8_t* n_manifolds, uint32_t* header8,
                                            float* points192);
/*
 * Dynamic boxes against EACH OTHER (round 3, after SURVEY.md section 8(f) rank 4).  In the reference every RigidBody is a
 * btRigidBody of one btDiscreteDynamicsWorld (src/physics/PhysicsSystem.cpp:122-131, 421-474): two Dynamic boxes collide, rest on
 * each other, and bodies whose AABBs overlap form ONE simulation island that is solved together and sleeps / wakes together.
 *   bge_world_set_dynamic_contacts  0 (default) = off; 1 = every pair of Dynamic bodies with BOX colliders whose fed AABBs overlap
 *       and whose layer / mask pass both ways is a pair of the cache (body A = the lower entity index): btBoxBoxDetector into a 4-point
 *       persistent manifold per pair; btSimulationIslandManager's rule — the bodies of every pair are united, touching or not; an
 *       island sleeps only when none of its bodies is ACTIVE_TAG, otherwise its sleeping bodies turn WANTS_DEACTIVATION (timer 0,
 *       no gravity until the stepSimulation call ends) — and btSequentialImpulseConstraintSolver over all manifolds of an island
 *       (the bodies in ascending entity index, each body's plane manifold, its manifolds with Static / Kinematic boxes, its pairs
 *       with Dynamic boxes of higher index: oracle/island_ref.h states why this order and not Bullet's pool order).  One device
 *       thread solves a small island, a workgroup a big one (rows that share no body side by side, level by level: the sequence's
 *       results exactly); the sub-step reads two counters back (pairs, island bodies).  Capsules take no part (GJK / EPA).
 *       Works with or without the plane and the static contacts.  The sub-step's pair search (all bodies, Static ones too) keeps
 *       pair_capacity pairs when bge_world_create was given one — more is BGE_ERR_INVALID, never a silent drop — and otherwise starts at
 *       8 per entity and doubles by itself whenever a sub-step finds more.
 *       Islands live inside ONE world: a scene sharded over several worlds (bge_partition_subtrees) does not collide bodies of
 *       different shards with each other.
 *   bge_world_download_dynamic_pairs  the pair cache after the last tick in ascending (lower, higher) entity index: *total pairs;
 *       for the first `cap`: header3 = lower index, higher index, points; points48 = 4 x (localA.xyz, localB.xyz, normalWorldOnB.xyz,
 *       distance, appliedImpulse, appliedImpulseLateral1).  header3 / points48 may be NULL.
 */
BGE_API int bge_world_set_dynamic_contacts(bge_world* world, int enabled);
BGE_API int bge_world_download_dynamic_pairs(bge_world* world, uint64_t cap, uint32_t* header3, float* points48, uint64_t* total);
/* Scene::CountDirtyTransforms (src/ecs/Scene.cpp:435-446): a device-side wave-reduced count. */
BGE_API int bge_world_dirty_count(bge_world* world, uint64_t* out);
/* Overlapping pairs of the last BROADPHASE tick as (a, b) entity indices, a < b, unordered list.
 * *total receives the number found on the device (at most cap are copied).  Fails with BGE_ERR_INVALID when the
 * device had to drop pairs (pair_capacity too small); pairs2 = NULL just queries the count. */
BGE_API int bge_world_pairs(bge_world* world, uint32_t* pairs2, uint64_t cap, uint64_t* total);

/*
 * Trigger volumes: TriggerVolume components (src/ecs/PhysicsComponents.h:39-48) and the
 * Enter / Stay / Exit events of PhysicsSystem::ProcessTriggerEvents (src/physics/PhysicsSystem.cpp:1017-1074).
 *   bge_world_upload_triggers replaces the world's whole trigger set (count may be 0).  Per trigger: the entity it
 *       sits on (needs a Transform), shape/size as for colliders, layer (0 is treated as 4, kDefaultTriggerLayer), mask,
 *       oneShot, active.  Triggers that were present before keep their remembered overlaps unless layer/mask changed.
 *   Every BGE_TICK_BROADPHASE tick then (a) poses each active trigger's ghost box from its entity's Transform as it is
 *       BEFORE the step (EnsureTrigger, PhysicsSystem.cpp:575), (b) tests it on the device against the fed AABB of every
 *       registered collision object — every rigid body, Static ones included, and every other active trigger ghost (each
 *       of two overlapping ghosts then reports the other) — with the filter (groupT & maskO) && (groupO & maskT): the
 *       content of Bullet's pair cache for the ghost (btGhostPairCallback, PhysicsSystem.cpp:132-133; the reference gives
 *       Bullet custom groups, :473,577, so no static-static exclusion applies; the ghost's own entity is skipped, :1033),
 *       (c) diffs the overlap set with the previous tick's — on the device, see bge_world_set_trigger_stay_events below; the host
 *       keeps the sets and takes over whenever it changed them itself —: Enter (0) / Stay (1) / Exit (2).  A one-shot trigger
 *       turns inactive after its first non-empty set, forgets it, and leaves the world at once: triggers processed AFTER
 *       it in the same tick no longer list it (:1062-1072).  The triggers are processed IN THE ORDER OF THE UPLOADED ARRAY
 *       (the reference walks a std::unordered_map — an order the language leaves open; the C++ adapter and the oracle use
 *       ascending EntityId).  With triggers present such a tick synchronises the stream (the events are for host code).
 *   bge_world_trigger_events returns (and clears) the events accumulated since the previous call.
 *   bge_world_trigger_active reports TriggerVolume::active per queried entity (0 after a one-shot fired, or no trigger).
 *   With more than 64 triggers (BGE_TRIGGER_GRID_MIN overrides the number) step (b) changes: a ghost whose box covers at
 *       most 8192 cells of the broadphase grid looks up the bodies sorted into those cells (plus the bodies too wide for the
 *       grid) instead of being tested against every body; wider ghosts keep the all-bodies test.  The overlap sets are the
 *       same either way.  bge_world_trigger_query_stats reports how the last tick split the ghosts that are in the world.
 */
typedef struct bge_trigger_event {
    uint32_t type;    /* 0 Enter, 1 Stay, 2 Exit (PhysicsSystem::TriggerEvent::Type, src/physics/PhysicsSystem.h:50-62) */
    uint32_t trigger; /* entity index of the trigger */
    uint32_t other;   /* entity index of the body or of the other trigger */
} bge_trigger_event;
BGE_API int bge_world_upload_triggers(bge_world* world, uint64_t count, const uint32_t* entity_index, const uint8_t* shape,
                                      const float* size3, const uint32_t* layer, const uint32_t* mask,
                                      const uint8_t* one_shot, const uint8_t* active);
BGE_API int bge_world_trigger_events(bge_world* world, bge_trigger_event* out, uint64_t cap, uint64_t* total);
/* Stay events.  PhysicsSystem::ProcessTriggerEvents (src/physics/PhysicsSystem.cpp:1017-1074) reports Stay for every remembered
 * overlap on every tick, and so does bge_world_trigger_events by default.  With many volumes that list is the cost of the trigger
 * pass (92,000 records a tick at 4 M bodies x 1000 volumes): the Enter / Exit difference itself is taken on the device (a table of
 * last tick's overlaps stays there; only the changes travel).  enabled = 0 leaves the Stay records out — Enter and Exit still come,
 * in the same order — and bge_world_trigger_diff_stats counts what was left out.
 * device_ticks / host_ticks: how many ticks took their difference on the device / on the host (the first tick after the volumes or
 * their activation changed, ticks with a one-shot volume in the world, a tick with more than 65,536 changes; BGE_TRIGGER_DEVICE_DIFF=0
 * in the environment keeps every tick on the host). */
BGE_API int bge_world_set_trigger_stay_events(bge_world* world, int enabled);
BGE_API int bge_world_trigger_diff_stats(bge_world* world, uint64_t* device_ticks, uint64_t* host_ticks, uint64_t* stay_suppressed);

BGE_API int bge_world_trigger_active(bge_world* world, uint64_t count, const uint32_t* entity_index, uint8_t* active);
BGE_API int bge_world_trigger_query_stats(bge_world* world, uint32_t* through_grid, uint32_t* against_all_bodies);

/* Multi-GPU support: compact the world matrices of all roots (entity order) into one buffer that the
 * caller all-gathers across ranks (one collective per frame).  dst = NULL packs into the world's own
 * BGE_ARRAY_ROOT_WORLDS buffer; otherwise dst is a device pointer with room for n_roots*16 floats. */
BGE_API int bge_world_pack_roots(bge_world* world, void* dst_device);
BGE_API int bge_world_device_array(bge_world* world, int which, void** device_ptr, uint64_t* elements);

/*
 * Native collective (RCCL over xGMI), one process per GPU.  The reference has no distributed layer; this is
 * the only exchange of the sharded tick (BASELINE.json configs[4]): every rank contributes the world matrices
 * of its roots and receives everybody's.
 *   bge_comm_unique_id   rank 0 fills a 128-byte id and distributes it by any means (the bench uses
 *                        torch.distributed's store); wraps ncclGetUniqueId.
 *   bge_world_comm_init  joins the communicator (ncclCommInitRank) and allocates a ring of 8 send/receive buffer
 *                        pairs of `rows_per_rank` rows (ranks pad to the largest root count) plus a side stream for
 *                        the collective.  A row on the wire is BGE_ROOT_ROW_FLOATS = 12 floats: the root's world matrix
 *                        without its fourth column, which is exactly (0, 0, 0, 1) for a root (world = local =
 *                        bx::mtxSRT, src/ecs/Transform.cpp:32-35) — elements 0,1,2, 4,5,6, 8,9,10, 12,13,14.
 *   bge_world_gather_roots   packs this rank's roots on the world's stream and enqueues ONE gather on the side
 *                        stream; buffers rotate per frame, so frame t's gather runs under the following ticks.
 *                        *table_device (may be NULL) receives the device pointer of the table being filled:
 *                        nranks x rows_per_rank x 12 floats, rank-major.  bge_world_download_gathered returns full
 *                        4x4 matrices.
 *   bge_world_comm_wait  makes the world's stream wait for every outstanding gather.
 * librccl.so.1 is loaded at run time (dlopen); BGE_ERR_UNSUPPORTED when it cannot be found.
 */
#define BGE_ROOT_ROW_FLOATS 12
BGE_API int bge_comm_unique_id(void* out128);
BGE_API int bge_world_comm_init(bge_world* world, int nranks, int rank, const void* id128, uint64_t rows_per_rank);
BGE_API int bge_world_gather_roots(bge_world* world, void** table_device);
BGE_API int bge_world_comm_wait(bge_world* world);
/* How the frame's root table travels: one ncclAllGather (default), or DIRECT — one ncclSend + ncclRecv per peer in one
 * group, every rank pushing its block to each peer over its own xGMI link (the node is a full mesh).  Same table either
 * way; call on every rank.  bench.py times both on the node it runs on and keeps the faster one. */
enum bge_gather_mode { BGE_GATHER_ALLGATHER = 0, BGE_GATHER_DIRECT = 1 };
BGE_API int bge_world_comm_set_mode(bge_world* world, int mode);
/* Copy the most recently gathered table to the host (waits for it): nranks x rows_per_rank x 16 floats. */
BGE_API int bge_world_download_gathered(bge_world* world, float* out, uint64_t floats);
BGE_API int bge_world_comm_destroy(bge_world* world);

/*
 * Sharded broadphase: the GLOBAL pair set of a scene whose bodies live on several GPUs.
 * The reference has one Bullet world, hence one pair cache over all bodies (src/physics/PhysicsSystem.cpp:124, 863);
 * bge_partition_subtrees shards by subtree, so a per-world BGE_TICK_BROADPHASE only sees the pairs inside one shard.
 * For the pair search the bodies are re-partitioned into slabs along one axis, one slab per rank
 * (slab s = [cuts[s], cuts[s+1]), cuts[0] = -inf, cuts[nranks] = +inf, interior cuts non-decreasing):
 *   bge_world_set_global_ids   id reported for each local entity index (default: the index itself)
 *   bge_world_aabb_bounds      min / max corner over the body AABBs of the last BGE_TICK_AABBS / BGE_TICK_BROADPHASE
 *                              tick, so that ranks can agree on cuts
 *   bge_world_bp_route         counts[d] = records this world sends to slab d: one per slab the body's extent touches
 *   bge_world_bp_pack          the 48-byte records grouped by slab (slab d at record offset sum(counts[0..d))) into
 *                              device memory the caller exchanges (all-to-all: RCCL, torch.distributed, hipMemcpyPeer)
 *   bge_world_bp_find          pair search over the records a rank received; a pair is kept only where the lower end
 *                              of its overlap interval along `axis`, max(min_a, min_b), lies in [window_lo, window_hi)
 *                              = the rank's own slab, so the union over ranks is the global set without duplicates.
 *                              Afterwards bge_world_pairs returns these pairs (global ids) until the next
 *                              BGE_TICK_BROADPHASE tick.
 *   bge_world_bp_exchange      all of the above over the world's RCCL communicator (bge_world_comm_init): all-reduced
 *                              extent and histogram -> balanced cuts, one all-gather of counts, one send/recv per peer.
 * Record layout (float4 x 3): min.xyz | global id,  max.xyz | 0,  group | mask | static flag | 0.
 */
#define BGE_BP_RECORD_BYTES 48
BGE_API int bge_world_set_global_ids(bge_world* world, uint64_t first, uint64_t count, const uint32_t* ids);
BGE_API int bge_world_aabb_bounds(bge_world* world, float min3[3], float max3[3], uint64_t* n_bodies);
/* hist[b] = bodies whose AABB min corner along `axis` falls into bin b of `bins` (<= 4096) equal bins over [lo, hi]
 * (out-of-range values land in the first / last bin).  Summed over ranks it yields balanced cuts: */
BGE_API int bge_world_axis_histogram(bge_world* world, uint32_t axis, float lo, float hi, uint32_t bins, uint64_t* hist);
/* Host only: cuts[0..nranks] with cuts[k] = upper edge of the first bin at which k/nranks of the bodies are reached. */
BGE_API int bge_balanced_cuts(const uint64_t* hist, uint32_t bins, float lo, float hi, uint32_t nranks, float* cuts);
BGE_API int bge_world_bp_route(bge_world* world, uint32_t axis, uint32_t nranks, const float* cuts, uint64_t* counts);
BGE_API int bge_world_bp_pack(bge_world* world, void* send_device);
BGE_API int bge_world_bp_find(bge_world* world, const void* records_device, uint64_t n_records, uint32_t axis,
                              float window_lo, float window_hi);
BGE_API int bge_world_bp_exchange(bge_world* world, uint32_t axis);
BGE_API int bge_world_get_info(bge_world* world, bge_world_info* info);

/*
 * Host-side helpers (no GPU needed).
 * bge_flatten_topology: the tile/pass layout bge_world_set_topology would build — for inspection and tests.
 *   slot_of_entity[n], level_of_entity[n] (depth inside its tile), pass_of_entity[n]; any may be NULL.
 * bge_partition_subtrees: assign every entity to one of `nranks` shards, whole subtrees only, balancing
 *   node counts greedily (largest subtree first); rank_of_entity[n].  Entities in cycles go to rank 0.
 */
BGE_API int bge_flatten_topology(uint64_t n, const uint32_t* parent, const uint8_t* has_transform,
                                 uint32_t* slot_of_entity, uint8_t* level_of_entity, uint32_t* pass_of_entity,
                                 bge_world_info* info);
BGE_API int bge_partition_subtrees(uint64_t n, const uint32_t* parent, const uint8_t* has_transform,
                                   uint32_t nranks, uint32_t* rank_of_entity, uint64_t* nodes_per_rank);

#ifdef __cplusplus
}
#endif
#endif /* BGE_WORLD_H */
