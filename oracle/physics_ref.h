// oracle/physics_ref.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the rigid-body slice of the reference's PhysicsSystem::Update
// for FREE bodies (no contacts, no constraints, no ground plane — the scope of SURVEY.md §8
// a-10..a-13 — plus Bullet's deactivation ("sleeping") of a free body, part of §8(f) rank 4).  Structure follows the reference's per-tick
// loops over a per-entity runtime hash map so that timing it is a fair "reference
// CPU path" (it omits Bullet's own broadphase-tree / island / solver bookkeeping, so
// it is FASTER than the real reference would be — a conservative baseline).
//
// Follows:
//   src/physics/PhysicsSystem.cpp:1222-1246  prune runtimes whose entity/components vanished
//   src/physics/PhysicsSystem.cpp:1248-1260  EnsureRigidBody for every RigidBody with a Collider
//   src/physics/PhysicsSystem.cpp:382-499    EnsureRigidBody: (re)create on collider/body dirty,
//                                            mass = max(mass,0.01) for Dynamic else 0,
//                                            layer 0 -> 1, new body => zero velocity, pose from Transform
//   src/physics/PhysicsSystem.cpp:686-707    CreateShape clamps
//   src/physics/PhysicsSystem.cpp:952-989    SyncKinematicBodiesToPhysics (teleport rule)
//   src/physics/PhysicsSystem.cpp:848-875    StepSimulation -> stepSimulation(dt, 4, fixedStep)
//   src/physics/PhysicsSystem.cpp:916-950    SyncRigidBodiesFromPhysics (Dynamic only; marks dirty)
// and, inside stepSimulation, Bullet's published per-step sequence for an active free body
// (BulletDynamics/Dynamics/btDiscreteDynamicsWorld.cpp, btRigidBody.cpp,
//  BulletDynamics/ConstraintSolver/btSequentialImpulseConstraintSolver.cpp):
//   applyGravity:            totalForce += m_gravity, m_gravity = g / invMass (btRigidBody::setGravity)
//   predictUnconstraintMotion: damping factor pow(1-0, dt) == 1; predicted pose = integrate(x, v, w, dt)
//   updateAabbs:             AABB(current pose) ∪ AABB(predicted pose), each grown by 0.02
//   solver writeback:        v += (totalForce * invMass) * dt          (externalForceImpulse)
//   integrateTransforms:     x += v * dt; orientation by exponential map
//   clearForces
// Deactivation (btCollisionObject activation states; btRigidBody::updateDeactivation / wantsSleeping,
// btDiscreteDynamicsWorld::updateActivationState, btSimulationIslandManager::buildIslands).  The reference
// creates Dynamic bodies ACTIVE_TAG and Kinematic ones DISABLE_DEACTIVATION (PhysicsSystem.cpp:454-463), never calls
// activate() and leaves the thresholds at Bullet's defaults (linear 0.8, angular 1.0, gDeactivationTime 2 s).
// A free body is an island of its own, so per sub-step:
//   applyGravity             only if isActive() (state != ISLAND_SLEEPING)
//   buildIslands             a lone WANTS_DEACTIVATION body is an all-sleeping island -> ISLAND_SLEEPING
//   solver, integrateTransforms   skipped for a sleeping island / inactive body
//   updateActivationState    not asleep: |v|^2 < 0.8^2 and |w|^2 < 1^2 ? time += dt : (time = 0, active);
//                            time > 2 -> WANTS_DEACTIVATION;  asleep: velocities are zeroed every step.
// Nothing in the free-body scope wakes a sleeping body except re-creation (RigidBody/Collider dirty); a teleport
// (SyncKinematicBodiesToPhysics) moves it but leaves it asleep, exactly as setWorldTransform does.
// With dt == fixedStep (src/core/Application.cpp:326, PhysicsSystem.h:77) that is exactly one
// sub-step per tick; the caller passes that dt.
// Bullet's accumulator around the sub-steps (btDiscreteDynamicsWorld::stepSimulation(timeStep, maxSubSteps = 4,
// fixedTimeStep = max(config.fixedStep, 1/240)), PhysicsSystem.cpp:855-863) is restated behind `accumulate`:
//   m_localTime += timeStep;  if (m_localTime >= fixedTimeStep) { n = int(m_localTime / fixedTimeStep);
//   m_localTime -= n * fixedTimeStep; }   min(n, maxSubSteps) x internalSingleStepSimulation(fixedTimeStep)
// (binary32; pinned against the compiled stepSimulation at VA 0x1401c31d0 of the reference's exe by
// oracle/tools/check_solver_rows.py: truncating conversion, (float)n * fixedTimeStep unfused, the UNclamped n is returned;
// applyGravity before and clearForces after the sub-steps make no difference to a
// free body: the force is re-derived per sub-step here).  With n == 0 nothing is simulated, no collision detection runs —
// the ghosts' pair caches keep last call's content, so ProcessTriggerEvents reports Stay for every remembered overlap —
// while the re-pose rule before and the write-back + MarkDirty after the step still run.
//
// Angular velocity: nothing in the reference sets one, so ω = 0 for every reachable body and the solver's gyroscopic
// impulse (BT_ENABLE_GYROSCOPIC_FORCE_IMPLICIT_BODY, Bullet's default flag) is exactly zero.  Seeding ω (bulk_set_velocity)
// is a test extension: ω then stays constant here, which is NOT what Bullet does for anisotropic inertia.
//
// Orientation state.  Bullet keeps a 3x3 basis and converts basis -> quaternion -> basis every
// step.  Three modes are implemented so the choice the GPU path makes can be quantified:
//   kOrientIdeal   (default, what the GPU path implements): state is a quaternion; a body whose
//                  angular velocity is exactly zero keeps its orientation and its Transform euler
//                  bit for bit after the tick in which it was (re)posed.
//   kOrientQuat    state is a quaternion, re-normalised and written back every tick.
//   kOrientBasis   Bullet's own scheme: 3x3 basis state, getRotation/setRotation every tick
//                  (the GPU path's BGE_TICK_BULLET_BASIS, bit-identical to this mode).
// tests/test_oracle_physics.py bounds the distance between the modes.
//
// PARITY STATUS: "parity unpinned" in the strict sense (Bullet absent; the reference holds no fixtures).  The arithmetic
// of every step below — pose conversions, integrateTransform, gravity force and impulse, shape margins, getAabb,
// updateSingleAabb — is checked against the reference's compiled code by oracle/tools/check_bullet_order.py; the
// deactivation rule was read from the same disassembly; the ORDER of the steps is checked there too (check_step_order);
// buildIslands' rule (an island without ACTIVE_TAG / DISABLE_DEACTIVATION bodies falls asleep) was compared as text; that a
// contact-free body is an island of its own is from Bullet's published code alone.
#pragma once

#include <algorithm>
#include <cstdint>
#include <map>
#include <unordered_map>
#include <vector>

#include "island_ref.h"
#include "bullet_math.h"
#include "contact_ref.h"
#include "ecs_ref.h"

namespace orc {

enum OrientMode : int { kOrientIdeal = 0, kOrientQuat = 1, kOrientBasis = 2 };
// btCollisionObject.h activation states
enum Activation : int { kActiveTag = 1, kIslandSleeping = 2, kWantsDeactivation = 3, kDisableDeactivation = 4 };

struct RefBodyRuntime {
    bool hasShape = false;
    bool hasBody = false;
    RefBodyType type = RefBodyType::Static;
    uint32_t layer = 1, mask = 0xffffffffu;
    bt::Vec3 aabbHalfExtents{0.5f, 0.5f, 0.5f};
    // "btRigidBody"
    bt::Vec3 origin{0, 0, 0};
    bt::Quat orn{0, 0, 0, 1};
    bt::Mat3 basis{{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}};
    bt::Vec3 linvel{0, 0, 0};
    bt::Vec3 angvel{0, 0, 0};
    float invMass = 0.0f;
    bool freshPose = true; // orientation was (re)posed and not yet written back
    int activation = 1;           // btCollisionObject::m_activationState1 (ACTIVE_TAG 1 ... DISABLE_DEACTIVATION 4)
    float deactivationTime = 0.0f; // btCollisionObject::m_deactivationTime
    float aabbMin[3] = {0, 0, 0};
    float aabbMax[3] = {0, 0, 0};
    // ground contact (contact_ref.h): the collider as Bullet holds it, mass properties, the manifold with the plane
    ct::Shape shape;
    float mass = 0.0f, friction = 0.5f, restitution = 0.0f;
    bt::Vec3 localInertia{0, 0, 0}, invInertiaLocal{0, 0, 0};
    float contactBreakingThreshold = 0.02f;
    ct::Manifold ground;
    // contacts with Static / Kinematic boxes (boxbox_ref.h): this Dynamic body's manifolds, ascending entity id of the other box
    std::vector<ct::BoxManifold> boxes;
    std::vector<uint32_t> boxGeneration; // generation of the other body when the manifold was made (a re-created body is a new pair)
    uint32_t generation = 0;             // bumped whenever the btRigidBody is (re)created
    bool collidedGround = false;         // (scratch of one sub-step: the plane pair was collided)
    bool hadGravity = true;              // applyGravity of this stepSimulation call reached the body (it was not asleep then)
};

// Trigger volumes.  Follows:
//   src/physics/PhysicsSystem.cpp:523-590   EnsureTrigger: ghost object (CF_NO_CONTACT_RESPONSE | CF_STATIC_OBJECT) whose
//                                            shape is CreateShape(trigger.shape, trigger.size); layer 0 -> 4; while the
//                                            trigger is active its pose is set from the Transform EVERY tick (:575);
//                                            (re)activation clears the remembered overlaps
//   src/physics/PhysicsSystem.cpp:1017-1074 ProcessTriggerEvents: current overlap set vs the previous one ->
//                                            Enter / Stay / Exit; a one-shot trigger deactivates after its first
//                                            non-empty set and forgets it (no Exit later)
// The ghost's overlap list (btPairCachingGhostObject::m_overlappingObjects) is Bullet's pair cache restricted to the ghost:
// btGhostPairCallback::addOverlappingPair (installed at PhysicsSystem.cpp:132-133) is called by
// btHashedOverlappingPairCache::addOverlappingPair for EVERY pair the broadphase reports whose filter passes
// (needsBroadphaseCollision: (group0 & mask1) && (group1 & mask0) — group / mask are the custom values the reference hands
// to addRigidBody / addCollisionObject, PhysicsSystem.cpp:473,577, so Bullet's default "static objects do not collide with
// static objects" mask is NOT in force) and records the other proxy's object in each of the two objects that is a ghost.
// btDbvtBroadphase::createProxy and ::setAabb collide a new / moved leaf against BOTH of its trees, so a ghost pairs with
// Static, Kinematic and Dynamic bodies alike AND with other ghosts (each of the two then lists the other).  As for the
// broadphase the specification is the history-free core of that cache: the ghost's AABB (shape AABB at its pose + 0.02,
// updateSingleAabb) overlaps the other object's fed AABB non-strictly and the filter passes both ways.  The reference maps
// objects to entities (FindEntityByCollisionObject, :672): rigid bodies (:474,491) and ghosts (:578) are registered, the
// ground plane is not (kInvalidEntity, skipped at :1033), the ghost's own entity is skipped (:1033) — so an entity that
// carries both a RigidBody and a TriggerVolume does not report itself — and an entity met both as a body and as a ghost
// counts once (std::unordered_set, :1026).
// One-shot and ghost-ghost: a one-shot trigger that fires is removed from the world INSIDE the loop of ProcessTriggerEvents
// (:1062-1072: removeCollisionObject -> the pair cache drops its pairs -> btGhostPairCallback::removeOverlappingPair takes it
// out of every other ghost's list at once), so a ghost processed LATER in the same loop no longer lists it.  The reference
// walks m_triggerRuntime, a std::unordered_map (PhysicsSystem.h) — an order the language leaves unspecified (MSVC's differs
// from libstdc++'s); the specification here fixes it: ASCENDING ENTITY ID.
// A call that simulates nothing (accumulate, n == 0) runs no collision detection: every list keeps last call's content,
// minus the ghosts that are no longer in the world (deactivated by EnsureTrigger, or fired as one-shot earlier in the loop).
// PARITY STATUS: unpinned (spec-derived); the three Bullet functions named above are located in the reference's exe and
// their tests read off the disassembly by oracle/tools/check_pair_cache.py.
struct RefTriggerEvent {
    int type; // 0 Enter, 1 Stay, 2 Exit   (PhysicsSystem.h:50-62)
    EntityId trigger;
    EntityId other;
};

struct RefTriggerRuntime {
    bool hasGhost = false;
    bool active = false;
    bool oneShot = false;
    uint32_t layer = 0, mask = 0;
    bt::Vec3 aabbHalfExtents{0.5f, 0.5f, 0.5f};
    float aabbMin[3] = {0, 0, 0}, aabbMax[3] = {0, 0, 0};
    std::vector<EntityId> overlaps;      // sorted; the union of the two below = TriggerRuntime::overlaps (what the events are diffed on)
    std::vector<EntityId> overlapBodies; // sorted: entities met as rigid bodies
    std::vector<EntityId> overlapGhosts; // sorted: entities met as other trigger ghosts
    void ClearOverlaps()
    {
        overlaps.clear();
        overlapBodies.clear();
        overlapGhosts.clear();
    }
};

class RefPhysicsSystem {
public:
    float gravityY = -9.81f; // assets/config/physics.json:2, PhysicsSystem.h:87
    int orientMode = kOrientIdeal;
    bool computeAabbs = false;
    bool deactivation = true; // !gDisableDeactivation
    float linearSleepingThreshold = 0.8f, angularSleepingThreshold = 1.0f; // btRigidBodyConstructionInfo defaults
    float deactivationTimeLimit = 2.0f;                                     // gDeactivationTime
    // stepSimulation's clock (off: every Update is exactly one sub-step of dt — the reference's steady state)
    bool accumulate = false;
    float fixedStep = 1.0f / 120.0f; // max(config.fixedStep, kMinStep) (PhysicsSystem.cpp:855)
    int maxSubSteps = 4;
    float localTime = 0.0f;          // btDiscreteDynamicsWorld::m_localTime
    int lastSubSteps = 1;            // stepSimulation's return value (m_lastStepSubsteps)
    // the static plane y = 0 the reference adds to every world (PhysicsSystem.cpp:149-166) with Bullet's narrowphase and
    // solver for it (contact_ref.h).  Off by default: BASELINE's workloads are free bodies.  A body whose
    // mask lacks btBroadphaseProxy::StaticFilter (2) does not collide with it.
    bool groundPlane = false;
    // Dynamic boxes collide with the Static / Kinematic BOX colliders of the scene (btBoxBoxCollisionAlgorithm, boxbox_ref.h) —
    // what the reference's world does for every pair the dispatcher accepts (PhysicsSystem.cpp:122-128); capsules against boxes
    // (GJK / EPA) are not restated.  Off by default, like the plane: BASELINE's workloads are free bodies.
    bool staticContacts = false;
    // Dynamic boxes collide with each other (btBoxBoxCollisionAlgorithm per pair, simulation islands of several bodies solved
    // together and put to sleep together: island_ref.h and CollideDynamicPairs / StepIsland below).  Off by default.
    bool dynamicContacts = false;

    struct DynPair {
        ct::BoxManifold m;
        uint32_t genA = 0, genB = 0;
    };

    std::unordered_map<EntityId, RefBodyRuntime>& Runtimes() { return runtime_; }
    const std::map<std::pair<EntityId, EntityId>, DynPair>& DynamicPairs() const { return dynPairs_; }
    std::unordered_map<EntityId, RefTriggerRuntime>& TriggerRuntimes() { return triggerRuntime_; }
    const std::vector<RefTriggerEvent>& LastTriggerEvents() const { return events_; }

    // The rigid-body slice of PhysicsSystem::Update(scene, camera, input, dt).
    void Update(RefScene& scene, double dt)
    {
        // :1222-1246 prune
        std::vector<EntityId> gone;
        for (const auto& kv : runtime_) {
            if (!scene.IsAlive(kv.first) || !scene.GetRigidBody(kv.first) || !scene.GetCollider(kv.first)) {
                gone.push_back(kv.first);
            }
        }
        for (EntityId id : gone) runtime_.erase(id);

        // :1248-1260 ensure
        for (auto& kv : scene.GetRigidBodies()) {
            if (!scene.IsAlive(kv.first)) continue;
            RefCollider* collider = scene.GetCollider(kv.first);
            if (!collider) continue;
            EnsureRigidBody(scene, kv.first, *collider, kv.second);
        }
        // :1262-1269 EnsureTrigger for every TriggerVolume (+ :1234-1246 pruning)
        std::vector<EntityId> goneTriggers;
        for (const auto& kv : triggerRuntime_) {
            if (!scene.IsAlive(kv.first) || !scene.GetTriggerVolume(kv.first)) goneTriggers.push_back(kv.first);
        }
        for (EntityId id : goneTriggers) triggerRuntime_.erase(id);
        for (auto& kv : scene.GetTriggerVolumes()) {
            if (scene.IsAlive(kv.first)) EnsureTrigger(scene, kv.first, kv.second);
        }
        SyncKinematicBodiesToPhysics(scene);
        const float timeStep = static_cast<float>(dt);
        int run = 1;
        float step = timeStep;
        lastSubSteps = 1;
        if (accumulate) {
            int due = 0;
            localTime = localTime + timeStep;
            if (localTime >= fixedStep) {
                due = static_cast<int>(localTime / fixedStep);
                localTime = localTime - static_cast<float>(due) * fixedStep;
            }
            lastSubSteps = due;
            run = std::min(due, maxSubSteps);
            step = fixedStep;
        }
        // applyGravity, once per stepSimulation call: only bodies that are active then
        for (auto& kv : runtime_) kv.second.hadGravity = kv.second.activation != kIslandSleeping;
        for (int k = 0; k < run; ++k) StepSimulation(step);
        SyncRigidBodiesFromPhysics(scene);
        ProcessTriggerEvents(scene, run == 0);
    }

    // Extension used by the synthetic workloads (the reference has no API to give a body an
    // initial velocity; bodies only ever gain velocity from gravity/contacts).
    void SetVelocity(EntityId id, const bt::Vec3& lin, const bt::Vec3& ang)
    {
        auto it = runtime_.find(id);
        if (it != runtime_.end()) {
            it->second.linvel = lin;
            it->second.angvel = ang;
        }
    }

private:
    static void PoseFromTransform(RefBodyRuntime& rt, const RefTransform& t)
    {
        rt.origin = bt::Vec3{t.position.x, t.position.y, t.position.z};
        rt.orn = bt::QuatFromTransformEuler(t.rotationEuler.x, t.rotationEuler.y, t.rotationEuler.z);
        rt.basis = bt::MatFromQuat(rt.orn); // btTransform::setRotation
        rt.freshPose = true;
    }

    void EnsureRigidBody(RefScene& scene, EntityId entity, RefCollider& collider, RefRigidBody& body)
    {
        RefTransform* transform = scene.GetTransform(entity);
        if (!transform) return;
        auto res = runtime_.try_emplace(entity);
        bool inserted = res.second;
        RefBodyRuntime& rt = res.first->second;

        if (collider.dirty || !rt.hasShape) {
            if (collider.shape == RefShape::Capsule) {
                const float radius = std::max(collider.size.x, 0.01f);
                const float halfHeight = std::max(collider.size.y, 0.0f);
                // btCapsuleShape(radius, 2*halfHeight): m_implicitShapeDimensions.y = 0.5*height
                rt.aabbHalfExtents = bt::CapsuleAabbHalfExtents(radius, 0.5f * (halfHeight * 2.0f));
                rt.shape.capsule = true;
                rt.shape.dims = bt::Vec3{radius, 0.5f * (halfHeight * 2.0f), radius};
            } else {
                rt.aabbHalfExtents = bt::BoxAabbHalfExtents(std::max(collider.size.x, 0.01f),
                                                           std::max(collider.size.y, 0.01f),
                                                           std::max(collider.size.z, 0.01f));
                rt.shape.capsule = false;
                rt.shape.dims = rt.aabbHalfExtents; // getHalfExtentsWithMargin() = implicit dimensions + margin
            }
            rt.contactBreakingThreshold = ct::ContactBreakingThreshold(rt.shape);
            rt.hasShape = true;
            collider.dirty = false;
            inserted = true;
        }

        const uint32_t desiredLayer = body.layer ? body.layer : 1u;
        if (inserted || !rt.hasBody || body.dirty) {
            float mass = 0.0f;
            if (body.type == RefBodyType::Dynamic) mass = std::max(body.mass, 0.01f);
            rt.invMass = mass != 0.0f ? 1.0f / mass : 0.0f; // btRigidBody::setMassProps
            rt.mass = mass;
            // shape->calculateLocalInertia(mass, inertia) (PhysicsSystem.cpp:431-434), info.m_friction = body.friction (:437)
            rt.localInertia = mass > 0.0f ? ct::LocalInertia(rt.shape, mass) : bt::Vec3{0, 0, 0};
            rt.invInertiaLocal = ct::InvInertiaLocal(rt.localInertia);
            rt.friction = body.friction;
            rt.restitution = body.restitution; // info.m_restitution (:438)
            rt.ground.Clear(); // removeRigidBody drops the broadphase pair and with it the manifold
            rt.boxes.clear();
            rt.boxGeneration.clear();
            rt.generation += 1;
            PoseFromTransform(rt, *transform);
            rt.linvel = bt::Vec3{0, 0, 0};
            rt.angvel = bt::Vec3{0, 0, 0};
            rt.hasBody = true;
            body.dirty = false;
            // new btRigidBody: ACTIVE_TAG (:462) / DISABLE_DEACTIVATION for Kinematic (:457); addRigidBody puts a
            // static object to ISLAND_SLEEPING
            rt.activation = body.type == RefBodyType::Kinematic ? kDisableDeactivation
                            : body.type == RefBodyType::Static  ? kIslandSleeping
                                                                : kActiveTag;
            rt.deactivationTime = 0.0f;
        }
        rt.type = body.type;
        rt.layer = desiredLayer;
        rt.mask = body.mask;
    }

    void SyncKinematicBodiesToPhysics(RefScene& scene)
    {
        for (auto& kv : runtime_) {
            RefRigidBody* body = scene.GetRigidBody(kv.first);
            RefTransform* transform = scene.GetTransform(kv.first);
            RefBodyRuntime& rt = kv.second;
            if (!body || !transform || !rt.hasBody) continue;
            if (!transform->dirty && !body->dirty) continue; // both branches of :963-971
            PoseFromTransform(rt, *transform);
            if (body->type == RefBodyType::Dynamic) {
                rt.linvel = bt::Vec3{0, 0, 0};
                rt.angvel = bt::Vec3{0, 0, 0};
            }
            body->dirty = false;
        }
    }

    void StepSimulation(float dt)
    {
        const bt::Vec3 g{0.0f, gravityY, 0.0f};
        // predictUnconstraintMotion + updateAabbs for every body, from the state the sub-step starts with (a body's fed box
        // depends on nothing but its own state, so this pass is what the per-body loop below used to do in place)
        if (computeAabbs || staticContacts || dynamicContacts) {
            for (auto& kv : runtime_) {
                RefBodyRuntime& rt = kv.second;
                if (!rt.hasBody) continue;
                const bool dynamic = rt.type == RefBodyType::Dynamic && rt.invMass != 0.0f;
                const bool spinning = rt.angvel.x != 0.0f || rt.angvel.y != 0.0f || rt.angvel.z != 0.0f;
                const bool rotate = orientMode != kOrientIdeal || spinning;
                bt::AabbOfPose(rt.origin, rt.basis, rt.aabbHalfExtents, rt.aabbMin, rt.aabbMax);
                if (dynamic) {
                    // predicted (interpolation) transform uses the velocity BEFORE the gravity impulse
                    const bt::Vec3 p{rt.origin.x + rt.linvel.x * dt, rt.origin.y + rt.linvel.y * dt,
                                     rt.origin.z + rt.linvel.z * dt};
                    bt::Mat3 pb = rt.basis;
                    if (rotate) pb = bt::MatFromQuat(bt::IntegrateOrientation(CurrentOrn(rt), rt.angvel, dt));
                    float mn[3], mx[3];
                    bt::AabbOfPose(p, pb, rt.aabbHalfExtents, mn, mx);
                    for (int a = 0; a < 3; ++a) {
                        rt.aabbMin[a] = std::min(rt.aabbMin[a], mn[a]);
                        rt.aabbMax[a] = std::max(rt.aabbMax[a], mx[a]);
                    }
                }
            }
        }
        // the boxes a Dynamic box can rest on: every Static / Kinematic body with a box collider, ascending entity id
        std::vector<std::pair<EntityId, const RefBodyRuntime*>> obstacles;
        if (staticContacts) {
            for (const auto& kv : runtime_) {
                const RefBodyRuntime& o = kv.second;
                if (o.hasBody && !o.shape.capsule && !(o.type == RefBodyType::Dynamic && o.invMass != 0.0f)) obstacles.emplace_back(kv.first, &o);
            }
            std::sort(obstacles.begin(), obstacles.end());
        }
        // performDiscreteCollisionDetection for the pairs of Dynamic boxes, then the islands they form (calculateSimulationIslands)
        std::unordered_map<EntityId, std::vector<EntityId>> islands; // lowest entity of a multi-body island -> its bodies, ascending
        std::unordered_map<EntityId, EntityId> islandOf;             // body of a multi-body island -> that lowest entity
        if (dynamicContacts) CollideDynamicPairs(islands, islandOf);

        for (auto& kv : runtime_) {
            RefBodyRuntime& rt = kv.second;
            if (!rt.hasBody) continue;
            const bool dynamic = rt.type == RefBodyType::Dynamic && rt.invMass != 0.0f;
            if (!dynamic) continue;
            // performDiscreteCollisionDetection comes BEFORE the island build: a body that wants to sleep (isActive() is still true
            // for WANTS_DEACTIVATION) is collided once more — its manifolds are refreshed — and only then falls asleep; a sleeping
            // body's pairs with static objects are skipped (btCollisionDispatcher::needsCollision: neither object is active)
            rt.collidedGround = CollideOwn(kv.first, rt, obstacles);
        }
        for (auto& kv : runtime_) {
            RefBodyRuntime& rt = kv.second;
            if (!rt.hasBody) continue;
            const bool dynamic = rt.type == RefBodyType::Dynamic && rt.invMass != 0.0f;
            if (!dynamic || islandOf.count(kv.first)) continue;
            StepFreeBody(rt, g, dt);
        }
        for (auto& isl : islands) StepIsland(isl.second, g, dt);
    }

    // a Dynamic body's own pairs: the plane and the Static / Kinematic boxes; returns whether the plane pair was collided
    bool CollideOwn(EntityId self, RefBodyRuntime& rt, const std::vector<std::pair<EntityId, const RefBodyRuntime*>>& obstacles)
    {
        const bool collides = rt.activation != kIslandSleeping;
        const bool withGround = collides && groundPlane && (rt.mask & 2u) != 0u;
        if (withGround) {
            // the pair (ground, body): group StaticFilter = 2 against the body's mask, the body's group against AllFilter
            ct::CollideWithGround(rt.ground, rt.shape, rt.contactBreakingThreshold, rt.origin, rt.basis);
        }
        if (collides && staticContacts && !rt.shape.capsule) CollideWithBoxes(self, rt, obstacles);
        return withGround;
    }

    // the island {body}: build, solve, integrate, updateActivationState
    void StepFreeBody(RefBodyRuntime& rt, const bt::Vec3& g, float dt)
    {
        const bool withGround = rt.collidedGround;
        const bool spinning = rt.angvel.x != 0.0f || rt.angvel.y != 0.0f || rt.angvel.z != 0.0f;
        // buildIslands: a free body is a one-body island; "all sleeping" unless ACTIVE_TAG / DISABLE_DEACTIVATION
        if (rt.activation == kWantsDeactivation) rt.activation = kIslandSleeping;
        if (rt.activation == kIslandSleeping) {
            // no gravity (it was asleep when applyGravity ran, or its force is dropped by clearForces), not solved,
            // not integrated; updateActivationState zeroes the velocities of a sleeping body every step
            rt.linvel = bt::Vec3{0, 0, 0};
            rt.angvel = bt::Vec3{0, 0, 0};
            return;
        }

        // applyGravity + solver write-back of the external force impulse.  m_gravity = acceleration / m_inverseMass:
        // a division per component in the reference's build (btRigidBody::setGravity, check_bullet_order.py)
        const bt::Vec3 force = rt.hadGravity ? bt::Vec3{g.x / rt.invMass, g.y / rt.invMass, g.z / rt.invMass} : bt::Vec3{0, 0, 0};
        bool solved = false;
        bool touching = withGround && rt.ground.n > 0;
        for (const ct::BoxManifold& bm : rt.boxes) touching = touching || bm.n > 0;
        if ((withGround || !rt.boxes.empty()) && (touching || spinning)) {
            // the island {body} through the solver.  A body without a contact and without angular velocity takes the plain
            // update below — the same arithmetic, (v + 0) + impulse.
            ct::BodyState b{rt.origin, rt.linvel, rt.angvel, CurrentOrn(rt), rt.basis};
            const bool moved = ct::SolveBody(b, withGround ? &rt.ground : nullptr, rt.boxes.data(), static_cast<int>(rt.boxes.size()), rt.invMass,
                                             rt.invInertiaLocal, rt.friction, force, dt);
            rt.linvel = b.linVel;
            rt.angvel = b.angVel;
            if (moved) { // the split impulse corrected the pose
                rt.origin = b.origin;
                rt.orn = b.orn;
                rt.basis = b.basis;
                rt.freshPose = true;
            }
            solved = true;
        }
        if (!solved) {
            rt.linvel.x = rt.linvel.x + (force.x * rt.invMass) * dt;
            rt.linvel.y = rt.linvel.y + (force.y * rt.invMass) * dt;
            rt.linvel.z = rt.linvel.z + (force.z * rt.invMass) * dt;
        }
        IntegrateAndUpdateActivation(rt, dt);
    }

    // integrateTransforms (with the velocities the solver left) and updateActivationState for an active body
    void IntegrateAndUpdateActivation(RefBodyRuntime& rt, float dt)
    {
        const bool spinningNow = rt.angvel.x != 0.0f || rt.angvel.y != 0.0f || rt.angvel.z != 0.0f;
        const bool rotateNow = orientMode != kOrientIdeal || spinningNow;
        rt.origin.x = rt.origin.x + rt.linvel.x * dt;
        rt.origin.y = rt.origin.y + rt.linvel.y * dt;
        rt.origin.z = rt.origin.z + rt.linvel.z * dt;
        if (rotateNow) {
            rt.orn = bt::IntegrateOrientation(CurrentOrn(rt), rt.angvel, dt);
            rt.basis = bt::MatFromQuat(rt.orn);
            rt.freshPose = true;
        }

        // updateActivationState: updateDeactivation + wantsSleeping.  The state is ACTIVE_TAG here, or — in an island of several
        // bodies that is kept awake by another body — WANTS_DEACTIVATION: such a body stays that way while it is slow
        // (wantsSleeping() is true for the state itself) and turns ACTIVE_TAG again, timer 0, once it is faster than the thresholds
        if (rt.activation != kDisableDeactivation) {
            const float lin2 = rt.linvel.x * rt.linvel.x + rt.linvel.y * rt.linvel.y + rt.linvel.z * rt.linvel.z;
            const float ang2 = rt.angvel.x * rt.angvel.x + rt.angvel.y * rt.angvel.y + rt.angvel.z * rt.angvel.z;
            bool fast = false;
            if (lin2 < linearSleepingThreshold * linearSleepingThreshold &&
                ang2 < angularSleepingThreshold * angularSleepingThreshold) {
                rt.deactivationTime = rt.deactivationTime + dt;
            } else {
                rt.deactivationTime = 0.0f;
                fast = true; // setActivationState(0)
            }
            // wantsSleeping: never while gDisableDeactivation or gDeactivationTime == 0
            const bool may = deactivation && deactivationTimeLimit != 0.0f;
            if (rt.activation == kWantsDeactivation) {
                if (fast || !may) rt.activation = kActiveTag;
            } else if (may && rt.deactivationTime > deactivationTimeLimit) {
                rt.activation = kWantsDeactivation;
            }
        }
    }

    // Pairs of Dynamic boxes.  As everywhere the pair cache is its history-free core: the fed AABBs overlap (non-strictly) and the
    // filter passes both ways; body A (the manifold's body0) is the lower entity id.  A pair is collided when at least one of
    // the two is active (btCollisionDispatcher::needsCollision; WANTS_DEACTIVATION counts as active), its manifold lives as long
    // as the pair and both btRigidBody generations.  btSimulationIslandManager::findUnions unites the two bodies of EVERY pair in
    // the cache, touching or not, asleep or not.

    void CollideDynamicPairs(std::unordered_map<EntityId, std::vector<EntityId>>& islands, std::unordered_map<EntityId, EntityId>& islandOf)
    {
        std::vector<EntityId> ids;
        for (const auto& kv : runtime_) {
            const RefBodyRuntime& rt = kv.second;
            if (rt.hasBody && rt.type == RefBodyType::Dynamic && rt.invMass != 0.0f && !rt.shape.capsule) ids.push_back(kv.first);
        }
        std::sort(ids.begin(), ids.end(), [&](EntityId a, EntityId b) {
            const float xa = runtime_[a].aabbMin[0], xb = runtime_[b].aabbMin[0];
            return xa < xb || (xa == xb && a < b);
        });
        std::map<std::pair<EntityId, EntityId>, DynPair> next;
        std::unordered_map<EntityId, EntityId> parent;
        auto find = [&](EntityId e) {
            while (true) {
                auto it = parent.find(e);
                if (it == parent.end() || it->second == e) return e;
                e = it->second;
            }
        };
        for (size_t i = 0; i < ids.size(); ++i) {
            const RefBodyRuntime& p = runtime_[ids[i]];
            for (size_t j = i + 1; j < ids.size(); ++j) {
                const RefBodyRuntime& q = runtime_[ids[j]];
                if (!(q.aabbMin[0] <= p.aabbMax[0])) break;
                if (!BoxesOverlap(p.aabbMin, p.aabbMax, q.aabbMin, q.aabbMax)) continue;
                if ((p.layer & q.mask) == 0 || (q.layer & p.mask) == 0) continue;
                const EntityId ea = std::min(ids[i], ids[j]), eb = std::max(ids[i], ids[j]);
                RefBodyRuntime& a = runtime_[ea];
                RefBodyRuntime& b = runtime_[eb];
                DynPair e;
                auto old = dynPairs_.find({ea, eb});
                if (old != dynPairs_.end() && old->second.genA == a.generation && old->second.genB == b.generation) {
                    e = old->second;
                } else {
                    e.genA = a.generation;
                    e.genB = b.generation;
                    e.m.other = eb;
                    e.m.breaking = std::min(a.contactBreakingThreshold, b.contactBreakingThreshold);
                    e.m.friction = std::max(-10.0f, std::min(10.0f, a.friction * b.friction));
                    e.m.restitution = a.restitution * b.restitution;
                }
                if (a.activation != kIslandSleeping || b.activation != kIslandSleeping) {
                    const ct::BoxPose pa{a.origin, a.basis, a.shape.dims}, pb{b.origin, b.basis, b.shape.dims};
                    ct::CollideBoxBox(e.m, pa, pb);
                }
                next[{ea, eb}] = e;
                const EntityId ra = find(ea), rb = find(eb);
                parent.emplace(ea, ea);
                parent.emplace(eb, eb);
                if (ra != rb) parent[std::max(ra, rb)] = std::min(ra, rb);
            }
        }
        dynPairs_.swap(next);
        for (const auto& kv : parent) {
            const EntityId root = find(kv.first);
            islandOf[kv.first] = root;
            islands[root].push_back(kv.first);
        }
        for (auto& kv : islands) std::sort(kv.second.begin(), kv.second.end());
    }

    // One island of several Dynamic bodies (ascending entity id): btSimulationIslandManager::buildIslands' activation rule, then
    // solveGroup over all its manifolds (island_ref.h), integrateTransforms and updateActivationState body by body.
    void StepIsland(const std::vector<EntityId>& ids, const bt::Vec3& g, float dt)
    {
        bool allSleeping = true;
        for (EntityId id : ids) {
            const int st = runtime_[id].activation;
            if (st == kActiveTag || st == kDisableDeactivation) allSleeping = false;
        }
        if (allSleeping) {
            for (EntityId id : ids) {
                RefBodyRuntime& rt = runtime_[id];
                rt.activation = kIslandSleeping;
                rt.linvel = bt::Vec3{0, 0, 0}; // (updateActivationState, every step)
                rt.angvel = bt::Vec3{0, 0, 0};
            }
            return;
        }
        std::vector<ct::BodyState> state(ids.size());
        std::vector<ct::IslandBody> bodies(ids.size());
        std::vector<ct::IslandManifold> ms;
        std::unordered_map<EntityId, int> indexOf;
        for (size_t i = 0; i < ids.size(); ++i) indexOf[ids[i]] = static_cast<int>(i);
        for (size_t i = 0; i < ids.size(); ++i) {
            RefBodyRuntime& rt = runtime_[ids[i]];
            if (rt.activation == kIslandSleeping) { // woken by its island
                rt.activation = kWantsDeactivation;
                rt.deactivationTime = 0.0f;
                // It was not collided this step, so its manifolds with Static / Kinematic boxes are what its last collision left —
                // minus the pairs that ended while it slept (Bullet's pair cache follows every object, asleep or not): the obstacle
                // is gone or re-created, the filter no longer passes, the fed AABBs no longer overlap
                std::vector<ct::BoxManifold> keep;
                std::vector<uint32_t> keepGen;
                for (size_t k = 0; k < rt.boxes.size(); ++k) {
                    auto it = runtime_.find(rt.boxes[k].other);
                    if (it == runtime_.end()) continue;
                    const RefBodyRuntime& o = it->second;
                    if (!o.hasBody || o.shape.capsule || (o.type == RefBodyType::Dynamic && o.invMass != 0.0f)) continue;
                    if (o.generation != rt.boxGeneration[k]) continue;
                    if ((rt.layer & o.mask) == 0 || (o.layer & rt.mask) == 0) continue;
                    if (!BoxesOverlap(rt.aabbMin, rt.aabbMax, o.aabbMin, o.aabbMax)) continue;
                    keep.push_back(rt.boxes[k]);
                    keepGen.push_back(rt.boxGeneration[k]);
                }
                rt.boxes.swap(keep);
                rt.boxGeneration.swap(keepGen);
            }
            state[i] = ct::BodyState{rt.origin, rt.linvel, rt.angvel, CurrentOrn(rt), rt.basis};
            bodies[i].state = &state[i];
            bodies[i].invMass = rt.invMass;
            bodies[i].invInertiaLocal = rt.invInertiaLocal;
            // m_totalForce: gravity / invMass from this call's applyGravity — which skipped the body if it was asleep then
            bodies[i].force = rt.hadGravity ? bt::Vec3{g.x / rt.invMass, g.y / rt.invMass, g.z / rt.invMass} : bt::Vec3{0, 0, 0};
            // its pair with the plane exists whether or not it was collided this step (a body woken just now keeps its cached points)
            const bool planePair = groundPlane && (rt.mask & 2u) != 0u;
            ct::AppendOwnManifolds(ms, static_cast<int>(i), planePair ? &rt.ground : nullptr, rt.boxes.data(), static_cast<int>(rt.boxes.size()), rt.friction);
            for (auto it = dynPairs_.lower_bound({ids[i], 0}); it != dynPairs_.end() && it->first.first == ids[i]; ++it) {
                ct::BoxManifold& bm = it->second.m;
                ct::IslandManifold m;
                m.a = static_cast<int>(i);
                m.b = indexOf.at(it->first.second);
                m.friction = bm.friction;
                m.restitution = bm.restitution;
                m.n = bm.n;
                for (int j = 0; j < bm.n; ++j) {
                    ct::BoxPoint& cp = bm.p[j];
                    m.p[j] = ct::IslandPoint{cp.worldA, cp.worldB, cp.normalB, cp.distance, &cp.appliedImpulse, &cp.appliedImpulseLateral1};
                }
                ms.push_back(m);
            }
        }
        ct::SolveIsland(bodies.data(), static_cast<int>(bodies.size()), ms.data(), static_cast<int>(ms.size()), dt);
        for (size_t i = 0; i < ids.size(); ++i) {
            RefBodyRuntime& rt = runtime_[ids[i]];
            rt.linvel = state[i].linVel;
            rt.angvel = state[i].angVel;
            if (bodies[i].moved) {
                rt.origin = state[i].origin;
                rt.orn = state[i].orn;
                rt.basis = state[i].basis;
                rt.freshPose = true;
            }
            IntegrateAndUpdateActivation(rt, dt);
        }
    }

    // The Dynamic box `rt` against the boxes it can touch this step: pairs = fed AABBs overlap and the filter passes both ways
    // (the history-free core of the pair cache), at most ct::kMaxBoxManifolds of them, lowest entity ids first.  A manifold
    // lives as long as its pair; a re-created other body is a new pair.
    void CollideWithBoxes(EntityId self, RefBodyRuntime& rt, const std::vector<std::pair<EntityId, const RefBodyRuntime*>>& obstacles)
    {
        std::vector<ct::BoxManifold> next;
        std::vector<uint32_t> nextGen;
        for (const auto& ob : obstacles) {
            if (static_cast<int>(next.size()) == ct::kMaxBoxManifolds) break;
            const RefBodyRuntime& o = *ob.second;
            if (ob.first == self) continue;
            if ((rt.layer & o.mask) == 0 || (o.layer & rt.mask) == 0) continue;
            bool overlap = true;
            for (int a = 0; a < 3; ++a) overlap = overlap && rt.aabbMin[a] <= o.aabbMax[a] && rt.aabbMax[a] >= o.aabbMin[a];
            if (!overlap) continue;
            ct::BoxManifold m;
            bool found = false;
            for (size_t k = 0; k < rt.boxes.size(); ++k) {
                if (rt.boxes[k].other == ob.first && rt.boxGeneration[k] == o.generation) {
                    m = rt.boxes[k];
                    found = true;
                }
            }
            if (!found) {
                m.other = ob.first;
                m.breaking = std::min(rt.contactBreakingThreshold, o.contactBreakingThreshold);
                m.friction = std::max(-10.0f, std::min(10.0f, rt.friction * o.friction)); // btManifoldResult::calculateCombinedFriction
                m.restitution = rt.restitution * o.restitution;                             // ... calculateCombinedRestitution
            }
            const ct::BoxPose a{rt.origin, rt.basis, rt.shape.dims}, b{o.origin, o.basis, o.shape.dims};
            ct::CollideBoxBox(m, a, b);
            next.push_back(m);
            nextGen.push_back(o.generation);
        }
        rt.boxes.swap(next);
        rt.boxGeneration.swap(nextGen);
    }

    bt::Quat CurrentOrn(const RefBodyRuntime& rt) const
    {
        return orientMode == kOrientBasis ? bt::QuatFromMat(rt.basis) : rt.orn;
    }

    void SyncRigidBodiesFromPhysics(RefScene& scene)
    {
        for (auto& kv : runtime_) {
            RefRigidBody* body = scene.GetRigidBody(kv.first);
            RefBodyRuntime& rt = kv.second;
            if (!body || !rt.hasBody) continue;
            if (body->type != RefBodyType::Dynamic) continue;
            RefTransform* transform = scene.GetTransform(kv.first);
            if (!transform) continue;
            transform->position = Float3{rt.origin.x, rt.origin.y, rt.origin.z};
            if (orientMode != kOrientIdeal || rt.freshPose) {
                const bt::Vec3 e = bt::TransformEulerFromMat(rt.basis);
                transform->rotationEuler = Float3{e.x, e.y, e.z};
                rt.freshPose = false;
            }
            transform->MarkDirty();
        }
    }

    static bt::Vec3 ShapeHalfExtents(RefShape shape, const Float3& size)
    {
        if (shape == RefShape::Capsule) {
            const float radius = std::max(size.x, 0.01f);
            const float halfHeight = std::max(size.y, 0.0f);
            return bt::CapsuleAabbHalfExtents(radius, 0.5f * (halfHeight * 2.0f));
        }
        return bt::BoxAabbHalfExtents(std::max(size.x, 0.01f), std::max(size.y, 0.01f), std::max(size.z, 0.01f));
    }

    void EnsureTrigger(RefScene& scene, EntityId entity, RefTriggerVolume& trigger)
    {
        RefTransform* transform = scene.GetTransform(entity);
        if (!transform) return;
        RefTriggerRuntime& rt = triggerRuntime_.try_emplace(entity).first->second;
        if (trigger.dirty || !rt.hasGhost) {
            rt.aabbHalfExtents = ShapeHalfExtents(trigger.shape, trigger.size);
            trigger.dirty = false;
            rt.hasGhost = true;
        }
        rt.oneShot = trigger.oneShot;
        const uint32_t desiredLayer = trigger.layer ? trigger.layer : 4u; // kDefaultTriggerLayer = 1 << 2
        if (rt.layer != desiredLayer || rt.mask != trigger.mask) {
            rt.layer = desiredLayer;
            rt.mask = trigger.mask;
            rt.active = false;
        }
        if (trigger.active) {
            // ghost->setWorldTransform(MakeBtTransform(*transform)) every tick; its AABB is refreshed by updateAabbs
            const bt::Vec3 origin{transform->position.x, transform->position.y, transform->position.z};
            const bt::Mat3 basis = bt::MatFromQuat(bt::QuatFromTransformEuler(transform->rotationEuler.x, transform->rotationEuler.y,
                                                                              transform->rotationEuler.z));
            bt::AabbOfPose(origin, basis, rt.aabbHalfExtents, rt.aabbMin, rt.aabbMax);
            if (!rt.active) {
                rt.active = true;
                rt.ClearOverlaps();
            }
        } else if (rt.active) {
            rt.active = false;
            rt.ClearOverlaps();
        }
    }

    static bool BoxesOverlap(const float* amn, const float* amx, const float* bmn, const float* bmx)
    {
        bool overlap = true;
        for (int a = 0; a < 3; ++a) overlap = overlap && amn[a] <= bmx[a] && amx[a] >= bmn[a];
        return overlap;
    }

    void ProcessTriggerEvents(RefScene& scene, bool noStep = false)
    {
        events_.clear();
        // (:1019 walks an unordered_map: unspecified order, fixed here — see the header — as ascending entity id)
        std::vector<EntityId> order;
        order.reserve(triggerRuntime_.size());
        for (const auto& kv : triggerRuntime_) order.push_back(kv.first);
        std::sort(order.begin(), order.end());
        for (EntityId id : order) {
            RefTriggerVolume* trigger = scene.GetTriggerVolume(id);
            RefTriggerRuntime& rt = triggerRuntime_[id];
            if (!trigger || !rt.hasGhost || !rt.active) continue;
            std::vector<EntityId> bodies, ghosts;
            if (noStep) {
                // no collision detection ran: the ghost's list is last call's, minus the ghosts no longer in the world
                bodies = rt.overlapBodies;
                for (EntityId g : rt.overlapGhosts) {
                    auto it = triggerRuntime_.find(g);
                    if (it != triggerRuntime_.end() && it->second.hasGhost && it->second.active) ghosts.push_back(g);
                }
            } else {
                for (const auto& bk : runtime_) { // every registered rigid body, whatever its type (:474)
                    const RefBodyRuntime& b = bk.second;
                    if (!b.hasBody || bk.first == id) continue;
                    if ((rt.layer & b.mask) == 0 || (b.layer & rt.mask) == 0) continue;
                    if (BoxesOverlap(rt.aabbMin, rt.aabbMax, b.aabbMin, b.aabbMax)) bodies.push_back(bk.first);
                }
                for (const auto& gk : triggerRuntime_) { // every other ghost that is in the world right now (:578)
                    const RefTriggerRuntime& g = gk.second;
                    if (gk.first == id || !g.hasGhost || !g.active) continue;
                    if ((rt.layer & g.mask) == 0 || (g.layer & rt.mask) == 0) continue;
                    if (BoxesOverlap(rt.aabbMin, rt.aabbMax, g.aabbMin, g.aabbMax)) ghosts.push_back(gk.first);
                }
                std::sort(bodies.begin(), bodies.end());
                std::sort(ghosts.begin(), ghosts.end());
            }
            std::vector<EntityId> current(bodies.size() + ghosts.size());
            current.erase(std::set_union(bodies.begin(), bodies.end(), ghosts.begin(), ghosts.end(), current.begin()), current.end());
            for (EntityId other : current) {
                const bool was = std::binary_search(rt.overlaps.begin(), rt.overlaps.end(), other);
                events_.push_back(RefTriggerEvent{was ? 1 : 0, id, other});
            }
            for (EntityId previous : rt.overlaps) {
                if (!std::binary_search(current.begin(), current.end(), previous)) events_.push_back(RefTriggerEvent{2, id, previous});
            }
            rt.overlaps = std::move(current);
            rt.overlapBodies = std::move(bodies);
            rt.overlapGhosts = std::move(ghosts);
            if (rt.oneShot && !rt.overlaps.empty()) {
                trigger->active = false;
                rt.active = false; // removeCollisionObject: gone from every other ghost's list from here on
                rt.ClearOverlaps();
            }
        }
    }

    std::unordered_map<EntityId, RefBodyRuntime> runtime_;
    std::map<std::pair<EntityId, EntityId>, DynPair> dynPairs_; // the pair cache of Dynamic boxes with its manifolds, (lower, higher) entity
    std::unordered_map<EntityId, RefTriggerRuntime> triggerRuntime_;
    std::vector<RefTriggerEvent> events_;
};

} // namespace orc
