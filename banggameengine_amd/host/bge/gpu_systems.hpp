// bge/gpu_systems.hpp — the reference's system call shapes on top of the C ABI (include/bge_world.h).
//
//   reference call site (src/core/Application.cpp)            replacement (same name, same arguments)
//   :256  m_physics.Update(m_scene, *m_camera, m_input, dt)   m_gpuPhysics.Update(m_scene, *m_camera, m_input, dt)   [rigid-body slice]
//   :284  TransformSystem::Update(m_scene)                    bge::GpuTransformSystem<Scene>::Update(m_scene)
//   :283,285  m_scene.CountDirtyTransforms()                  unchanged (host flags are kept coherent)
//   :84   m_physics.ReloadConfigIfNeeded(m_scene)             m_gpuPhysics.ReloadConfigIfNeeded(m_scene)   [gravity, fixedStep]
//   :324-326  OnSceneReloaded / GetFixedStep                  same names
// GpuPhysicsSystem carries PhysicsSystem's public surface for this path (src/physics/PhysicsSystem.h:43-48, 77):
// SetConfigPath, void Initialize(), OnSceneReloaded, bool ReloadConfigIfNeeded, Update(Scene&, const Camera&, const
// InputSystem&, double), LogStats, GetFixedStep — and steps the world as the reference does: Bullet's
// stepSimulation(dt, 4, max(fixedStep, 1/240)) with its sub-step clock (bge_world_step_simulation).
//
// Header-only and templated on the scene type: it needs only the accessors the reference's Scene already has
// (GetTransforms, GetTransform, GetParent, HasTransform, GetRigidBodies, GetCollider — src/ecs/Scene.h:28-95)
// and the field names of Transform / RigidBody / Collider, so it compiles against the reference's own headers
// as well as against bge/scene.hpp.
//
// Coherence model ("coherent mode"): the host structs stay the source of truth at the points the reference's
// callers read them —  Transform::world and Transform::dirty after TransformSystem::Update,
// Transform::position / rotationEuler / dirty after PhysicsSystem::Update.  Per call the adapter
//   1. re-derives the dense entity index <-> EntityId map and the parent array when the set of Transforms or a
//      parent link changed (the reference has no topology version counter, so this is an O(N) scan — the
//      reference's own ForEachRootTransform does the same scan every call, src/ecs/Scene.cpp:523-533);
//   2. uploads the TRS of Transforms whose `dirty` flag is set (sparse, bge_world_upload_trs_indexed);
//   3. ticks the device world;
//   4. copies the results back into the host structs.
// Errors never throw (the reference's hot path has no exceptions): a failed call logs "[GPU] ..." to stderr and
// returns false, leaving the host structs as they were.
//
// Bodies collide with the ground plane only, not with each other (no convex-convex narrowphase on this path).
// Known deviations (documented in DESIGN.md): Transform::local is not refreshed; a Dynamic body whose Transform
// is edited BETWEEN PhysicsSystem::Update and TransformSystem::Update of the same tick continues from the edited
// position (the reference's Bullet body would ignore the edit).
#pragma once

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <memory>
#include <sstream>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../../include/bge_world.h"
#include "scene_json.hpp" // the JSON reader (physics.json)

namespace bge {

// Trigger events as PhysicsSystem publishes them on its EventBus (src/physics/PhysicsSystem.h:50-62)
struct GpuTriggerEvent {
    enum class Type { Enter, Stay, Exit };
    Type type = Type::Enter;
    uint32_t trigger = 0; // EntityId
    uint32_t other = 0;   // EntityId
};

namespace detail {
template <class S, class = void> struct has_trigger_volumes : std::false_type {};
template <class S> struct has_trigger_volumes<S, std::void_t<decltype(std::declval<S&>().GetTriggerVolumes())>> : std::true_type {};
} // namespace detail

template <class SceneT> class GpuSceneMirror {
public:
    using Id = uint32_t;

    explicit GpuSceneMirror(int device = -1, void* stream = nullptr)
    {
        bge_world_desc d{};
        d.struct_size = sizeof d;
        d.device = device;
        d.stream = stream;
        if (bge_world_create(&d, &world_) != BGE_OK) Log("bge_world_create");
        // every world of the reference has the static plane y = 0 (PhysicsSystem::EnsureGround, PhysicsSystem.cpp:149-166)
        else if (bge_world_set_ground_plane(world_, 1) != BGE_OK) Log("bge_world_set_ground_plane");
        // ... and its dispatcher collides Dynamic bodies with the Static / Kinematic ones (PhysicsSystem.cpp:122-128): box against box
        // here (Bullet's btBoxBoxDetector, include/bge_world.h) — what carries a body on demo.json's "Ground"
        else if (bge_world_set_static_contacts(world_, 1) != BGE_OK) Log("bge_world_set_static_contacts");
        // ... and with each other: boxes pile up, touching bodies sleep and wake as one island (include/bge_world.h)
        else if (bge_world_set_dynamic_contacts(world_, 1) != BGE_OK) Log("bge_world_set_dynamic_contacts");
    }
    ~GpuSceneMirror() { bge_world_destroy(world_); }
    GpuSceneMirror(const GpuSceneMirror&) = delete;
    GpuSceneMirror& operator=(const GpuSceneMirror&) = delete;

    bool ok() const { return world_ != nullptr; }
    bge_world* world() { return world_; }
    float gravity[3] = {0.0f, -9.81f, 0.0f}; // PhysicsSystem.cpp:130 with assets/config/physics.json:2
    // BGE_TICK_BULLET_BASIS: carry every Dynamic body's orientation the way Bullet does (basis round trip each step, euler
    // rewritten from it) instead of leaving a non-spinning body's orientation and euler untouched
    bool bullet_basis = false;
    // The ground plane is part of every reference world; SetGroundPlane(false) turns the bodies into free bodies (what
    // BASELINE's synthetic workloads are).  Takes effect on the next UpdatePhysics.
    void SetGroundPlane(bool on)
    {
        if (ok() && bge_world_set_ground_plane(world_, on ? 1 : 0) != BGE_OK) Log("bge_world_set_ground_plane");
        ground_plane_state = on ? 1 : 0;
    }
    // Contacts of Dynamic boxes with Static / Kinematic boxes (on by default, as in every reference world)
    void SetStaticContacts(bool on)
    {
        if (ok() && bge_world_set_static_contacts(world_, on ? 1 : 0) != BGE_OK) Log("bge_world_set_static_contacts");
    }

    // Contacts of Dynamic boxes with each other (on by default, as in every reference world; a scene of free bodies that never meet
    // saves the sub-step's pair search by switching them off)
    void SetDynamicContacts(bool on)
    {
        if (ok() && bge_world_set_dynamic_contacts(world_, on ? 1 : 0) != BGE_OK) Log("bge_world_set_dynamic_contacts");
    }

    // Resident mode: the world matrices stay on the device after TransformSystem::Update; only the host `dirty` flags
    // are kept coherent and the caller fetches the matrices it needs (FetchWorld) — e.g. the visible set.  Coherent
    // mode (default) copies every world matrix back each call, which is what makes the adapter a drop-in but also what
    // bounds it (64 B per entity over PCIe + a hash-map scatter).
    bool resident = false;

    // world matrices of the given entities straight from the device into their Transform::world
    bool FetchWorld(SceneT& scene, const std::vector<Id>& entities)
    {
        index_list_.clear();
        for (Id id : entities) {
            auto it = index_of_.find(id);
            if (it != index_of_.end()) index_list_.push_back(it->second);
        }
        if (!down_.resize(index_list_.size() * 16 + 1)) return Log("bge_host_alloc");
        if (!index_list_.empty() &&
            bge_world_download_world_indexed(world_, index_list_.size(), index_list_.data(), down_.data()) != BGE_OK) {
            return Log("bge_world_download_world_indexed");
        }
        for (size_t k = 0; k < index_list_.size(); ++k) {
            if (auto* t = scene.GetTransform(ids_[index_list_[k]])) std::memcpy(t->world, &down_[16 * k], 64);
        }
        return true;
    }

    // --- TransformSystem::Update(Scene&)
    bool UpdateTransforms(SceneT& scene)
    {
        if (!ok() || !RefreshTopology(scene) || !UploadDirtyTransforms(scene)) return false;
        if (bge_world_tick(world_, 0.0f, gravity, BGE_TICK_TRANSFORMS) != BGE_OK) return Log("bge_world_tick");
        const size_t n = ids_.size();
        if (resident) {
            limbo_.resize(n);
            if (n && bge_world_download_dirty(world_, 0, n, limbo_.data()) != BGE_OK) return Log("bge_world_download_dirty");
            for (auto& kv : scene.GetTransforms()) {
                const uint32_t i = index_of_[kv.first];
                if (limbo_[i]) continue;
                kv.second.dirty = false;
                written_[i] = 0;
            }
            return true;
        }
        if (!down_.resize(n * 16 + 1)) return Log("bge_host_alloc");
        if (n && bge_world_download_world(world_, 0, n, down_.data()) != BGE_OK) return Log("bge_world_download_world");
        limbo_.resize(n);
        if (n && bge_world_download_dirty(world_, 0, n, limbo_.data()) != BGE_OK) return Log("bge_world_download_dirty");
        for (auto& kv : scene.GetTransforms()) {
            const uint32_t i = index_of_[kv.first];
            if (limbo_[i]) continue; // inside a parent cycle: the reference never reaches it, it stays dirty
            std::memcpy(kv.second.world, &down_[16 * static_cast<size_t>(i)], 64);
            kv.second.dirty = false;
            written_[i] = 0;
        }
        return true;
    }

    // --- rigid-body slice of PhysicsSystem::Update(Scene&, camera, input, dt)
    // How UpdatePhysics steps the world: Bullet's stepSimulation(dt, max_sub_steps, fixed_step) (PhysicsSystem.cpp:855-863).
    // max_sub_steps < 0: exactly one step of dt per call (no clock) — identical while the caller passes dt == fixed_step.
    int max_sub_steps = 4;
    float fixed_step = 1.0f / 120.0f;
    int last_sub_steps = 0; // stepSimulation's return value of the last call
    int ground_plane_state = 1; // what the world was last told (GpuPhysicsSystem keeps it in step with its own switch)

    bool UpdatePhysics(SceneT& scene, double dt)
    {
        if (!ok() || !RefreshTopology(scene) || !UploadBodies(scene) || !UploadDirtyTransforms(scene)) return false;
        uint32_t flags = BGE_TICK_PHYSICS;
        if (bullet_basis) flags |= BGE_TICK_BULLET_BASIS;
        bool triggers = false;
        if constexpr (detail::has_trigger_volumes<SceneT>::value) {
            if (!UploadTriggers(scene, triggers)) return false;
            if (triggers) flags |= BGE_TICK_BROADPHASE; // the ghost overlaps come out of the broadphase step
        }
        if (max_sub_steps < 0) {
            last_sub_steps = 1;
            if (bge_world_tick(world_, static_cast<float>(dt), gravity, flags) != BGE_OK) return Log("bge_world_tick");
        } else if (bge_world_step_simulation(world_, dt, max_sub_steps, fixed_step, gravity, flags, &last_sub_steps) != BGE_OK) {
            return Log("bge_world_step_simulation");
        }
        trigger_events_.clear();
        if constexpr (detail::has_trigger_volumes<SceneT>::value) {
            if (triggers && !FetchTriggerEvents(scene)) return false;
        }
        // SyncRigidBodiesFromPhysics: Dynamic bodies only (PhysicsSystem.cpp:926-948)
        index_list_.clear();
        for (auto& kv : scene.GetRigidBodies()) {
            auto it = index_of_.find(kv.first);
            if (it == index_of_.end() || !body_[it->second].exists) continue;
            if (static_cast<int>(kv.second.type) == 1) index_list_.push_back(it->second);
        }
        const size_t m = index_list_.size();
        if (!down_.resize(m * 6 + 1)) return Log("bge_host_alloc");
        if (m && bge_world_download_pose_indexed(world_, m, index_list_.data(), down_.data(), down_.data() + 3 * m) != BGE_OK) {
            return Log("bge_world_download_pose_indexed");
        }
        for (size_t k = 0; k < m; ++k) {
            const uint32_t i = index_list_[k];
            auto* t = scene.GetTransform(ids_[i]);
            if (!t) continue;
            std::memcpy(static_cast<void*>(&t->position), &down_[3 * k], 12);
            std::memcpy(static_cast<void*>(&t->rotationEuler), &down_[3 * m + 3 * k], 12);
            t->MarkDirty();
            std::memcpy(&last_pose_[6 * static_cast<size_t>(i)], &down_[3 * k], 12);
            std::memcpy(&last_pose_[6 * static_cast<size_t>(i) + 3], &down_[3 * m + 3 * k], 12);
            written_[i] = 1;
        }
        return true;
    }

    // what LogStats prints as "bodies": m_world->getNumCollisionObjects() (PhysicsSystem.cpp:1332) — rigid bodies, active trigger
    // ghosts and, while it is in the world, the ground plane's object (:149-166)
    int CollisionObjectCount()
    {
        bge_world_info info{};
        if (!ok() || bge_world_get_info(world_, &info) != BGE_OK) return 0;
        int n = static_cast<int>(info.n_bodies) + (ground_plane_state ? 1 : 0);
        for (uint8_t a : t_active_) n += a ? 1 : 0;
        return n;
    }

    // Enter / Stay / Exit of the last UpdatePhysics (what ProcessTriggerEvents publishes, PhysicsSystem.cpp:1017-1074)
    const std::vector<GpuTriggerEvent>& TriggerEvents() const { return trigger_events_; }

private:
    // EnsureTrigger (PhysicsSystem.cpp:523-590): the trigger set is re-sent when a TriggerVolume appeared, vanished or
    // changed; remembered overlaps survive on the device side for unchanged triggers
    bool UploadTriggers(SceneT& scene, bool& any)
    {
        auto& vols = scene.GetTriggerVolumes();
        t_entity_.clear(); t_shape_.clear(); t_size_.clear(); t_layer_.clear(); t_mask_.clear(); t_oneshot_.clear(); t_active_.clear();
        bool dirty = false;
        seen_trigger_.assign(ids_.size(), 0);
        // ProcessTriggerEvents' loop order matters once a one-shot trigger overlaps another trigger (bge_world.h): the world walks
        // the uploaded array, the reference an unordered_map (unspecified order) — ascending EntityId here, as in the oracle
        t_order_.clear();
        for (auto& kv : vols) t_order_.push_back(kv.first);
        std::sort(t_order_.begin(), t_order_.end());
        for (const auto id : t_order_) {
            auto vit = vols.find(id);
            auto& kv = *vit;
            auto it = index_of_.find(kv.first);
            uint32_t index;
            bool without_transform = false;
            if (it != index_of_.end()) {
                index = it->second;
            } else {
                // a trigger needs a Transform to be created or re-posed (PhysicsSystem.cpp:530-534) — but a ghost whose entity lost
                // its Transform stays in the world where it was: the entity's index was kept for it (RefreshTopology)
                auto o = orphan_of_.find(kv.first);
                if (o == orphan_of_.end() || !has_trigger_[o->second]) continue;
                index = o->second;
                without_transform = true;
            }
            seen_trigger_[index] = 1;
            has_trigger_[index] = 1;
            auto& tv = kv.second;
            t_entity_.push_back(index);
            t_shape_.push_back(static_cast<uint8_t>(static_cast<int>(tv.shape)));
            t_size_.push_back(tv.size.x); t_size_.push_back(tv.size.y); t_size_.push_back(tv.size.z);
            t_layer_.push_back(tv.layer);
            t_mask_.push_back(tv.mask);
            t_oneshot_.push_back(tv.oneShot ? 1 : 0);
            t_active_.push_back(tv.active ? 1 : 0);
            if (!without_transform) { // (EnsureTrigger does not reach the other ones: their dirty flag waits)
                dirty = dirty || tv.dirty;
                tv.dirty = false;
            }
        }
        for (uint32_t i = 0; i < ids_.size(); ++i) {
            if (has_trigger_[i] && !seen_trigger_[i]) { // the component is gone
                has_trigger_[i] = 0;
                RetireIfNothingLeft(i);
            }
        }
        any = !t_entity_.empty();
        // signature of the set: everything that reaches the device
        std::vector<uint32_t> sig;
        sig.reserve(t_entity_.size() * 8);
        for (size_t k = 0; k < t_entity_.size(); ++k) {
            uint32_t sz[3];
            std::memcpy(sz, &t_size_[3 * k], 12);
            sig.insert(sig.end(), {t_entity_[k], t_shape_[k], sz[0], sz[1], sz[2], t_layer_[k], t_mask_[k],
                                   static_cast<uint32_t>(t_oneshot_[k] | (t_active_[k] << 1))});
        }
        if (!dirty && sig == t_signature_) return true;
        t_signature_ = sig;
        if (bge_world_upload_triggers(world_, t_entity_.size(), t_entity_.data(), t_shape_.data(), t_size_.data(), t_layer_.data(),
                                      t_mask_.data(), t_oneshot_.data(), t_active_.data()) != BGE_OK) {
            return Log("bge_world_upload_triggers");
        }
        return true;
    }

    bool FetchTriggerEvents(SceneT& scene)
    {
        uint64_t total = 0;
        if (bge_world_trigger_events(world_, nullptr, 0, &total) != BGE_OK) return Log("bge_world_trigger_events");
        raw_events_.resize(total);
        if (bge_world_trigger_events(world_, raw_events_.data(), total, &total) != BGE_OK) return Log("bge_world_trigger_events");
        for (const bge_trigger_event& e : raw_events_) {
            trigger_events_.push_back(GpuTriggerEvent{static_cast<GpuTriggerEvent::Type>(e.type), last_ids_[e.trigger], last_ids_[e.other]});
        }
        // one-shot triggers that fired are now inactive (PhysicsSystem.cpp:1062-1071)
        if (!t_entity_.empty()) {
            t_active_.resize(t_entity_.size());
            if (bge_world_trigger_active(world_, t_entity_.size(), t_entity_.data(), t_active_.data()) != BGE_OK) return Log("bge_world_trigger_active");
            for (size_t k = 0; k < t_entity_.size(); ++k) {
                if (auto* tv = scene.GetTriggerVolume(ids_[t_entity_[k]])) {
                    if (tv->active && !t_active_[k]) tv->active = false;
                }
            }
        }
        return true;
    }

    struct BodyState {
        bool exists = false;
    };

    bool Log(const char* what) const
    {
        std::fprintf(stderr, "[GPU] %s failed: %s\n", what, bge_last_error());
        return false;
    }

    // Stable dense indices: an entity keeps its index while it owns a Transform; freed indices are reused.
    bool RefreshTopology(SceneT& scene)
    {
        auto& transforms = scene.GetTransforms();
        bool changed = transforms.size() != live_ || topology_stale_;
        topology_stale_ = false;
        if (!changed) {
            for (auto& kv : transforms) {
                auto it = index_of_.find(kv.first);
                if (it == index_of_.end() || parent_[it->second] != ParentIndex(scene, kv.first)) {
                    changed = true;
                    break;
                }
            }
        }
        if (!changed) return true;

        // drop entities that lost their Transform; remember which index each id had, because the reference reuses
        // EntityIds (Scene.cpp:24-27) and everything keyed by id there (e.g. a trigger's remembered overlaps) is keyed
        // by index here: a re-created id gets its old index back
        for (auto it = index_of_.begin(); it != index_of_.end();) {
            if (!scene.HasTransform(it->first)) {
                has_tf_[it->second] = 0;
                const bool body_lives_on = body_[it->second].exists && scene.GetRigidBody(it->first) && scene.GetCollider(it->first);
                const bool ghost_lives_on = has_trigger_[it->second] && scene.GetTriggerVolume(it->first);
                if (!body_lives_on && body_[it->second].exists) {
                    gone_bodies_.push_back(it->second); // (its components went with the Transform)
                    body_[it->second] = BodyState{};
                }
                if (body_lives_on || ghost_lives_on) {
                    // The entity lost its Transform but keeps its body components: the reference keeps stepping the Bullet body
                    // (EnsureRigidBody returns before it looks at the runtime, PhysicsSystem.cpp:389-393), and a trigger ghost stays where
                    // it was (EnsureTrigger returns early too).  The index stays the entity's — the world keeps body and ghost on
                    // it — until the components go (UploadBodies / UploadTriggers) or the Transform returns.
                    orphan_of_[it->first] = it->second;
                } else {
                    // (the world keeps the body of an index that merely loses its Transform: one whose components went too —
                    //  a destroyed entity — is taken out explicitly, while its slot still exists)
                    if (body_[it->second].exists) gone_bodies_.push_back(it->second);
                    ids_[it->second] = 0;
                    has_trigger_[it->second] = 0;
                    body_[it->second] = BodyState{};
                    free_.push_back(it->second);
                    is_free_[it->second] = 1;
                    retired_[it->first] = it->second;
                }
                it = index_of_.erase(it);
            } else {
                ++it;
            }
        }
        if (!gone_bodies_.empty()) {
            const std::vector<uint8_t> none(gone_bodies_.size(), BGE_BODY_NONE);
            if (bge_world_upload_bodies_indexed(world_, gone_bodies_.size(), gone_bodies_.data(), none.data(), nullptr, nullptr, nullptr, nullptr,
                                                nullptr) != BGE_OK) {
                return Log("bge_world_upload_bodies_indexed (bodies of entities that are gone)");
            }
            gone_bodies_.clear();
        }
        // first the ids that come back, so that nobody else takes their index
        for (auto& kv : transforms) {
            if (index_of_.count(kv.first)) continue;
            auto o = orphan_of_.find(kv.first);
            if (o != orphan_of_.end()) { // the Transform returns to a body that lived on without one: same index, body state kept
                const uint32_t i = o->second;
                has_tf_[i] = 1;
                written_[i] = 0;
                index_of_[kv.first] = i;
                orphan_of_.erase(o);
                continue;
            }
            auto r = retired_.find(kv.first);
            if (r == retired_.end() || !is_free_[r->second]) continue;
            const uint32_t i = r->second;
            is_free_[i] = 0;
            ids_[i] = kv.first;
            last_ids_[i] = kv.first;
            has_tf_[i] = 1;
            written_[i] = 0;
            index_of_[kv.first] = i;
            fresh_.push_back(i);
        }
        // New entities take their indices in ASCENDING ENTITY ID (the store is an unordered_map): where the device orders entities — the
        // lowest four obstacles of a body, body A of a pair of Dynamic boxes, the bodies and manifolds of a simulation island — it
        // orders by index, the oracle by entity id; for entities that first appear together (a loaded scene) the two agree.  A reused
        // index (free list) can break the agreement: results stay deterministic, the solver's row order is then another one.
        new_ids_.clear();
        for (auto& kv : transforms) {
            if (!index_of_.count(kv.first)) new_ids_.push_back(kv.first);
        }
        std::sort(new_ids_.begin(), new_ids_.end());
        for (const Id new_id : new_ids_) {
            struct { Id first; } kv{new_id};
            uint32_t i;
            while (!free_.empty() && !is_free_[free_.back()]) free_.pop_back(); // taken back above
            if (!free_.empty()) {
                i = free_.back();
                free_.pop_back();
                is_free_[i] = 0;
            } else {
                i = static_cast<uint32_t>(ids_.size());
                ids_.push_back(0);
                last_ids_.push_back(0);
                has_tf_.push_back(0);
                parent_.push_back(BGE_NO_PARENT);
                body_.emplace_back();
                is_free_.push_back(0);
                has_trigger_.push_back(0);
                written_.push_back(0);
                last_pose_.resize(last_pose_.size() + 6, 0.0f);
                last_scale_.resize(last_scale_.size() + 3, 0.0f);
            }
            ids_[i] = kv.first;
            last_ids_[i] = kv.first; // survives the entity: an Exit event may name a body that was destroyed
            has_tf_[i] = 1;
            written_[i] = 0;
            index_of_[kv.first] = i;
            fresh_.push_back(i);
        }
        for (auto& kv : transforms) parent_[index_of_[kv.first]] = ParentIndex(scene, kv.first);
        // (an index whose entity lost its Transform keeps the parent it had: to the device that entity's place in the
        //  hierarchy has not changed, so nothing below it is taken for re-parented)
        live_ = transforms.size();
        if (bge_world_set_topology(world_, ids_.size(), parent_.data(), has_tf_.data()) != BGE_OK) return Log("bge_world_set_topology");
        // a reused index must not inherit the previous owner's device state: force a full upload
        for (uint32_t i : fresh_) {
            if (auto* t = scene.GetTransform(ids_[i])) t->dirty = true;
        }
        fresh_.clear();
        return true;
    }

    // An index kept for an entity without a Transform (its body and / or its trigger ghost live on) is given up when neither is left.
    void RetireIfNothingLeft(uint32_t i)
    {
        if (has_tf_[i] || ids_[i] == 0 || body_[i].exists || has_trigger_[i]) return;
        orphan_of_.erase(ids_[i]);
        retired_[ids_[i]] = i;
        ids_[i] = 0;
        free_.push_back(i);
        is_free_[i] = 1;
        topology_stale_ = true;
    }

    uint32_t ParentIndex(SceneT& scene, Id id)
    {
        const Id p = scene.GetParent(id);
        if (p == 0) return BGE_NO_PARENT;
        if (!scene.HasTransform(p)) {
            // The child is a root (Scene.cpp:528).  If the parent entity owned a Transform before, the device is told the SAME
            // parent index (has_transform = 0 there) for as long as nobody else has taken it: bge_world_set_topology then sees
            // an unchanged parent entity that lost its Transform — it neither marks the child dirty nor recomputes its world
            // matrix, as the reference does not (Scene::RemoveTransform marks nobody) — instead of a changed parent.
            auto o = orphan_of_.find(p);
            if (o != orphan_of_.end()) return o->second; // (its body still holds the index)
            auto r = retired_.find(p);
            return (r != retired_.end() && is_free_[r->second]) ? r->second : BGE_NO_PARENT;
        }
        auto it = index_of_.find(p);
        return it == index_of_.end() ? 0xfffffffeu /* not indexed yet: forces a refresh */ : it->second;
    }

    bool UploadDirtyTransforms(SceneT& scene)
    {
        index_list_.clear();
        stage_.clear();
        for (auto& kv : scene.GetTransforms()) {
            auto& t = kv.second;
            if (!t.dirty) continue;
            const uint32_t i = index_of_[kv.first];
            if (written_[i] && std::memcmp(&t.position, &last_pose_[6 * static_cast<size_t>(i)], 12) == 0 &&
                std::memcmp(&t.rotationEuler, &last_pose_[6 * static_cast<size_t>(i) + 3], 12) == 0 &&
                std::memcmp(&t.scale, &last_scale_[3 * static_cast<size_t>(i)], 12) == 0) {
                continue; // dirty only because the physics write-back marked it: the device already has these values
            }
            // (the scale is compared too: a scale edit + MarkDirty between PhysicsSystem::Update and
            //  TransformSystem::Update must reach the device — the reference recomputes `local` from it in that frame)
            std::memcpy(&last_scale_[3 * static_cast<size_t>(i)], &t.scale, 12);
            index_list_.push_back(i);
            const float* p = &t.position.x;
            stage_.insert(stage_.end(), p, p + 9); // position, rotationEuler, scale are contiguous (Transform.h:14-16)
        }
        const size_t m = index_list_.size();
        if (!m) return true;
        // repack [pos|euler|scale] rows into three arrays
        repack_.resize(m * 9);
        for (size_t k = 0; k < m; ++k) {
            std::memcpy(&repack_[3 * k], &stage_[9 * k], 12);
            std::memcpy(&repack_[3 * m + 3 * k], &stage_[9 * k + 3], 12);
            std::memcpy(&repack_[6 * m + 3 * k], &stage_[9 * k + 6], 12);
        }
        if (bge_world_upload_trs_indexed(world_, m, index_list_.data(), repack_.data(), repack_.data() + 3 * m,
                                         repack_.data() + 6 * m) != BGE_OK) {
            return Log("bge_world_upload_trs_indexed");
        }
        for (uint32_t i : index_list_) written_[i] = 0;
        return true;
    }

    // EnsureRigidBody / prune loops (PhysicsSystem.cpp:1222-1260, 382-499): a body exists where RigidBody, Collider
    // and Transform all do; it is (re)created when RigidBody.dirty or Collider.dirty is set.
    bool UploadBodies(SceneT& scene)
    {
        index_list_.clear();
        b_type_.clear(); b_mass_.clear(); b_shape_.clear(); b_size_.clear(); b_layer_.clear(); b_mask_.clear(); b_friction_.clear(); b_restitution_.clear();
        seen_.assign(ids_.size(), 0);
        for (auto& kv : scene.GetRigidBodies()) {
            auto it = index_of_.find(kv.first);
            auto* col = scene.GetCollider(kv.first);
            if (it == index_of_.end()) {
                // no Transform: nothing is (re)created (its dirty flags wait); a body that lives on without one stays as it is
                auto o = orphan_of_.find(kv.first);
                if (o != orphan_of_.end() && col) seen_[o->second] = 1;
                continue;
            }
            if (!col) continue;
            const uint32_t i = it->second;
            seen_[i] = 1;
            auto& rb = kv.second;
            if (body_[i].exists && !rb.dirty && !col->dirty) continue;
            index_list_.push_back(i);
            b_type_.push_back(static_cast<uint8_t>(static_cast<int>(rb.type)));
            b_mass_.push_back(rb.mass);
            b_shape_.push_back(static_cast<uint8_t>(static_cast<int>(col->shape)));
            b_size_.push_back(col->size.x); b_size_.push_back(col->size.y); b_size_.push_back(col->size.z);
            b_layer_.push_back(rb.layer);
            b_mask_.push_back(rb.mask);
            b_friction_.push_back(rb.friction); // info.m_friction = body.friction (PhysicsSystem.cpp:437)
            b_restitution_.push_back(rb.restitution); // info.m_restitution = body.restitution (:438)
            body_[i].exists = true;
            rb.dirty = false;   // PhysicsSystem.cpp:476
            col->dirty = false; // PhysicsSystem.cpp:403
        }
        for (uint32_t i = 0; i < ids_.size(); ++i) {
            if (body_[i].exists && !seen_[i]) { // component removed: RemoveRigidBody (PhysicsSystem.cpp:1230-1233)
                index_list_.push_back(i);
                b_type_.push_back(BGE_BODY_NONE);
                b_mass_.push_back(0.0f);
                b_shape_.push_back(0);
                b_size_.insert(b_size_.end(), {0.5f, 0.5f, 0.5f});
                b_layer_.push_back(1);
                b_mask_.push_back(0xffffffffu);
                b_friction_.push_back(0.5f);
                b_restitution_.push_back(0.0f);
                body_[i].exists = false;
                RetireIfNothingLeft(i);
            }
        }
        if (index_list_.empty()) return true;
        if (bge_world_upload_bodies_indexed(world_, index_list_.size(), index_list_.data(), b_type_.data(), b_mass_.data(),
                                            b_shape_.data(), b_size_.data(), b_layer_.data(), b_mask_.data()) != BGE_OK) {
            return Log("bge_world_upload_bodies_indexed");
        }
        if (bge_world_upload_friction_indexed(world_, index_list_.size(), index_list_.data(), b_friction_.data()) != BGE_OK) {
            return Log("bge_world_upload_friction_indexed");
        }
        if (bge_world_upload_restitution_indexed(world_, index_list_.size(), index_list_.data(), b_restitution_.data()) != BGE_OK) {
            return Log("bge_world_upload_restitution_indexed");
        }
        return true;
    }

    bge_world* world_ = nullptr;
    std::vector<Id> ids_;                       // dense index -> EntityId (0 = free)
    std::vector<Id> last_ids_;                  // dense index -> the id it last belonged to
    std::unordered_map<Id, uint32_t> index_of_;
    std::vector<uint32_t> parent_, free_, fresh_, index_list_, gone_bodies_;
    std::vector<Id> new_ids_;
    std::vector<uint8_t> has_tf_, written_, limbo_, seen_, is_free_, has_trigger_, seen_trigger_;
    std::unordered_map<Id, uint32_t> retired_;  // last index of ids that lost their Transform
    std::unordered_map<Id, uint32_t> orphan_of_; // ids without a Transform whose body lives on, on the index they had
    bool topology_stale_ = false;               // an index was given up outside RefreshTopology
    std::vector<BodyState> body_;
    std::vector<float> last_pose_;              // position + euler the physics write-back stored (6 floats per index)
    std::vector<float> last_scale_;             // scale as last uploaded (3 floats per index)
    std::vector<float> stage_, repack_;
    // Destination of the downloads: page-locked (bge_host_alloc), so the copies run at the PCIe rate instead of the
    // pageable ~10 GB/s; grow-only.
    struct PinnedFloats {
        float* p = nullptr;
        size_t cap = 0;
        ~PinnedFloats() { bge_host_free(p); }
        PinnedFloats() = default;
        PinnedFloats(const PinnedFloats&) = delete;
        PinnedFloats& operator=(const PinnedFloats&) = delete;
        float* resize(size_t n)
        {
            if (n > cap) {
                bge_host_free(p);
                p = nullptr;
                cap = 0;
                void* q = nullptr;
                const size_t want = n + n / 4;
                if (bge_host_alloc(want * sizeof(float), &q) == BGE_OK) {
                    p = static_cast<float*>(q);
                    cap = want;
                }
            }
            return p;
        }
        float& operator[](size_t i) { return p[i]; }
        float* data() { return p; }
    } down_;
    std::vector<uint8_t> b_type_, b_shape_;
    std::vector<float> b_mass_, b_size_, b_friction_, b_restitution_;
    std::vector<uint32_t> b_layer_, b_mask_;
    size_t live_ = 0;
    std::vector<uint32_t> t_entity_, t_layer_, t_mask_, t_signature_;
    std::vector<uint8_t> t_shape_, t_oneshot_, t_active_;
    std::vector<float> t_size_;
    std::vector<uint32_t> t_order_; // EntityIds of the scene's TriggerVolumes, ascending
    std::vector<bge_trigger_event> raw_events_;
    std::vector<GpuTriggerEvent> trigger_events_;
};

// One mirror per Scene object, found by address (TransformSystem::Update is a static function without state,
// src/ecs/TransformSystem.h:5-9).  OnSceneReloaded drops it (the reference move-assigns the whole Scene on
// reload, src/scene/SceneLoader.cpp:742).
template <class SceneT> class GpuMirrors {
public:
    static GpuSceneMirror<SceneT>& Of(SceneT& scene)
    {
        auto& slot = Table()[&scene];
        if (!slot) slot = std::make_unique<GpuSceneMirror<SceneT>>();
        return *slot;
    }
    static void Drop(SceneT& scene) { Table().erase(&scene); }

private:
    static std::unordered_map<const void*, std::unique_ptr<GpuSceneMirror<SceneT>>>& Table()
    {
        static std::unordered_map<const void*, std::unique_ptr<GpuSceneMirror<SceneT>>> t;
        return t;
    }
};

template <class SceneT> class GpuTransformSystem {
public:
    static void Update(SceneT& scene) { GpuMirrors<SceneT>::Of(scene).UpdateTransforms(scene); }
};

template <class SceneT> class GpuPhysicsSystem {
public:
    // ---- PhysicsSystem's public surface for this path (src/physics/PhysicsSystem.h:43-48, 77)
    void SetConfigPath(std::filesystem::path path)
    {
        configPath_ = std::move(path);
        hasLastWriteTime_ = false;
    }
    // EnsureWorld (PhysicsSystem.cpp:108-120): the device world of a scene is created on its first Update; here the
    // GPU is probed once so that a missing device is reported at start-up, as a failed Bullet set-up would be
    void Initialize()
    {
        GpuSceneMirror<SceneT> probe;
        if (!probe.ok()) std::fprintf(stderr, "[GPU] Initialize: no usable device — %s\n", bge_last_error());
    }
    void OnSceneReloaded(SceneT& scene) { GpuMirrors<SceneT>::Drop(scene); } // the mirror (and its clock) is rebuilt by the next Update
    // mtime poll of the config file, every frame (PhysicsSystem.cpp:216-240)
    bool ReloadConfigIfNeeded(SceneT& scene)
    {
        if (configPath_.empty()) return false;
        std::error_code ec;
        const auto currentTime = std::filesystem::last_write_time(configPath_, ec);
        if (ec) return false;
        if (!hasLastWriteTime_ || currentTime != lastWriteTime_) {
            lastWriteTime_ = currentTime;
            hasLastWriteTime_ = true;
            config_ = LoadConfigFromDisk();
            (void)scene; // ApplyConfig's scene work is the character controllers' (out of scope); gravity reaches the world on the next Update
            return true;
        }
        return false;
    }
    // rigid-body slice of PhysicsSystem::Update(Scene&, const Camera&, const InputSystem&, double) (PhysicsSystem.cpp:1208-1328);
    // camera and input feed the character controller, which is not part of this path
    template <class CameraT, class InputT> void Update(SceneT& scene, const CameraT&, const InputT&, double dt) { Step(scene, dt); }
    void Update(SceneT& scene, double dt) { Step(scene, dt); }
    // PhysicsSystem.cpp:1330-1341
    void LogStats() const
    {
        std::printf("[Physics] bodies=%d characters=%zu stepTime=%.4fms substeps=%d fixedStep=%.4f actualDt=%.4f\n", lastBodies_,
                    static_cast<size_t>(0), lastStepDurationMs_, lastStepSubsteps_, config_.fixedStep, lastStepDt_);
    }
    double GetFixedStep() const { return config_.fixedStep; }

    // ---- additions
    void SetGravity(float g) { config_.gravity = g; }
    // Bullet's own orientation scheme for every Dynamic body (include/bge_world.h, BGE_TICK_BULLET_BASIS); choose before
    // the first Update of a scene
    void SetBulletBasis(bool on) { bulletBasis_ = on; }
    // the static ground plane every reference world has (on by default); off = free bodies
    void SetGroundPlane(bool on) { groundPlane_ = on; }
    int LastSubSteps() const { return lastStepSubsteps_; }
    // trigger events of the last Update (publish them on the engine's EventBus, src/core/EventBus.h)
    const std::vector<GpuTriggerEvent>& TriggerEvents(SceneT& scene) const { return GpuMirrors<SceneT>::Of(scene).TriggerEvents(); }

    // PhysicsSystem::Config (src/physics/PhysicsSystem.h:85-95); the character fields are read and kept, nothing here uses them
    struct Config {
        float gravity = -9.81f;
        float fixedStep = 1.0f / 120.0f;
        float stepHeight = 0.35f;
        float maxSlopeDeg = 50.0f;
        float capsuleHeight = 1.7f;
        float capsuleRadius = 0.35f;
        float walkSpeed = 3.5f;
        float jumpImpulse = 5.0f;
    };
    const Config& GetConfig() const { return config_; }

private:
    // LoadConfigFromDisk (PhysicsSystem.cpp:242-282): missing keys keep the current value, an unreadable or malformed
    // file keeps the whole current config, a non-positive fixedStep becomes 1/120
    Config LoadConfigFromDisk() const
    {
        Config cfg = config_;
        std::ifstream file(configPath_);
        if (!file.is_open()) {
            std::fprintf(stderr, "[Physics] Failed to open config: %s\n", configPath_.string().c_str());
            return cfg;
        }
        std::stringstream ss;
        ss << file.rdbuf();
        const std::string text = ss.str();
        json::Value data;
        std::string err;
        json::Parser parser(text);
        if (!parser.parse(data, &err) || data.kind != json::Value::Object) {
            std::fprintf(stderr, "[Physics] Failed to parse config: %s\n", err.c_str());
            return cfg;
        }
        cfg.gravity = detail::read_float(data, "gravity", cfg.gravity);
        cfg.fixedStep = detail::read_float(data, "fixedStep", cfg.fixedStep);
        cfg.stepHeight = detail::read_float(data, "stepHeight", cfg.stepHeight);
        cfg.maxSlopeDeg = detail::read_float(data, "maxSlopeDeg", cfg.maxSlopeDeg);
        cfg.walkSpeed = detail::read_float(data, "walkSpeed", cfg.walkSpeed);
        cfg.jumpImpulse = detail::read_float(data, "jumpImpulse", cfg.jumpImpulse);
        if (const json::Value* capsule = data.find("capsule"); capsule && capsule->kind == json::Value::Object) {
            cfg.capsuleHeight = detail::read_float(*capsule, "height", cfg.capsuleHeight);
            cfg.capsuleRadius = detail::read_float(*capsule, "radius", cfg.capsuleRadius);
        }
        if (!(cfg.fixedStep > 0.0f)) cfg.fixedStep = 1.0f / 120.0f;
        return cfg;
    }

    void Step(SceneT& scene, double dt)
    {
        auto& m = GpuMirrors<SceneT>::Of(scene);
        m.gravity[0] = 0.0f;
        m.gravity[1] = config_.gravity; // m_world->setGravity(btVector3(0, m_config.gravity, 0)), PhysicsSystem.cpp:130, 292
        m.gravity[2] = 0.0f;
        m.bullet_basis = bulletBasis_;
        if (m.ground_plane_state != static_cast<int>(groundPlane_)) {
            m.SetGroundPlane(groundPlane_);
            m.ground_plane_state = static_cast<int>(groundPlane_);
        }
        m.max_sub_steps = 4;                                        // PhysicsSystem.cpp:863
        m.fixed_step = std::max(config_.fixedStep, 1.0f / 240.0f); // kMinStep, PhysicsSystem.cpp:33, 855
        const auto start = std::chrono::high_resolution_clock::now();
        m.UpdatePhysics(scene, dt);
        const auto end = std::chrono::high_resolution_clock::now();
        lastStepDurationMs_ = std::chrono::duration<double, std::milli>(end - start).count(); // includes the pose download
        lastStepDt_ = dt;
        lastStepSubsteps_ = m.last_sub_steps;
        lastBodies_ = m.CollisionObjectCount();
    }

    std::filesystem::path configPath_;
    std::filesystem::file_time_type lastWriteTime_{};
    bool hasLastWriteTime_ = false;
    Config config_{};
    bool bulletBasis_ = false;
    bool groundPlane_ = true;
    double lastStepDurationMs_ = 0.0, lastStepDt_ = 0.0;
    int lastStepSubsteps_ = 0, lastBodies_ = 0;
};

} // namespace bge
