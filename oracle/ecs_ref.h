// oracle/ecs_ref.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the reference's ECS store and transform system, keeping the
// reference's data structures (node-based hash maps of 168-byte AoS Transforms,
// recursive DFS) because this is also "the reference CPU path" that bench.py times
// as cpu_baseline (kind "port", 1 thread — the reference is single-threaded).
//
// Follows:
//   src/ecs/Entity.h:4-5                 EntityId = uint32, 0 invalid
//   src/ecs/Transform.h:12-26            struct Transform (168 B)
//   src/ecs/Transform.cpp:18-36          RecalculateLocalMatrix / UpdateWorldMatrix
//   src/ecs/PhysicsComponents.h:7-37     Collider, RigidBody
//   src/ecs/Scene.cpp:21-41              CreateEntity (LIFO id reuse)
//   src/ecs/Scene.cpp:43-83              DestroyEntity (children orphaned + marked dirty)
//   src/ecs/Scene.cpp:90-102             AddTransform (emplace keeps old value, marks dirty)
//   src/ecs/Scene.cpp:354-393            SetParent (no cycle check, marks subtree dirty)
//   src/ecs/Scene.cpp:395-413            GetParent / GetChildren
//   src/ecs/Scene.cpp:435-446            CountDirtyTransforms
//   src/ecs/Scene.cpp:523-533            ForEachRootTransform (root = no parent or parent lacks Transform)
//   src/ecs/Scene.cpp:535-550            MarkHierarchyDirty
//   src/ecs/TransformSystem.cpp:10-46    UpdateNode / TransformSystem::Update
//
// The reference's transform half does not build in this container (it needs the bx and
// bgfx headers, which are absent); see DESIGN.md "Oracle".
#pragma once

#include <algorithm>
#include <cstdint>
#include <functional>
#include <unordered_map>
#include <vector>

#include "bx_math.h"

namespace orc {

using EntityId = uint32_t;
inline constexpr EntityId kInvalidEntity = 0;

struct Float3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
};

struct RefTransform {
    Float3 position{0.0f, 0.0f, 0.0f};
    Float3 rotationEuler{0.0f, 0.0f, 0.0f};
    Float3 scale{1.0f, 1.0f, 1.0f};
    float local[16];
    float world[16];
    bool dirty = true;

    RefTransform()
    {
        bxm::mtxIdentity(local);
        bxm::mtxIdentity(world);
    }
    void MarkDirty() { dirty = true; }
    void RecalculateLocalMatrix()
    {
        bxm::mtxSRT(local, scale.x, scale.y, scale.z, rotationEuler.x, rotationEuler.y, rotationEuler.z,
                    position.x, position.y, position.z);
    }
    void UpdateWorldMatrix(const float* parentWorld)
    {
        if (parentWorld != nullptr) {
            bxm::mtxMul(world, parentWorld, local); // parent · local — the reference's order
        } else {
            std::memcpy(world, local, sizeof world);
        }
    }
};
static_assert(sizeof(RefTransform) == 168, "layout of src/ecs/Transform.h:12-26");

enum class RefShape : int { Box = 0, Capsule = 1 };
enum class RefBodyType : int { Static = 0, Dynamic = 1, Kinematic = 2 };

struct RefCollider {
    RefShape shape = RefShape::Box;
    Float3 size{0.5f, 0.5f, 0.5f};
    bool dirty = true;
};

struct RefRigidBody {
    RefBodyType type = RefBodyType::Static;
    float mass = 0.0f;
    float friction = 0.5f;
    float restitution = 0.0f;
    uint32_t layer = 1u;
    uint32_t mask = 0xffffffffu;
    bool dirty = true;
};

// src/ecs/PhysicsComponents.h:39-48
struct RefTriggerVolume {
    RefShape shape = RefShape::Box;
    Float3 size{0.5f, 0.5f, 0.5f};
    uint32_t layer = 0u;
    uint32_t mask = 0xffffffffu;
    bool oneShot = false;
    bool active = true;
    bool dirty = true;
};

class RefScene {
public:
    EntityId CreateEntity()
    {
        EntityId id;
        if (!freeIds_.empty()) {
            id = freeIds_.back();
            freeIds_.pop_back();
        } else {
            id = ++nextId_;
            if (id == kInvalidEntity) id = ++nextId_;
        }
        alive_[id] = 0u;
        children_[id];
        return id;
    }

    void DestroyEntity(EntityId id)
    {
        if (!IsAlive(id)) return;
        RemoveTransform(id);
        RemoveTriggerVolume(id);
        RemoveRigidBody(id);
        RemoveCollider(id);
        if (const EntityId parent = GetParent(id); parent != kInvalidEntity) {
            auto sib = children_.find(parent);
            if (sib != children_.end()) {
                auto& v = sib->second;
                v.erase(std::remove(v.begin(), v.end(), id), v.end());
            }
        }
        if (auto kids = children_.find(id); kids != children_.end()) {
            for (EntityId child : kids->second) {
                parents_.erase(child);
                MarkHierarchyDirty(child);
            }
            children_.erase(kids);
        }
        parents_.erase(id);
        alive_.erase(id);
        freeIds_.push_back(id);
    }

    bool IsAlive(EntityId id) const { return alive_.find(id) != alive_.end(); }

    RefTransform* AddTransform(EntityId id)
    {
        if (!IsAlive(id)) return nullptr;
        auto res = transforms_.emplace(id, RefTransform{});
        res.first->second.MarkDirty();
        return &res.first->second;
    }
    RefTransform* GetTransform(EntityId id)
    {
        auto it = transforms_.find(id);
        return it == transforms_.end() ? nullptr : &it->second;
    }
    const RefTransform* GetTransform(EntityId id) const
    {
        auto it = transforms_.find(id);
        return it == transforms_.end() ? nullptr : &it->second;
    }
    void RemoveTransform(EntityId id) { transforms_.erase(id); }
    bool HasTransform(EntityId id) const { return transforms_.find(id) != transforms_.end(); }

    RefCollider* AddCollider(EntityId id)
    {
        if (!IsAlive(id)) return nullptr;
        auto res = colliders_.emplace(id, RefCollider{});
        res.first->second.dirty = true;
        return &res.first->second;
    }
    RefCollider* GetCollider(EntityId id)
    {
        auto it = colliders_.find(id);
        return it == colliders_.end() ? nullptr : &it->second;
    }
    void RemoveCollider(EntityId id) { colliders_.erase(id); }

    RefRigidBody* AddRigidBody(EntityId id)
    {
        if (!IsAlive(id)) return nullptr;
        auto res = rigidBodies_.emplace(id, RefRigidBody{});
        res.first->second.dirty = true;
        return &res.first->second;
    }
    RefRigidBody* GetRigidBody(EntityId id)
    {
        auto it = rigidBodies_.find(id);
        return it == rigidBodies_.end() ? nullptr : &it->second;
    }
    void RemoveRigidBody(EntityId id) { rigidBodies_.erase(id); }

    RefTriggerVolume* AddTriggerVolume(EntityId id)
    {
        if (!IsAlive(id)) return nullptr;
        auto res = triggers_.emplace(id, RefTriggerVolume{});
        res.first->second.dirty = true;
        return &res.first->second;
    }
    RefTriggerVolume* GetTriggerVolume(EntityId id)
    {
        auto it = triggers_.find(id);
        return it == triggers_.end() ? nullptr : &it->second;
    }
    void RemoveTriggerVolume(EntityId id) { triggers_.erase(id); }
    std::unordered_map<EntityId, RefTriggerVolume>& GetTriggerVolumes() { return triggers_; }

    void SetParent(EntityId child, EntityId parent)
    {
        if (!IsAlive(child)) return;
        if (parent != kInvalidEntity && !IsAlive(parent)) return;
        const EntityId current = GetParent(child);
        if (current == parent) return;
        if (current != kInvalidEntity) {
            auto sib = children_.find(current);
            if (sib != children_.end()) {
                auto& v = sib->second;
                v.erase(std::remove(v.begin(), v.end(), child), v.end());
            }
        }
        if (parent != kInvalidEntity) {
            children_[parent].push_back(child);
            parents_[child] = parent;
        } else {
            parents_.erase(child);
        }
        MarkHierarchyDirty(child);
    }
    EntityId GetParent(EntityId child) const
    {
        auto it = parents_.find(child);
        return it == parents_.end() ? kInvalidEntity : it->second;
    }
    const std::vector<EntityId>& GetChildren(EntityId parent) const
    {
        static const std::vector<EntityId> none;
        auto it = children_.find(parent);
        return it == children_.end() ? none : it->second;
    }

    size_t GetEntityCount() const { return alive_.size(); }
    size_t GetTransformCount() const { return transforms_.size(); }
    size_t CountDirtyTransforms() const
    {
        size_t n = 0;
        for (const auto& kv : transforms_) n += kv.second.dirty ? 1 : 0;
        return n;
    }

    std::unordered_map<EntityId, RefTransform>& GetTransforms() { return transforms_; }
    std::unordered_map<EntityId, RefCollider>& GetColliders() { return colliders_; }
    std::unordered_map<EntityId, RefRigidBody>& GetRigidBodies() { return rigidBodies_; }

    void ForEachRootTransform(const std::function<void(EntityId)>& fn) const
    {
        for (const auto& kv : transforms_) {
            const EntityId parent = GetParent(kv.first);
            if (parent == kInvalidEntity || !HasTransform(parent)) fn(kv.first);
        }
    }

    void MarkHierarchyDirty(EntityId id)
    {
        if (RefTransform* t = GetTransform(id)) t->MarkDirty();
        auto it = children_.find(id);
        if (it != children_.end()) {
            for (EntityId child : it->second) MarkHierarchyDirty(child);
        }
    }

private:
    std::unordered_map<EntityId, uint32_t> alive_;
    std::unordered_map<EntityId, RefTransform> transforms_;
    std::unordered_map<EntityId, RefCollider> colliders_;
    std::unordered_map<EntityId, RefRigidBody> rigidBodies_;
    std::unordered_map<EntityId, RefTriggerVolume> triggers_;
    std::unordered_map<EntityId, EntityId> parents_;
    std::unordered_map<EntityId, std::vector<EntityId>> children_;
    std::vector<EntityId> freeIds_;
    EntityId nextId_ = kInvalidEntity;
};

namespace detail {
// TransformSystem.cpp:10-37 — recompute local iff own dirty; world iff own or ancestor dirty.
inline void UpdateNode(RefScene& scene, EntityId entity, const float* parentWorld, bool parentDirty)
{
    RefTransform* t = scene.GetTransform(entity);
    if (t == nullptr) return; // a node without a Transform ends the recursion
    const bool localDirty = t->dirty;
    if (localDirty) t->RecalculateLocalMatrix();
    const bool worldDirty = localDirty || parentDirty;
    if (worldDirty) t->UpdateWorldMatrix(parentWorld);
    t->dirty = false;
    for (EntityId child : scene.GetChildren(entity)) {
        UpdateNode(scene, child, t->world, worldDirty);
    }
}
} // namespace detail

// TransformSystem::Update(Scene&)   (TransformSystem.cpp:40-46)
inline void RefTransformSystemUpdate(RefScene& scene)
{
    scene.ForEachRootTransform([&scene](EntityId e) { detail::UpdateNode(scene, e, nullptr, false); });
}

} // namespace orc
