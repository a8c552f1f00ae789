// bge/scene.hpp — a small ECS store with the accessor surface of the reference's Scene
// (src/ecs/Scene.h:19-109) for the components the world tick touches.  It exists so that the adapter in
// gpu_systems.hpp can be built and tested without the reference's sources; the adapter itself is a template
// and works on the reference's own `Scene` unchanged.
//
// Same observable behaviour as the reference for the members below (ids start at 1 and are reused LIFO,
// AddTransform on an existing Transform keeps its value and marks it dirty, SetParent marks the subtree
// dirty, DestroyEntity orphans children and marks them dirty, component pointers stay valid until that
// component is erased) — with one deliberate difference: MarkHierarchyDirty is iterative and visits each
// node once, so closing a parent cycle does not overflow the stack as the reference does.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace bge {

using EntityId = uint32_t;
static constexpr EntityId kInvalidEntity = 0;

struct float3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
};

// Field-for-field the reference's Transform (src/ecs/Transform.h:12-26); `local` is kept for layout
// compatibility but the GPU path never materialises it.
struct Transform {
    float3 position{0.0f, 0.0f, 0.0f};
    float3 rotationEuler{0.0f, 0.0f, 0.0f};
    float3 scale{1.0f, 1.0f, 1.0f};
    float local[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float world[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    bool dirty = true;
    void MarkDirty() { dirty = true; }
};
static_assert(sizeof(Transform) == 168, "same layout as the reference's Transform");

enum class ColliderShape { Box, Capsule };
struct Collider {
    ColliderShape shape = ColliderShape::Box;
    float3 size{0.5f, 0.5f, 0.5f};
    bool dirty = true;
};
enum class RigidBodyType { Static, Dynamic, Kinematic };
struct RigidBody {
    RigidBodyType type = RigidBodyType::Static;
    float mass = 0.0f;
    float friction = 0.5f;
    float restitution = 0.0f;
    uint32_t layer = 1u;
    uint32_t mask = 0xffffffffu;
    bool dirty = true;
};

// src/ecs/PhysicsComponents.h:39-48
struct TriggerVolume {
    ColliderShape shape = ColliderShape::Box;
    float3 size{0.5f, 0.5f, 0.5f};
    uint32_t layer = 0u;
    uint32_t mask = 0xffffffffu;
    bool oneShot = false;
    bool active = true;
    bool dirty = true;
};

class Scene {
public:
    EntityId CreateEntity()
    {
        EntityId id;
        if (!free_.empty()) {
            id = free_.back();
            free_.pop_back();
        } else {
            id = ++next_;
            if (id == kInvalidEntity) id = ++next_;
        }
        alive_.insert(id);
        return id;
    }
    void DestroyEntity(EntityId id)
    {
        if (!IsAlive(id)) return;
        transforms_.erase(id);
        triggers_.erase(id);
        rigidBodies_.erase(id);
        colliders_.erase(id);
        Unlink(id);
        auto kids = children_.find(id);
        if (kids != children_.end()) {
            const std::vector<EntityId> orphaned = std::move(kids->second);
            children_.erase(kids);
            for (EntityId c : orphaned) {
                parents_.erase(c);
                MarkHierarchyDirty(c);
            }
        }
        alive_.erase(id);
        free_.push_back(id);
    }
    bool IsAlive(EntityId id) const { return alive_.count(id) != 0; }

    Transform* AddTransform(EntityId id) { return Add(transforms_, id); }
    Transform* GetTransform(EntityId id) { return Get(transforms_, id); }
    const Transform* GetTransform(EntityId id) const { return Get(transforms_, id); }
    void RemoveTransform(EntityId id) { transforms_.erase(id); }
    bool HasTransform(EntityId id) const { return transforms_.count(id) != 0; }

    Collider* AddCollider(EntityId id) { return Add(colliders_, id); }
    Collider* GetCollider(EntityId id) { return Get(colliders_, id); }
    void RemoveCollider(EntityId id) { colliders_.erase(id); }

    RigidBody* AddRigidBody(EntityId id) { return Add(rigidBodies_, id); }
    RigidBody* GetRigidBody(EntityId id) { return Get(rigidBodies_, id); }
    void RemoveRigidBody(EntityId id) { rigidBodies_.erase(id); }

    TriggerVolume* AddTriggerVolume(EntityId id) { return Add(triggers_, id); }
    TriggerVolume* GetTriggerVolume(EntityId id) { return Get(triggers_, id); }
    void RemoveTriggerVolume(EntityId id) { triggers_.erase(id); }
    std::unordered_map<EntityId, TriggerVolume>& GetTriggerVolumes() { return triggers_; }

    void SetParent(EntityId child, EntityId parent)
    {
        if (!IsAlive(child) || (parent != kInvalidEntity && !IsAlive(parent))) return;
        if (GetParent(child) == parent) return;
        Unlink(child);
        if (parent != kInvalidEntity) {
            children_[parent].push_back(child);
            parents_[child] = parent;
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
        for (const auto& kv : transforms_) n += kv.second.dirty;
        return n;
    }
    std::unordered_map<EntityId, Transform>& GetTransforms() { return transforms_; }
    const std::unordered_map<EntityId, Transform>& GetTransforms() const { return transforms_; }
    std::unordered_map<EntityId, Collider>& GetColliders() { return colliders_; }
    std::unordered_map<EntityId, RigidBody>& GetRigidBodies() { return rigidBodies_; }

    void ForEachRootTransform(const std::function<void(EntityId)>& fn) const
    {
        for (const auto& kv : transforms_) {
            const EntityId p = GetParent(kv.first);
            if (p == kInvalidEntity || !HasTransform(p)) fn(kv.first);
        }
    }
    void MarkHierarchyDirty(EntityId id)
    {
        std::vector<EntityId> stack{id};
        std::unordered_set<EntityId> seen;
        while (!stack.empty()) {
            const EntityId e = stack.back();
            stack.pop_back();
            if (!seen.insert(e).second) continue;
            if (Transform* t = GetTransform(e)) t->MarkDirty();
            for (EntityId c : GetChildren(e)) stack.push_back(c);
        }
    }

private:
    template <class M> static typename M::mapped_type* Get(M& m, EntityId id)
    {
        auto it = m.find(id);
        return it == m.end() ? nullptr : &it->second;
    }
    template <class M> static const typename M::mapped_type* Get(const M& m, EntityId id)
    {
        auto it = m.find(id);
        return it == m.end() ? nullptr : &it->second;
    }
    template <class M> typename M::mapped_type* Add(M& m, EntityId id)
    {
        if (!IsAlive(id)) return nullptr;
        auto& c = m[id]; // keeps an existing component's value
        c.dirty = true;
        return &c;
    }
    void Unlink(EntityId child)
    {
        auto p = parents_.find(child);
        if (p == parents_.end()) return;
        auto sib = children_.find(p->second);
        if (sib != children_.end()) {
            auto& v = sib->second;
            v.erase(std::remove(v.begin(), v.end(), child), v.end());
        }
        parents_.erase(p);
    }

    std::unordered_set<EntityId> alive_;
    std::unordered_map<EntityId, Transform> transforms_;
    std::unordered_map<EntityId, Collider> colliders_;
    std::unordered_map<EntityId, RigidBody> rigidBodies_;
    std::unordered_map<EntityId, TriggerVolume> triggers_;
    std::unordered_map<EntityId, EntityId> parents_;
    std::unordered_map<EntityId, std::vector<EntityId>> children_;
    std::vector<EntityId> free_;
    EntityId next_ = kInvalidEntity;
};

} // namespace bge
