// tests/cpp/reference_shapes_mock.hpp — DECLARATIONS ONLY: types with the member signatures the reference's callers and
// this repository's adapter see on the hot path, spelled as the reference spells them (global namespace; citations are
// to /root/reference).  Nothing here is defined or executed: tests/cpp/compile_reference_shapes.cpp is compiled to an
// object file and never linked, which is enough to prove that the templates in bge/gpu_systems.hpp and
// bge/scene_json.hpp instantiate against these exact shapes ("duck-typing is unverified").
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

using EntityId = uint32_t;                     // src/ecs/Entity.h:4
static constexpr EntityId kInvalidEntity = 0;  // src/ecs/Entity.h:5

struct float3 { float x = 0.0f, y = 0.0f, z = 0.0f; };  // src/ecs/Transform.h:5-10

struct Transform {                             // src/ecs/Transform.h:12-26
    float3 position{0.0f, 0.0f, 0.0f};
    float3 rotationEuler{0.0f, 0.0f, 0.0f};
    float3 scale{1.0f, 1.0f, 1.0f};
    float local[16]{};
    float world[16]{};
    bool dirty = true;
    Transform();
    void MarkDirty();
    void RecalculateLocalMatrix();
    void UpdateWorldMatrix(const float* parentWorld);
};

enum class ColliderShape { Box, Capsule };     // src/ecs/PhysicsComponents.h:7-11
struct Collider { ColliderShape shape = ColliderShape::Box; float3 size{0.5f, 0.5f, 0.5f}; bool dirty = true; };  // :13-19
enum class RigidBodyType { Static, Dynamic, Kinematic };  // :21-26
struct RigidBody {                             // :28-37
    RigidBodyType type = RigidBodyType::Static;
    float mass = 0.0f, friction = 0.5f, restitution = 0.0f;
    uint32_t layer = 1u, mask = 0xffffffffu;
    bool dirty = true;
};
struct TriggerVolume {                         // :39-48
    ColliderShape shape = ColliderShape::Box;
    float3 size{0.5f, 0.5f, 0.5f};
    uint32_t layer = 0u, mask = 0xffffffffu;
    bool oneShot = false, active = true, dirty = true;
};

// the four accessors every component type has (src/ecs/Scene.h:28-56): Add / Get / Get const / Remove
#define REF_COMPONENT_ACCESSORS(T)   \
    T* Add##T(EntityId id);          \
    T* Get##T(EntityId id);          \
    const T* Get##T(EntityId id) const; \
    void Remove##T(EntityId id);

class Scene {                                  // src/ecs/Scene.h:19-109
public:
    Scene() = default;
    EntityId CreateEntity();                   // :24
    void DestroyEntity(EntityId id);           // :25
    bool IsAlive(EntityId id) const;           // :26
    REF_COMPONENT_ACCESSORS(Transform)         // :28-31
    REF_COMPONENT_ACCESSORS(Collider)          // :38-41
    REF_COMPONENT_ACCESSORS(RigidBody)         // :43-46
    REF_COMPONENT_ACCESSORS(TriggerVolume)     // :48-51
    void SetParent(EntityId child, EntityId parent);                   // :58
    EntityId GetParent(EntityId child) const;                          // :59
    const std::vector<EntityId>& GetChildren(EntityId parent) const;   // :60
    size_t GetEntityCount() const;             // :62
    size_t GetTransformCount() const;          // :63
    size_t CountDirtyTransforms() const;       // :66
    const std::unordered_map<EntityId, Transform>& GetTransforms() const;          // :68
    std::unordered_map<EntityId, Transform>& GetTransforms();                      // :69
    const std::unordered_map<EntityId, Collider>& GetColliders() const;            // :72
    std::unordered_map<EntityId, Collider>& GetColliders();                        // :73
    const std::unordered_map<EntityId, RigidBody>& GetRigidBodies() const;         // :74
    std::unordered_map<EntityId, RigidBody>& GetRigidBodies();                     // :75
    const std::unordered_map<EntityId, TriggerVolume>& GetTriggerVolumes() const;  // :76
    std::unordered_map<EntityId, TriggerVolume>& GetTriggerVolumes();              // :77
    void ForEachRootTransform(const std::function<void(EntityId)>& fn) const;      // :85
    void MarkHierarchyDirty(EntityId id);      // :87
    bool HasTransform(EntityId id) const;      // :89
};
#undef REF_COMPONENT_ACCESSORS

class Camera;       // src/camera/Camera.h — only ever passed by const reference (src/physics/PhysicsSystem.h:47)
class InputSystem;  // src/input/InputSystem.h — likewise
