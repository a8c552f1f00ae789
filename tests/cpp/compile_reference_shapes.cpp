// tests/cpp/compile_reference_shapes.cpp — compile-only (g++ -c, never linked): the adapter and the scene ingest,
// instantiated on types that have exactly the reference's member signatures (reference_shapes_mock.hpp), through exactly
// the calls the reference's Application makes (src/core/Application.cpp:84-86, 256, 283-285, 324-326; Renderer.cpp:606).
#include <filesystem>
#include <string>

#include "reference_shapes_mock.hpp"

#include "../../banggameengine_amd/host/bge/gpu_systems.hpp"
#include "../../banggameengine_amd/host/bge/scene_json.hpp"

struct ApplicationShape {
    Scene m_scene;
    bge::GpuPhysicsSystem<Scene> m_physics; // in place of `PhysicsSystem m_physics;` (src/core/Application.h:41)
    double m_fixedDt = 1.0 / 120.0;
    size_t m_lastDirtyBefore = 0, m_lastDirtyAfter = 0;

    void Init()
    {
        m_physics.SetConfigPath(std::filesystem::path("assets/config/physics.json"));
        m_physics.Initialize(); // void Initialize()
    }
    void Frame(const Camera& camera, const InputSystem& input)
    {
        if (m_physics.ReloadConfigIfNeeded(m_scene)) { // bool ReloadConfigIfNeeded(Scene&)
            m_fixedDt = m_physics.GetFixedStep();      // double GetFixedStep() const
        }
        Update(camera, input, m_fixedDt);
    }
    void Update(const Camera& camera, const InputSystem& input, double dt)
    {
        m_physics.Update(m_scene, camera, input, dt); // void Update(Scene&, const Camera&, const InputSystem&, double)
        m_lastDirtyBefore = m_scene.CountDirtyTransforms();
        bge::GpuTransformSystem<Scene>::Update(m_scene); // static void Update(Scene&)
        m_lastDirtyAfter = m_scene.CountDirtyTransforms();
        for (const bge::GpuTriggerEvent& e : m_physics.TriggerEvents(m_scene)) (void)e;
    }
    void ReloadScene(const std::string& text)
    {
        std::string error;
        (void)bge::LoadSceneFromJsonText(text, m_scene, &error); // needs AddCollider / AddRigidBody / AddTriggerVolume / SetParent
        bge::GpuTransformSystem<Scene>::Update(m_scene);
        m_physics.OnSceneReloaded(m_scene);
        m_physics.ReloadConfigIfNeeded(m_scene);
        m_fixedDt = m_physics.GetFixedStep();
        m_physics.LogStats();
    }
    void Resident(const std::vector<EntityId>& visible)
    {
        auto& mirror = bge::GpuMirrors<Scene>::Of(m_scene);
        mirror.resident = true;
        (void)mirror.FetchWorld(m_scene, visible);
    }
};

// force the member templates out
void instantiate(ApplicationShape& app, const Camera& c, const InputSystem& i)
{
    app.Init();
    app.Frame(c, i);
    app.ReloadScene("{}");
    app.Resident({});
}
static_assert(sizeof(Transform) == 168, "the mock has the reference's Transform layout (src/ecs/Transform.h:12-26)");
