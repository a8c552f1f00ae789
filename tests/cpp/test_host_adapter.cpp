// tests/cpp/test_host_adapter.cpp — the C++ adapter (bge/gpu_systems.hpp) against the CPU oracle.
//
// The same scripted sequence of scene edits and ticks is applied to
//   * orc::RefScene + orc::RefPhysicsSystem + RefTransformSystemUpdate   (oracle, CPU)
//   * bge::Scene    + bge::GpuPhysicsSystem + bge::GpuTransformSystem    (product, GPU through the C ABI)
// and every Transform (position, rotationEuler, world, dirty) must match bit for bit after every tick.
// Reads like the reference's frame: physics.Update(scene, dt); TransformSystem::Update(scene)
// (src/core/Application.cpp:256, 284).  Needs a GPU; run by tests/test_host_adapter.py.
#include <algorithm>
#include <array>
#include <iterator>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../banggameengine_amd/host/bge/gpu_systems.hpp"
#include "../../banggameengine_amd/host/bge/scene.hpp"
#include "../../oracle/physics_ref.h"
#include "../../oracle/synth.h"

namespace {

int g_failures = 0;
#define CHECK(cond, ...)                                   \
    do {                                                   \
        if (!(cond)) {                                     \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);                      \
            std::printf("\n");                             \
            if (++g_failures > 20) std::exit(1);           \
        }                                                  \
    } while (0)

inline void Put(void* dst, const float* src) { std::memcpy(dst, src, 12); }

struct Pair {
    orc::RefScene ref;
    orc::RefPhysicsSystem refPhysics;
    bge::Scene gpu;
    bge::GpuPhysicsSystem<bge::Scene> gpuPhysics;
    const double dt = static_cast<double>(0.0083333333f);

    uint32_t Create(const float* p, const float* e, const float* s)
    {
        const uint32_t a = ref.CreateEntity();
        const uint32_t b = gpu.CreateEntity();
        CHECK(a == b, "entity ids diverged: %u vs %u", a, b);
        orc::RefTransform* rt = ref.AddTransform(a);
        bge::Transform* gt = gpu.AddTransform(b);
        Put(&rt->position, p); Put(&rt->rotationEuler, e); Put(&rt->scale, s);
        Put(&gt->position, p); Put(&gt->rotationEuler, e); Put(&gt->scale, s);
        return a;
    }
    void Body(uint32_t id, int type, float mass, const float* size, int shape, uint32_t layer, uint32_t mask)
    {
        auto* rc = ref.AddCollider(id);
        auto* gc = gpu.AddCollider(id);
        rc->shape = static_cast<orc::RefShape>(shape);
        gc->shape = static_cast<bge::ColliderShape>(shape);
        Put(&rc->size, size);
        Put(&gc->size, size);
        auto* rb = ref.AddRigidBody(id);
        auto* gb = gpu.AddRigidBody(id);
        rb->type = static_cast<orc::RefBodyType>(type); gb->type = static_cast<bge::RigidBodyType>(type);
        rb->mass = gb->mass = mass;
        rb->layer = gb->layer = layer;
        rb->mask = gb->mask = mask;
        rb->friction = gb->friction = 0.25f + 0.25f * static_cast<float>(id % 5); // RigidBody::friction reaches the contacts
    }
    void Trigger(uint32_t id, int shape, const float* size, uint32_t layer, uint32_t mask, bool oneShot)
    {
        auto* rt = ref.AddTriggerVolume(id);
        auto* gt = gpu.AddTriggerVolume(id);
        rt->shape = static_cast<orc::RefShape>(shape); gt->shape = static_cast<bge::ColliderShape>(shape);
        Put(&rt->size, size); Put(&gt->size, size);
        rt->layer = gt->layer = layer;
        rt->mask = gt->mask = mask;
        rt->oneShot = gt->oneShot = oneShot;
    }
    void CompareTriggerEvents()
    {
        std::vector<std::array<uint32_t, 3>> a, b;
        for (const auto& e : refPhysics.LastTriggerEvents()) a.push_back({static_cast<uint32_t>(e.type), e.trigger, e.other});
        for (const auto& e : gpuPhysics.TriggerEvents(gpu)) b.push_back({static_cast<uint32_t>(e.type), e.trigger, e.other});
        std::sort(a.begin(), a.end());
        std::sort(b.begin(), b.end());
        CHECK(a == b, "tick %d: trigger events differ: %zu (oracle) vs %zu (gpu)", ticks, a.size(), b.size());
        if (a != b) {
            std::vector<std::array<uint32_t, 3>> only_a, only_b;
            std::set_difference(a.begin(), a.end(), b.begin(), b.end(), std::back_inserter(only_a));
            std::set_difference(b.begin(), b.end(), a.begin(), a.end(), std::back_inserter(only_b));
            for (size_t k = 0; k < only_a.size() && k < 6; ++k) std::printf("  only oracle: type %u trigger %u other %u\n", only_a[k][0], only_a[k][1], only_a[k][2]);
            for (size_t k = 0; k < only_b.size() && k < 6; ++k) std::printf("  only gpu:    type %u trigger %u other %u\n", only_b[k][0], only_b[k][1], only_b[k][2]);
        }
        events_seen += a.size();
        for (auto& kv : ref.GetTriggerVolumes()) {
            auto* g = gpu.GetTriggerVolume(kv.first);
            CHECK(g && g->active == kv.second.active, "trigger %u active flag", kv.first);
        }
    }
    size_t events_seen = 0;
    void SetParent(uint32_t c, uint32_t p) { ref.SetParent(c, p); gpu.SetParent(c, p); }
    void Destroy(uint32_t id) { ref.DestroyEntity(id); gpu.DestroyEntity(id); }
    void Move(uint32_t id, const float* p, bool markDirty)
    {
        if (auto* t = ref.GetTransform(id)) { Put(&t->position, p); if (markDirty) t->MarkDirty(); }
        if (auto* t = gpu.GetTransform(id)) { Put(&t->position, p); if (markDirty) t->MarkDirty(); }
    }
    double dt_override = 0.0;      // when non-zero: the dt of the next Tick
    bool useReferenceShape = false; // call Update(Scene&, const Camera&, const InputSystem&, double)
    int ticks = 0;
    struct AnyCamera {} camera;
    struct AnyInput {} input;
    void Tick()
    {
        const double d = dt_override != 0.0 ? dt_override : dt;
        ++ticks;
        refPhysics.Update(ref, d);
        if (useReferenceShape) gpuPhysics.Update(gpu, camera, input, d);
        else gpuPhysics.Update(gpu, d);
        ComparePose("after physics");
        CompareTriggerEvents();
        orc::RefTransformSystemUpdate(ref);
        bge::GpuTransformSystem<bge::Scene>::Update(gpu);
        CompareAll("after transforms");
    }
    void ComparePose(const char* when)
    {
        for (auto& kv : ref.GetTransforms()) {
            const bge::Transform* g = gpu.GetTransform(kv.first);
            CHECK(g != nullptr, "%s: entity %u missing", when, kv.first);
            if (!g) continue;
            CHECK(std::memcmp(&kv.second.position, &g->position, 12) == 0, "%s: position of %u: %g %g %g vs %g %g %g", when,
                  kv.first, kv.second.position.x, kv.second.position.y, kv.second.position.z, g->position.x, g->position.y, g->position.z);
            CHECK(std::memcmp(&kv.second.rotationEuler, &g->rotationEuler, 12) == 0, "%s: euler of %u", when, kv.first);
            CHECK(kv.second.dirty == g->dirty, "%s: dirty of %u: %d vs %d", when, kv.first, kv.second.dirty, g->dirty);
        }
    }
    void CompareAll(const char* when)
    {
        CHECK(ref.GetTransformCount() == gpu.GetTransformCount(), "%s: transform counts", when);
        CHECK(ref.CountDirtyTransforms() == gpu.CountDirtyTransforms(), "%s: dirty counts %zu vs %zu", when,
              ref.CountDirtyTransforms(), gpu.CountDirtyTransforms());
        ComparePose(when);
        for (auto& kv : ref.GetTransforms()) {
            const bge::Transform* g = gpu.GetTransform(kv.first);
            if (!g) continue;
            CHECK(std::memcmp(kv.second.world, g->world, 64) == 0, "%s: world of %u (row3 %g %g %g vs %g %g %g)", when, kv.first,
                  kv.second.world[12], kv.second.world[13], kv.second.world[14], g->world[12], g->world[13], g->world[14]);
        }
    }
};

} // namespace

int main(int argc, char** argv)
{
    {
        bge::GpuSceneMirror<bge::Scene> probe;
        if (!probe.ok()) {
            std::printf("no usable GPU: %s\n", bge_last_error());
            return 77;
        }
    }
    Pair w;
    w.refPhysics.groundPlane = true; // the adapter's worlds have the reference's ground plane, as every PhysicsSystem world does
    w.refPhysics.staticContacts = true; // ... and collide Dynamic boxes with Static / Kinematic ones, as its dispatcher does
    // Application::ReloadScene -> m_physics.ReloadConfigIfNeeded(m_scene); m_fixedDt = m_physics.GetFixedStep()
    // (src/core/Application.cpp:324-326) with the values of the reference's assets/config/physics.json
    CHECK(w.gpuPhysics.GetFixedStep() == static_cast<double>(1.0f / 120.0f) && w.gpuPhysics.GetConfig().gravity == -9.81f, "defaults (PhysicsSystem.h:85-95)");
    CHECK(!w.gpuPhysics.ReloadConfigIfNeeded(w.gpu), "no config path: nothing to reload");
    if (argc > 1) {
        w.gpuPhysics.SetConfigPath(argv[1]);
        CHECK(w.gpuPhysics.ReloadConfigIfNeeded(w.gpu), "first poll of %s must load it", argv[1]);
        CHECK(!w.gpuPhysics.ReloadConfigIfNeeded(w.gpu), "unchanged file: no reload");
        const auto& cfg = w.gpuPhysics.GetConfig();
        CHECK(cfg.gravity == -9.81f && cfg.fixedStep == 0.0083333333f && cfg.maxSlopeDeg == 55.0f && cfg.capsuleHeight == 2.6f &&
                  cfg.capsuleRadius == 0.65f && cfg.walkSpeed == 3.6f && cfg.jumpImpulse == 8.5f && cfg.stepHeight == 0.35f,
              "physics.json fields");
        // the file's 0.0083333333 is one ulp BELOW 1.0f / 120.0f: the tick must be driven with GetFixedStep(), as the reference does
        CHECK(w.gpuPhysics.GetFixedStep() == w.dt && w.gpuPhysics.GetFixedStep() != static_cast<double>(1.0f / 120.0f), "fixedStep from the file");
    } else {
        std::printf("usage: test_host_adapter <physics_config.json>\n");
        return 2;
    }
    std::mt19937 rng(1234);
    const int n = 4000;
    std::vector<uint32_t> ids;
    for (int i = 0; i < n; ++i) {
        float p[3], e[3], s[3];
        orc::synth::trs(0x5EED, i, 0, p, e, s);
        ids.push_back(w.Create(p, e, s));
    }
    // forest: 85 % of the entities hang under an earlier one
    for (int i = 1; i < n; ++i) {
        if (rng() % 100 < 85) w.SetParent(ids[i], ids[i - 1 - rng() % std::min(i, 40)]);
    }
    // bodies: mixed types, a huge static ground, capsules, filtered layers; some on children (local TRS quirk)
    for (int i = 0; i < n; ++i) {
        const int r = rng() % 10;
        if (r < 4) {
            float size[3] = {0.5f, 0.5f, 0.5f};
            const int type = (r == 0) ? 0 : (r == 1 ? 2 : 1);
            w.Body(ids[i], type, 0.5f + (rng() % 8) * 0.25f, size, rng() % 5 == 0, 1u << (rng() % 3), 0xffffffffu);
        }
    }
    // trigger volumes: big enough to catch passers-by; one-shots, a filtered layer, one on a Dynamic body
    w.refPhysics.computeAabbs = true; // the oracle derives the ghost overlaps from the fed body AABBs
    for (int k = 0; k < 25; ++k) {
        float size[3] = {6.0f + (k % 5) * 4.0f, 10.0f, 6.0f + (k % 3) * 5.0f};
        w.Trigger(ids[40 * k + 3], k % 4 == 0, size, k % 3 == 0 ? 0u : 4u, k % 5 == 0 ? 6u : 0xffffffffu, k % 4 == 1);
    }
    for (int k = 0; k < 3; ++k) w.Tick();

    // second Update without changes must be a no-op (Renderer::BeginFrame calls it again, src/render/Renderer.cpp:606)
    bge::GpuTransformSystem<bge::Scene>::Update(w.gpu);
    orc::RefTransformSystemUpdate(w.ref);
    w.CompareAll("idle update");

    // edits between ticks
    for (int k = 0; k < 60; ++k) {                       // re-link subtrees (never under a descendant: ids only grow)
        const int c = 1 + rng() % (n - 1);
        w.SetParent(ids[c], rng() % 4 == 0 ? 0 : ids[rng() % c]);
    }
    for (int k = 0; k < 40; ++k) w.Destroy(ids[100 + 37 * k]);   // orphans children, frees ids
    for (int k = 0; k < 25; ++k) {                        // new entities reuse the freed ids
        float p[3], e[3], s[3];
        orc::synth::trs(0xFEED, k, 0, p, e, s);
        const uint32_t id = w.Create(p, e, s);
        if (k % 2) {
            float size[3] = {0.3f, 0.6f, 0.4f};
            w.Body(id, 1, 2.0f, size, 0, 1, 0xffffffffu);
        }
        if (k % 3 == 0) w.SetParent(id, ids[5]);
    }
    for (int k = 0; k < 50; ++k) {                        // teleports (dirty) and stale edits (not dirty)
        float p[3] = {float(k), 10.0f + k, -float(k)};
        w.Move(ids[2000 + 13 * k], p, k % 2 == 0);
    }
    w.ref.RemoveRigidBody(ids[7]); w.gpu.RemoveRigidBody(ids[7]);
    w.ref.RemoveCollider(ids[9]); w.gpu.RemoveCollider(ids[9]);
    for (int k = 0; k < 4; ++k) w.Tick();

    // body flagged dirty: re-created from its Transform with zero velocity
    for (auto& kv : w.ref.GetRigidBodies()) { if (kv.first % 7 == 0) kv.second.dirty = true; }
    for (auto& kv : w.gpu.GetRigidBodies()) { if (kv.first % 7 == 0) kv.second.dirty = true; }
    for (int k = 0; k < 3; ++k) w.Tick();

    // Transforms removed from live parents, and given back.  Scene::RemoveTransform marks nobody: the children become roots
    // (Scene.cpp:528) but TransformSystem::Update recomputes a node only when it or an ancestor is dirty, so a clean child keeps
    // its parent * local world matrix, and a child with a body is NOT re-posed.  A body on the parent itself lives on without the
    // Transform (the reference keeps stepping it; so does the world, on the index the entity had).
    {
        std::vector<uint32_t> parents;
        for (auto& kv : w.ref.GetTransforms()) {
            if (parents.size() < 24 && kv.first % 5 == 1 && !w.ref.GetChildren(kv.first).empty()) parents.push_back(kv.first);
        }
        CHECK(parents.size() >= 12, "only %zu parents found for the RemoveTransform phase", parents.size());
        size_t clean_children = 0;
        for (uint32_t id : parents) {
            for (uint32_t c : w.ref.GetChildren(id)) {
                const auto* t = w.ref.GetTransform(c);
                clean_children += t && !t->dirty;
            }
            w.ref.RemoveTransform(id); w.gpu.RemoveTransform(id); // (a body on it lives on: the reference keeps stepping it)
        }
        CHECK(clean_children > 0, "no clean child under the parents that lose their Transform");
        for (int k = 0; k < 3; ++k) w.Tick();
        for (size_t k = 0; k < parents.size(); k += 2) {   // half of them come back, somewhere else
            float p[3], e[3], s[3];
            orc::synth::trs(0xABCD, static_cast<uint32_t>(k), 0, p, e, s);
            orc::RefTransform* rt = w.ref.AddTransform(parents[k]);
            bge::Transform* gt = w.gpu.AddTransform(parents[k]);
            Put(&rt->position, p); Put(&rt->rotationEuler, e); Put(&rt->scale, s);
            Put(&gt->position, p); Put(&gt->rotationEuler, e); Put(&gt->scale, s);
        }
        for (int k = 0; k < 3; ++k) w.Tick();
    }

    CHECK(w.events_seen > 0, "the scripted scene produced no trigger events");
    // resident mode: nothing is copied back by Update, FetchWorld brings exactly what is asked for
    {
        auto& mirror = bge::GpuMirrors<bge::Scene>::Of(w.gpu);
        mirror.resident = true;
        std::vector<uint32_t> before_id;
        std::vector<float> before_world;
        for (auto& kv : w.gpu.GetTransforms()) {
            before_id.push_back(kv.first);
            before_world.insert(before_world.end(), kv.second.world, kv.second.world + 16);
        }
        w.refPhysics.Update(w.ref, w.dt);
        w.gpuPhysics.Update(w.gpu, w.dt);
        orc::RefTransformSystemUpdate(w.ref);
        bge::GpuTransformSystem<bge::Scene>::Update(w.gpu);
        CHECK(w.gpu.CountDirtyTransforms() == w.ref.CountDirtyTransforms(), "resident: dirty counts");
        size_t untouched = 0;
        for (size_t k = 0; k < before_id.size(); ++k) {
            untouched += std::memcmp(w.gpu.GetTransform(before_id[k])->world, &before_world[16 * k], 64) == 0;
        }
        CHECK(untouched == before_id.size(), "resident: Update must not copy world matrices back (%zu of %zu untouched)", untouched, before_id.size());
        std::vector<uint32_t> wanted;
        for (size_t k = 0; k < before_id.size(); k += 7) wanted.push_back(before_id[k]);
        CHECK(mirror.FetchWorld(w.gpu, wanted), "FetchWorld");
        for (uint32_t id : wanted) {
            CHECK(std::memcmp(w.gpu.GetTransform(id)->world, w.ref.GetTransform(id)->world, 64) == 0, "resident: fetched world of %u", id);
        }
        mirror.resident = false;
        bge::GpuTransformSystem<bge::Scene>::Update(w.gpu); // nothing dirty: must not disturb anything
        w.refPhysics.Update(w.ref, w.dt);
        w.gpuPhysics.Update(w.gpu, w.dt);
        orc::RefTransformSystemUpdate(w.ref);
        bge::GpuTransformSystem<bge::Scene>::Update(w.gpu);
        w.CompareAll("back in coherent mode");
    }

    // sleeping through the adapter: no gravity, freshly re-created bodies stand still, Bullet's deactivation puts them to
    // sleep after 2 s (240 ticks); a teleport moves a sleeper without waking it; host `dirty` flags stay coherent throughout
    {
        w.gpuPhysics.SetGravity(0.0f);
        w.refPhysics.gravityY = 0.0f;
        for (auto& kv : w.ref.GetRigidBodies()) kv.second.dirty = true;
        for (auto& kv : w.gpu.GetRigidBodies()) kv.second.dirty = true;
        for (int k = 0; k < 246; ++k) {
            if (k == 243) {
                float p[3] = {3.0f, 4.0f, 5.0f};
                w.Move(ids[2000], p, true);
            }
            w.Tick();
        }
        size_t asleep = 0;
        for (auto& kv : w.refPhysics.Runtimes()) asleep += kv.second.activation == orc::kIslandSleeping && kv.second.invMass != 0.0f;
        CHECK(asleep > 100, "only %zu bodies fell asleep in the oracle", asleep);
        w.gpuPhysics.SetGravity(-9.81f);
        w.refPhysics.gravityY = -9.81f;
        for (int k = 0; k < 3; ++k) w.Tick(); // gravity does not touch sleepers
    }

    // Bullet's own orientation scheme through the adapter (BGE_TICK_BULLET_BASIS vs the oracle's kOrientBasis): every body
    // is re-created (a mode is chosen for the life of a body), then falls for 20 ticks; rotationEuler of every Dynamic
    // body is rewritten from the round-tripped basis each tick and must agree bit for bit
    {
        w.gpuPhysics.SetBulletBasis(true);
        w.refPhysics.orientMode = orc::kOrientBasis;
        for (auto& kv : w.ref.GetRigidBodies()) kv.second.dirty = true;
        for (auto& kv : w.gpu.GetRigidBodies()) kv.second.dirty = true;
        for (int k = 0; k < 20; ++k) w.Tick();
    }

    // A scale edit + MarkDirty between PhysicsSystem::Update and TransformSystem::Update: the Transform is dirty
    // anyway (the physics write-back marked it) and position / euler equal what the device holds — the edit must still
    // reach the device, the reference recomputes `local` from the new scale in this very frame
    {
        w.refPhysics.Update(w.ref, w.dt);
        w.gpuPhysics.Update(w.gpu, w.dt);
        int edited = 0;
        for (auto& kv : w.ref.GetRigidBodies()) {
            if (static_cast<int>(kv.second.type) != 1 || kv.first % 3) continue;
            const float s3[3] = {1.25f + 0.01f * (kv.first % 7), 0.75f, 2.0f};
            auto* rt = w.ref.GetTransform(kv.first);
            auto* gt = w.gpu.GetTransform(kv.first);
            if (!rt || !gt) continue;
            Put(&rt->scale, s3); rt->MarkDirty();
            Put(&gt->scale, s3); gt->MarkDirty();
            ++edited;
        }
        CHECK(edited > 50, "only %d scale edits", edited);
        orc::RefTransformSystemUpdate(w.ref);
        bge::GpuTransformSystem<bge::Scene>::Update(w.gpu);
        w.CompareAll("scale edited between the two systems");
        w.Tick();
    }

    // Bullet's sub-step clock (stepSimulation(dt, 4, fixedStep), PhysicsSystem.cpp:855-863) through the reference-shaped
    // Update(Scene&, const Camera&, const InputSystem&, double): dt = 0.5x (every other call steps), 1x, 2.5x (2 or 3
    // sub-steps), 6x (6 due, 4 simulated, the rest dropped), a hot-reloaded fixedStep, and teleports / body re-creation /
    // trigger events in calls that simulate nothing
    {
        struct Camera {} camera;
        struct InputSystem {} input;
        w.refPhysics.accumulate = true;
        w.refPhysics.localTime = 0.0f;
        w.refPhysics.fixedStep = std::max(w.gpuPhysics.GetConfig().fixedStep, 1.0f / 240.0f); // PhysicsSystem.cpp:855
        w.useReferenceShape = true;
        const double fixed = w.gpuPhysics.GetFixedStep();
        CHECK(fixed == w.dt, "fixedStep of the loaded config");
        const double factors[] = {0.5, 0.5, 0.5, 1.0, 2.5, 2.5, 6.0, 0.25, 0.25, 0.25, 0.25, 1.0, 0.3, 3.7, 0.1};
        int k = 0, zero_calls = 0, clamped_calls = 0;
        for (double fct : factors) {
            w.dt_override = fct * fixed;
            if (k == 1 || k == 8) { // a call that simulates nothing still teleports and re-creates
                float p[3] = {1.0f + k, 20.0f, -3.0f};
                w.Move(ids[2100 + k], p, true);
                for (auto& kv : w.ref.GetRigidBodies()) { if (kv.first % 11 == 0) kv.second.dirty = true; }
                for (auto& kv : w.gpu.GetRigidBodies()) { if (kv.first % 11 == 0) kv.second.dirty = true; }
            }
            w.Tick();
            CHECK(w.gpuPhysics.LastSubSteps() == w.refPhysics.lastSubSteps, "call %d (dt = %.2f x fixedStep): %d sub-steps vs %d in the oracle", k, fct,
                  w.gpuPhysics.LastSubSteps(), w.refPhysics.lastSubSteps);
            zero_calls += w.refPhysics.lastSubSteps == 0;
            clamped_calls += w.refPhysics.lastSubSteps > 4;
            ++k;
        }
        CHECK(zero_calls >= 5 && clamped_calls >= 1, "the dt script must cover 0 sub-steps (%d calls) and > 4 (%d calls)", zero_calls, clamped_calls);
        (void)camera; (void)input;
        w.dt_override = 0.0;
        w.refPhysics.accumulate = false;
        w.useReferenceShape = false;
    }

    // Trigger volumes whose entity loses its Transform: the ghost stays in the world at its last pose and keeps reporting overlaps
    // (EnsureTrigger returns early, nothing removes it); when the Transform returns it is posed from it again.
    {
        std::vector<uint32_t> with_trigger;
        for (auto& kv : w.ref.GetTriggerVolumes()) {
            if (with_trigger.size() < 8 && w.ref.GetTransform(kv.first)) with_trigger.push_back(kv.first);
        }
        CHECK(with_trigger.size() >= 4, "only %zu trigger volumes left for the lost-Transform phase", with_trigger.size());
        for (uint32_t id : with_trigger) { w.ref.RemoveTransform(id); w.gpu.RemoveTransform(id); }
        const size_t before = w.events_seen;
        for (int k = 0; k < 4; ++k) w.Tick();
        CHECK(w.events_seen > before, "no trigger events while the ghosts had no Transform");
        for (size_t k = 0; k < with_trigger.size(); ++k) {
            float p[3], e[3], s[3];
            orc::synth::trs(0xD00D, static_cast<uint32_t>(k), 0, p, e, s);
            orc::RefTransform* rt = w.ref.AddTransform(with_trigger[k]);
            bge::Transform* gt = w.gpu.AddTransform(with_trigger[k]);
            Put(&rt->position, p); Put(&rt->rotationEuler, e); Put(&rt->scale, s);
            Put(&gt->position, p); Put(&gt->rotationEuler, e); Put(&gt->scale, s);
        }
        for (int k = 0; k < 3; ++k) w.Tick();
    }

    // Random edits, one per tick, through every path of the adapter at once: index reuse, topology refreshes, sparse uploads,
    // body / collider / trigger components coming and going, Transforms removed and given back, in Bullet's orientation scheme
    // with the ground plane on.  (Parents are never chosen among a node's descendants: Scene::SetParent does not check.)
    {
        std::mt19937 r2(77);
        // (Transform fields edited WITHOUT MarkDirty — the "stale edits" above — are flushed first and not made again here: the
        //  reference's physics reads such fields when it (re)creates a body or poses a ghost, the adapter uploads a Transform
        //  only when it is dirty — DESIGN.md 3, a documented difference)
        for (auto& kv : w.ref.GetTransforms()) kv.second.MarkDirty();
        for (auto& kv : w.gpu.GetTransforms()) kv.second.MarkDirty();
        w.Tick();
        auto pick_alive = [&]() -> uint32_t {
            for (int tries = 0; tries < 64; ++tries) {
                const uint32_t id = 1 + r2() % static_cast<uint32_t>(n + 40);
                if (w.ref.IsAlive(id)) return id;
            }
            return 0;
        };
        int done[10] = {0};
        std::vector<std::array<uint32_t, 3>> history;
        for (int round = 0; round < 160; ++round) {
            const int op = static_cast<int>(r2() % 10);
            const uint32_t id = pick_alive();
            if (!id) continue;
            if (op == 0) {
                uint32_t p = r2() % 5 == 0 ? 0 : pick_alive();
                for (uint32_t a = p; a != 0; a = w.ref.GetParent(a)) {
                    if (a == id) { p = 0; break; }
                }
                w.SetParent(id, p);
            } else if (op == 1) {
                w.Destroy(id);
            } else if (op == 2) {
                float p[3], e[3], s[3];
                orc::synth::trs(0xC0DE, static_cast<uint32_t>(round), 0, p, e, s);
                p[1] = 1.0f + static_cast<float>(round % 7);
                const uint32_t fresh = w.Create(p, e, s);
                if (round % 2) {
                    float size[3] = {0.3f + 0.1f * (round % 5), 0.6f, 0.4f};
                    w.Body(fresh, static_cast<int>(r2() % 3), 0.5f + static_cast<float>(r2() % 4), size, static_cast<int>(r2() % 2), 1u << (r2() % 3), 0xffffffffu);
                }
                if (round % 3 == 0) w.SetParent(fresh, id);
            } else if (op == 3) {
                float p[3] = {static_cast<float>(round) - 80.0f, 2.0f + static_cast<float>(round % 9), static_cast<float>(r2() % 100) - 50.0f};
                w.Move(id, p, true);
            } else if (op == 4) {
                if (auto* b = w.ref.GetRigidBody(id)) b->dirty = true;
                if (auto* b = w.gpu.GetRigidBody(id)) b->dirty = true;
            } else if (op == 5) {
                if (r2() % 2) { w.ref.RemoveRigidBody(id); w.gpu.RemoveRigidBody(id); }
                else { w.ref.RemoveCollider(id); w.gpu.RemoveCollider(id); }
            } else if (op == 6) {
                // (Transforms taken from entities with bodies and trigger volumes too: body and ghost live on without one)
                if (w.ref.GetTransform(id)) {
                    w.ref.RemoveTransform(id); w.gpu.RemoveTransform(id);
                } else {
                    float p[3], e[3], s[3];
                    orc::synth::trs(0xBEEF, static_cast<uint32_t>(round), 0, p, e, s);
                    p[1] = 3.0f;
                    orc::RefTransform* rt = w.ref.AddTransform(id);
                    bge::Transform* gt = w.gpu.AddTransform(id);
                    Put(&rt->position, p); Put(&rt->rotationEuler, e); Put(&rt->scale, s);
                    Put(&gt->position, p); Put(&gt->rotationEuler, e); Put(&gt->scale, s);
                }
            } else if (op == 7) {
                const float s3[3] = {0.5f + 0.25f * static_cast<float>(r2() % 6), 1.0f, 0.75f + 0.25f * static_cast<float>(r2() % 4)};
                auto* rt = w.ref.GetTransform(id);
                auto* gt = w.gpu.GetTransform(id);
                if (!rt || !gt) continue;
                Put(&rt->scale, s3); rt->MarkDirty();
                Put(&gt->scale, s3); gt->MarkDirty();
            } else if (op == 8) {
                if (!w.ref.GetTransform(id)) continue;
                float size[3] = {0.2f + 0.1f * static_cast<float>(r2() % 8), 0.2f + 0.1f * static_cast<float>(r2() % 8), 0.2f + 0.1f * static_cast<float>(r2() % 8)};
                w.Body(id, static_cast<int>(r2() % 3), 0.25f * static_cast<float>(1 + r2() % 12), size, static_cast<int>(r2() % 2), 1u << (r2() % 3),
                       r2() % 4 ? 0xffffffffu : 0xfffffffdu);
                w.ref.GetRigidBody(id)->dirty = true; w.gpu.GetRigidBody(id)->dirty = true;
                w.ref.GetCollider(id)->dirty = true; w.gpu.GetCollider(id)->dirty = true;
            } else {
                if (w.ref.GetTriggerVolume(id)) {
                    w.ref.RemoveTriggerVolume(id); w.gpu.RemoveTriggerVolume(id);
                } else if (w.ref.GetTransform(id)) {
                    float size[3] = {4.0f + static_cast<float>(r2() % 9), 8.0f, 4.0f + static_cast<float>(r2() % 7)};
                    w.Trigger(id, static_cast<int>(r2() % 2), size, r2() % 3 ? 4u : 0u, 0xffffffffu, r2() % 4 == 0);
                }
            }
            ++done[op];
            const int failures_before = g_failures;
            w.Tick();
            history.push_back({static_cast<uint32_t>(round), static_cast<uint32_t>(op), id});
            if (g_failures != failures_before) {
                std::printf("random edits: first failure in round %d; the operations so far (round op entity):", round);
                for (const auto& h : history) std::printf(" %u:%u:%u", h[0], h[1], h[2]);
                std::printf("\n");
                break;
            }
        }
        for (int op = 0; op < 10 && history.size() >= 150; ++op) CHECK(done[op] > 3, "random edits: operation %d ran only %d times", op, done[op]);
    }

    if (g_failures == 0) std::printf("host adapter: all checks passed (%zu transforms, %d ticks, %zu trigger events)\n", w.gpu.GetTransformCount(), w.ticks, w.events_seen);
    return g_failures ? 1 : 0;
}
