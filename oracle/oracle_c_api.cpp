// oracle/oracle_c_api.cpp — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Flat C entry points over the CPU restatement so that tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg can drive it through ctypes.  Nothing in the product
// (banggameengine_amd/, include/) may link or load this library.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <vector>

#if defined(_OPENMP)
#include <omp.h>
#endif

#include "broadphase_ref.h"
#include "bullet_math.h"
#include "bx_math.h"
#include "ecs_ref.h"
#include "physics_ref.h"
#include "soa_ref.h"
#include "synth.h"

using namespace orc;

namespace {
struct Session {
    RefScene scene;
    RefPhysicsSystem physics;
};
inline Session* S(void* h) { return static_cast<Session*>(h); }
inline Float3 F3(const float* p) { return Float3{p[0], p[1], p[2]}; }
} // namespace

extern "C" {

// ---------------------------------------------------------------- math known-answer hooks
void orc_mtx_srt(float* out16, const float* s, const float* r, const float* t)
{
    bxm::mtxSRT(out16, s[0], s[1], s[2], r[0], r[1], r[2], t[0], t[1], t[2]);
}
void orc_mtx_mul(float* out16, const float* a, const float* b) { bxm::mtxMul(out16, a, b); }
void orc_bx_eval(int fn, const float* in, float* out, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) {
        out[i] = fn == 0 ? bxm::cos_(in[i]) : fn == 1 ? bxm::sin_(in[i]) : bxm::floor_(in[i]);
    }
}
// normal matrices of n world matrices (src/render/Renderer.cpp:633-636)
void orc_normal_matrices(const float* world16, float* out16, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) bxm::normalMatrix(out16 + 16 * i, world16 + 16 * i);
}
void orc_set_libm(int which) { bt::g_libm = which; }
// fn: 0 sin, 1 cos, 2 asin (clamped), 3 atan2(a,b)
void orc_libm_eval(int fn, const float* a, const float* b, float* out, uint64_t n)
{
    for (uint64_t i = 0; i < n; ++i) {
        switch (fn) {
        case 0: out[i] = bt::Sin(a[i]); break;
        case 1: out[i] = bt::Cos(a[i]); break;
        case 2: out[i] = bt::Asin(a[i]); break;
        default: out[i] = bt::Atan2(a[i], b[i]); break;
        }
    }
}
void orc_quat_from_transform_euler(const float* e, float* q)
{
    const bt::Quat r = bt::QuatFromTransformEuler(e[0], e[1], e[2]);
    q[0] = r.x; q[1] = r.y; q[2] = r.z; q[3] = r.w;
}
void orc_transform_euler_from_quat(const float* q, float* e)
{
    const bt::Vec3 r = bt::TransformEulerFromMat(bt::MatFromQuat(bt::Quat{q[0], q[1], q[2], q[3]}));
    e[0] = r.x; e[1] = r.y; e[2] = r.z;
}
void orc_box_aabb_half_extents(const float* size, float* out)
{
    const bt::Vec3 r = bt::BoxAabbHalfExtents(size[0], size[1], size[2]);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

// ---------------------------------------------------------------- scene session
void* orc_scene_new() { return new Session(); }
void orc_scene_free(void* h) { delete S(h); }

uint32_t orc_create_entity(void* h) { return S(h)->scene.CreateEntity(); }
void orc_destroy_entity(void* h, uint32_t id) { S(h)->scene.DestroyEntity(id); }
int orc_is_alive(void* h, uint32_t id) { return S(h)->scene.IsAlive(id) ? 1 : 0; }
int orc_add_transform(void* h, uint32_t id) { return S(h)->scene.AddTransform(id) ? 1 : 0; }
void orc_remove_transform(void* h, uint32_t id) { S(h)->scene.RemoveTransform(id); }
int orc_set_trs(void* h, uint32_t id, const float* pos, const float* euler, const float* scale, int markDirty)
{
    RefTransform* t = S(h)->scene.GetTransform(id);
    if (!t) return 0;
    if (pos) t->position = F3(pos);
    if (euler) t->rotationEuler = F3(euler);
    if (scale) t->scale = F3(scale);
    if (markDirty) t->MarkDirty();
    return 1;
}
void orc_mark_transform_dirty(void* h, uint32_t id)
{
    if (RefTransform* t = S(h)->scene.GetTransform(id)) t->MarkDirty();
}
void orc_set_parent(void* h, uint32_t child, uint32_t parent) { S(h)->scene.SetParent(child, parent); }
uint32_t orc_get_parent(void* h, uint32_t child) { return S(h)->scene.GetParent(child); }
int orc_add_collider(void* h, uint32_t id, int shape, const float* size)
{
    RefCollider* c = S(h)->scene.AddCollider(id);
    if (!c) return 0;
    c->shape = static_cast<RefShape>(shape);
    if (size) c->size = F3(size);
    c->dirty = true;
    return 1;
}
void orc_remove_collider(void* h, uint32_t id) { S(h)->scene.RemoveCollider(id); }
int orc_add_rigidbody(void* h, uint32_t id, int type, float mass, uint32_t layer, uint32_t mask)
{
    RefRigidBody* b = S(h)->scene.AddRigidBody(id);
    if (!b) return 0;
    b->type = static_cast<RefBodyType>(type);
    b->mass = mass;
    b->layer = layer;
    b->mask = mask;
    b->dirty = true;
    return 1;
}
void orc_remove_rigidbody(void* h, uint32_t id) { S(h)->scene.RemoveRigidBody(id); }
void orc_mark_body_dirty(void* h, uint32_t id)
{
    if (RefRigidBody* b = S(h)->scene.GetRigidBody(id)) b->dirty = true;
}

int orc_add_trigger(void* h, uint32_t id, int shape, const float* size, uint32_t layer, uint32_t mask, int oneShot, int active)
{
    RefTriggerVolume* t = S(h)->scene.AddTriggerVolume(id);
    if (!t) return 0;
    t->shape = static_cast<RefShape>(shape);
    if (size) t->size = F3(size);
    t->layer = layer;
    t->mask = mask;
    t->oneShot = oneShot != 0;
    t->active = active != 0;
    t->dirty = true;
    return 1;
}
void orc_remove_trigger(void* h, uint32_t id) { S(h)->scene.RemoveTriggerVolume(id); }
int orc_trigger_is_active(void* h, uint32_t id)
{
    const RefTriggerVolume* t = S(h)->scene.GetTriggerVolume(id);
    return t && t->active ? 1 : 0;
}
// events of the last physics update as (type, trigger id, other id) triples; returns how many there are
uint64_t orc_trigger_events(void* h, uint32_t* out3, uint64_t cap)
{
    const auto& ev = S(h)->physics.LastTriggerEvents();
    for (uint64_t k = 0; k < ev.size() && k < cap; ++k) {
        out3[3 * k] = static_cast<uint32_t>(ev[k].type);
        out3[3 * k + 1] = ev[k].trigger;
        out3[3 * k + 2] = ev[k].other;
    }
    return ev.size();
}

void orc_set_physics_options(void* h, float gravityY, int orientMode, int computeAabbs)
{
    S(h)->physics.gravityY = gravityY;
    S(h)->physics.orientMode = orientMode;
    S(h)->physics.computeAabbs = computeAabbs != 0;
}
void orc_transform_update(void* h) { RefTransformSystemUpdate(S(h)->scene); }
// per-index activation state (0 = no body) and deactivation time of the bodies
void orc_bulk_get_activation(void* h, uint64_t n, int32_t* state, float* time)
{
    auto& rts = S(h)->physics.Runtimes();
    for (uint64_t i = 0; i < n; ++i) {
        auto it = rts.find(static_cast<EntityId>(i + 1));
        const bool ok = it != rts.end() && it->second.hasBody;
        if (state) state[i] = ok ? it->second.activation : 0;
        if (time) time[i] = ok ? it->second.deactivationTime : 0.0f;
    }
}
void orc_set_deactivation(void* h, int enabled) { S(h)->physics.deactivation = enabled != 0; }

void orc_physics_update(void* h, double dt) { S(h)->physics.Update(S(h)->scene, dt); }
// Bullet's stepSimulation clock around the sub-steps (physics_ref.h `accumulate`); returns nothing, see orc_last_substeps
void orc_set_accumulator(void* h, int enabled, float fixedStep, int maxSubSteps)
{
    S(h)->physics.accumulate = enabled != 0;
    S(h)->physics.fixedStep = fixedStep;
    S(h)->physics.maxSubSteps = maxSubSteps;
    S(h)->physics.localTime = 0.0f;
}
int orc_last_substeps(void* h) { return S(h)->physics.lastSubSteps; }
// the reference's static ground plane y = 0 with Bullet's contact handling for it (contact_ref.h)
void orc_set_ground_plane(void* h, int enabled) { S(h)->physics.groundPlane = enabled != 0; }
void orc_set_friction(void* h, uint32_t id, float friction)
{
    if (RefRigidBody* b = S(h)->scene.GetRigidBody(id)) b->friction = friction;
}
// contacts of a body with the ground: count, and per point (localA.xyz, appliedImpulse, localB.x, distance, localB.z, appliedImpulseLateral1)
int orc_get_ground_contacts(void* h, uint32_t id, float* out32)
{
    auto& rts = S(h)->physics.Runtimes();
    auto it = rts.find(id);
    if (it == rts.end()) return 0;
    const auto& m = it->second.ground;
    for (int k = 0; k < m.n && out32; ++k) {
        float* o = out32 + 8 * k;
        o[0] = m.p[k].localA.x; o[1] = m.p[k].localA.y; o[2] = m.p[k].localA.z; o[3] = m.p[k].appliedImpulse;
        o[4] = m.p[k].localB.x; o[5] = m.p[k].distance; o[6] = m.p[k].localB.z; o[7] = m.p[k].appliedImpulseLateral1; // (localB.y is exactly 0)
    }
    return m.n;
}
// Dynamic boxes against the scene's Static / Kinematic box colliders (boxbox_ref.h)
void orc_set_static_contacts(void* h, int enabled) { S(h)->physics.staticContacts = enabled != 0; }
void orc_set_restitution(void* h, uint32_t id, float restitution)
{
    if (RefRigidBody* b = S(h)->scene.GetRigidBody(id)) b->restitution = restitution;
}
// a body's manifolds with boxes: returns how many; hdr[2 k] = the other entity, hdr[2 k + 1] = points; per point 12 floats:
// localA.xyz, localB.xyz, normalWorldOnB.xyz, distance, appliedImpulse, appliedImpulseLateral1
int orc_get_box_contacts(void* h, uint32_t id, uint32_t* hdr8, float* out192)
{
    auto& rts = S(h)->physics.Runtimes();
    auto it = rts.find(id);
    if (it == rts.end()) return 0;
    const auto& boxes = it->second.boxes;
    for (size_t k = 0; k < boxes.size() && k < 4; ++k) {
        const auto& m = boxes[k];
        if (hdr8) {
            hdr8[2 * k] = m.other;
            hdr8[2 * k + 1] = static_cast<uint32_t>(m.n);
        }
        for (int j = 0; j < m.n && out192; ++j) {
            float* o = out192 + 48 * k + 12 * j;
            const auto& c = m.p[j];
            o[0] = c.localA.x; o[1] = c.localA.y; o[2] = c.localA.z;
            o[3] = c.localB.x; o[4] = c.localB.y; o[5] = c.localB.z;
            o[6] = c.normalB.x; o[7] = c.normalB.y; o[8] = c.normalB.z;
            o[9] = c.distance; o[10] = c.appliedImpulse; o[11] = c.appliedImpulseLateral1;
        }
    }
    return static_cast<int>(boxes.size());
}
// Dynamic boxes against each other (island_ref.h, physics_ref.h CollideDynamicPairs / StepIsland)
void orc_set_dynamic_contacts(void* h, int enabled) { S(h)->physics.dynamicContacts = enabled != 0; }
// the pair cache of Dynamic boxes in ascending (lower entity, higher entity): returns how many pairs there are; for the first `cap`
// of them hdr[3 k ..] = lower entity, higher entity, points and 4 x 12 floats as orc_get_box_contacts lays a point out
int orc_get_dynamic_pairs(void* h, int cap, uint32_t* hdr, float* out)
{
    const auto& pairs = S(h)->physics.DynamicPairs();
    int k = 0;
    for (const auto& kv : pairs) {
        if (k >= cap) break;
        const auto& m = kv.second.m;
        hdr[3 * k] = kv.first.first;
        hdr[3 * k + 1] = kv.first.second;
        hdr[3 * k + 2] = static_cast<uint32_t>(m.n);
        for (int j = 0; j < 4; ++j) {
            float* o = out + 48 * k + 12 * j;
            for (int q = 0; q < 12; ++q) o[q] = 0.0f;
            if (j >= m.n) continue;
            const auto& c = m.p[j];
            o[0] = c.localA.x; o[1] = c.localA.y; o[2] = c.localA.z;
            o[3] = c.localB.x; o[4] = c.localB.y; o[5] = c.localB.z;
            o[6] = c.normalB.x; o[7] = c.normalB.y; o[8] = c.normalB.z;
            o[9] = c.distance; o[10] = c.appliedImpulse; o[11] = c.appliedImpulseLateral1;
        }
        ++k;
    }
    return static_cast<int>(pairs.size());
}
uint64_t orc_count_dirty(void* h) { return S(h)->scene.CountDirtyTransforms(); }
uint64_t orc_transform_count(void* h) { return S(h)->scene.GetTransformCount(); }

int orc_get_transform(void* h, uint32_t id, float* pos, float* euler, float* scale, float* local16, float* world16,
                      uint8_t* dirty)
{
    const RefTransform* t = S(h)->scene.GetTransform(id);
    if (!t) return 0;
    if (pos) { pos[0] = t->position.x; pos[1] = t->position.y; pos[2] = t->position.z; }
    if (euler) { euler[0] = t->rotationEuler.x; euler[1] = t->rotationEuler.y; euler[2] = t->rotationEuler.z; }
    if (scale) { scale[0] = t->scale.x; scale[1] = t->scale.y; scale[2] = t->scale.z; }
    if (local16) std::memcpy(local16, t->local, 64);
    if (world16) std::memcpy(world16, t->world, 64);
    if (dirty) *dirty = t->dirty ? 1 : 0;
    return 1;
}

int orc_get_body(void* h, uint32_t id, float* origin, float* quat, float* linvel, float* angvel, float* aabb6)
{
    auto& rts = S(h)->physics.Runtimes();
    auto it = rts.find(id);
    if (it == rts.end() || !it->second.hasBody) return 0;
    const RefBodyRuntime& rt = it->second;
    if (origin) { origin[0] = rt.origin.x; origin[1] = rt.origin.y; origin[2] = rt.origin.z; }
    if (quat) { quat[0] = rt.orn.x; quat[1] = rt.orn.y; quat[2] = rt.orn.z; quat[3] = rt.orn.w; }
    if (linvel) { linvel[0] = rt.linvel.x; linvel[1] = rt.linvel.y; linvel[2] = rt.linvel.z; }
    if (angvel) { angvel[0] = rt.angvel.x; angvel[1] = rt.angvel.y; angvel[2] = rt.angvel.z; }
    if (aabb6) {
        for (int a = 0; a < 3; ++a) { aabb6[a] = rt.aabbMin[a]; aabb6[3 + a] = rt.aabbMax[a]; }
    }
    return 1;
}
int orc_set_velocity(void* h, uint32_t id, const float* lin, const float* ang)
{
    auto& rts = S(h)->physics.Runtimes();
    if (rts.find(id) == rts.end()) return 0;
    S(h)->physics.SetVelocity(id, bt::Vec3{lin[0], lin[1], lin[2]}, bt::Vec3{ang[0], ang[1], ang[2]});
    return 1;
}

// ---------------------------------------------------------------- bulk helpers (entity index i <-> id i+1)
// bodyType: 0 Static, 1 Dynamic, 2 Kinematic, 255 = no body.  Arrays may be null where noted.
int orc_bulk_build(void* h, uint64_t n, const int32_t* parentIdx, const uint8_t* hasTransform /*nullable*/,
                   const float* pos, const float* euler, const float* scale, const uint8_t* bodyType /*nullable*/,
                   const float* mass /*nullable*/, const float* size3 /*nullable*/, const uint8_t* shape /*nullable*/,
                   const uint32_t* layer /*nullable*/, const uint32_t* mask /*nullable*/)
{
    RefScene& sc = S(h)->scene;
    if (sc.GetEntityCount() != 0) return 0;
    for (uint64_t i = 0; i < n; ++i) {
        const EntityId id = sc.CreateEntity();
        if (id != i + 1) return 0;
        if (!hasTransform || hasTransform[i]) {
            RefTransform* t = sc.AddTransform(id);
            t->position = F3(pos + 3 * i);
            t->rotationEuler = F3(euler + 3 * i);
            t->scale = F3(scale + 3 * i);
        }
        if (bodyType && bodyType[i] != 255) {
            RefCollider* c = sc.AddCollider(id);
            c->shape = shape ? static_cast<RefShape>(shape[i]) : RefShape::Box;
            if (size3) c->size = F3(size3 + 3 * i);
            RefRigidBody* b = sc.AddRigidBody(id);
            b->type = static_cast<RefBodyType>(bodyType[i]);
            b->mass = mass ? mass[i] : 1.0f;
            if (layer) b->layer = layer[i];
            if (mask) b->mask = mask[i];
        }
    }
    for (uint64_t i = 0; i < n; ++i) {
        if (parentIdx && parentIdx[i] >= 0) sc.SetParent(static_cast<EntityId>(i + 1), static_cast<EntityId>(parentIdx[i] + 1));
    }
    return 1;
}

void orc_bulk_set_trs(void* h, uint64_t first, uint64_t n, const float* pos, const float* euler, const float* scale,
                      int markDirty)
{
    for (uint64_t i = 0; i < n; ++i) {
        RefTransform* t = S(h)->scene.GetTransform(static_cast<EntityId>(first + i + 1));
        if (!t) continue;
        if (pos) t->position = F3(pos + 3 * i);
        if (euler) t->rotationEuler = F3(euler + 3 * i);
        if (scale) t->scale = F3(scale + 3 * i);
        if (markDirty) t->MarkDirty();
    }
}

void orc_bulk_get_world(void* h, uint64_t n, float* world16 /*nullable*/, float* local16 /*nullable*/,
                        uint8_t* dirty /*nullable*/)
{
    for (uint64_t i = 0; i < n; ++i) {
        const RefTransform* t = S(h)->scene.GetTransform(static_cast<EntityId>(i + 1));
        if (!t) {
            if (world16) std::memset(world16 + 16 * i, 0, 64);
            if (local16) std::memset(local16 + 16 * i, 0, 64);
            if (dirty) dirty[i] = 0;
            continue;
        }
        if (world16) std::memcpy(world16 + 16 * i, t->world, 64);
        if (local16) std::memcpy(local16 + 16 * i, t->local, 64);
        if (dirty) dirty[i] = t->dirty ? 1 : 0;
    }
}

void orc_bulk_get_pose(void* h, uint64_t n, float* pos /*nullable*/, float* euler /*nullable*/)
{
    for (uint64_t i = 0; i < n; ++i) {
        const RefTransform* t = S(h)->scene.GetTransform(static_cast<EntityId>(i + 1));
        for (int a = 0; a < 3; ++a) {
            if (pos) pos[3 * i + a] = t ? (&t->position.x)[a] : 0.0f;
            if (euler) euler[3 * i + a] = t ? (&t->rotationEuler.x)[a] : 0.0f;
        }
    }
}

void orc_bulk_set_velocity(void* h, uint64_t n, const float* lin, const float* ang /*nullable*/)
{
    for (uint64_t i = 0; i < n; ++i) {
        const bt::Vec3 l{lin[3 * i], lin[3 * i + 1], lin[3 * i + 2]};
        const bt::Vec3 a = ang ? bt::Vec3{ang[3 * i], ang[3 * i + 1], ang[3 * i + 2]} : bt::Vec3{0, 0, 0};
        S(h)->physics.SetVelocity(static_cast<EntityId>(i + 1), l, a);
    }
}

// per-index body state; rows of bodies that do not exist are zero-filled, exists[i]=0
void orc_bulk_get_bodies(void* h, uint64_t n, float* origin /*nullable*/, float* quat /*nullable*/,
                         float* linvel /*nullable*/, float* angvel /*nullable*/, float* aabb6 /*nullable*/,
                         uint8_t* exists /*nullable*/)
{
    auto& rts = S(h)->physics.Runtimes();
    for (uint64_t i = 0; i < n; ++i) {
        auto it = rts.find(static_cast<EntityId>(i + 1));
        const bool ok = it != rts.end() && it->second.hasBody;
        if (exists) exists[i] = ok ? 1 : 0;
        const RefBodyRuntime* rt = ok ? &it->second : nullptr;
        if (origin) { origin[3*i] = ok ? rt->origin.x : 0; origin[3*i+1] = ok ? rt->origin.y : 0; origin[3*i+2] = ok ? rt->origin.z : 0; }
        if (quat) { quat[4*i] = ok ? rt->orn.x : 0; quat[4*i+1] = ok ? rt->orn.y : 0; quat[4*i+2] = ok ? rt->orn.z : 0; quat[4*i+3] = ok ? rt->orn.w : 0; }
        if (linvel) { linvel[3*i] = ok ? rt->linvel.x : 0; linvel[3*i+1] = ok ? rt->linvel.y : 0; linvel[3*i+2] = ok ? rt->linvel.z : 0; }
        if (angvel) { angvel[3*i] = ok ? rt->angvel.x : 0; angvel[3*i+1] = ok ? rt->angvel.y : 0; angvel[3*i+2] = ok ? rt->angvel.z : 0; }
        if (aabb6) {
            for (int a = 0; a < 3; ++a) { aabb6[6*i+a] = ok ? rt->aabbMin[a] : 0; aabb6[6*i+3+a] = ok ? rt->aabbMax[a] : 0; }
        }
    }
}

// Pair set of the AABBs computed by the last physics update (computeAabbs must be on).
// Pairs are (index_a, index_b), a<b, sorted; returns the total number found (may exceed cap).
// method: 0 brute force, 1 sweep.
uint64_t orc_pairs(void* h, uint64_t n, int method, uint32_t* outPairs, uint64_t cap)
{
    auto& rts = S(h)->physics.Runtimes();
    std::vector<BroadphaseBody> bodies;
    std::vector<uint32_t> index;
    bodies.reserve(rts.size());
    for (uint64_t i = 0; i < n; ++i) {
        auto it = rts.find(static_cast<EntityId>(i + 1));
        if (it == rts.end() || !it->second.hasBody) continue;
        const RefBodyRuntime& rt = it->second;
        BroadphaseBody b;
        for (int a = 0; a < 3; ++a) { b.mn[a] = rt.aabbMin[a]; b.mx[a] = rt.aabbMax[a]; }
        b.group = rt.layer;
        b.mask = rt.mask;
        b.isStatic = rt.type == RefBodyType::Static ? 1 : 0;
        bodies.push_back(b);
        index.push_back(static_cast<uint32_t>(i));
    }
    const PairList pairs = method == 0 ? PairsBruteForce(bodies) : PairsSweep(bodies);
    uint64_t k = 0;
    for (const auto& p : pairs) {
        if (k < cap) { outPairs[2 * k] = index[p.first]; outPairs[2 * k + 1] = index[p.second]; }
        ++k;
    }
    return k;
}

// ---------------------------------------------------------------- synthetic inputs
void orc_synth_fill(int shape, int posBox, uint64_t seed, uint64_t first, uint64_t n, int32_t* parent /*nullable*/,
                    float* pos, float* euler, float* scale, float* vel /*nullable*/)
{
    for (uint64_t k = 0; k < n; ++k) {
        const int64_t i = static_cast<int64_t>(first + k);
        if (parent) parent[k] = static_cast<int32_t>(synth::parent_of(shape, i));
        synth::trs(seed, i, posBox, pos + 3 * k, euler + 3 * k, scale + 3 * k);
        if (vel) synth::velocity(seed, i, vel + 3 * k);
    }
}

// ---------------------------------------------------------------- CPU baseline ("the reference CPU path", 1 thread)
// Builds the synthetic scene in the hash-map AoS store and times `ticks` iterations of
//   physics.Update(scene, dt); TransformSystem::Update(scene)
// (src/core/Application.cpp:256,284).  bodiesOnRootsOnly: 1 => only parent-less entities carry a
// Dynamic body (configs 3/5), 0 => every entity does (configs 1/2/4).  Returns seconds for the
// timed ticks; *updates = transforms recomputed per tick (== n here: everything is dirty each tick).
double orc_bench_tick(int shape, int posBox, int bodiesOnRootsOnly, int computeAabbs, uint64_t n, uint64_t seed,
                      int warm, int ticks, double dt, uint64_t* updates)
{
    Session s;
    s.physics.computeAabbs = computeAabbs != 0;
    std::vector<float> vel(3 * n);
    for (uint64_t i = 0; i < n; ++i) {
        const EntityId id = s.scene.CreateEntity();
        RefTransform* t = s.scene.AddTransform(id);
        float p[3], e[3], sc[3];
        synth::trs(seed, static_cast<int64_t>(i), posBox, p, e, sc);
        t->position = F3(p); t->rotationEuler = F3(e); t->scale = F3(sc);
        synth::velocity(seed, static_cast<int64_t>(i), &vel[3 * i]);
        const bool isRoot = synth::parent_of(shape, static_cast<int64_t>(i)) < 0;
        if (!bodiesOnRootsOnly || isRoot) {
            s.scene.AddCollider(id);
            RefRigidBody* b = s.scene.AddRigidBody(id);
            b->type = RefBodyType::Dynamic;
            b->mass = 1.0f;
        }
    }
    for (uint64_t i = 0; i < n; ++i) {
        const int64_t p = synth::parent_of(shape, static_cast<int64_t>(i));
        if (p >= 0) s.scene.SetParent(static_cast<EntityId>(i + 1), static_cast<EntityId>(p + 1));
    }
    // first tick creates the bodies (velocity reset), then seed the synthetic velocities
    s.physics.Update(s.scene, dt);
    RefTransformSystemUpdate(s.scene);
    for (uint64_t i = 0; i < n; ++i) {
        s.physics.SetVelocity(static_cast<EntityId>(i + 1), bt::Vec3{vel[3*i], vel[3*i+1], vel[3*i+2]}, bt::Vec3{0, 0, 0});
    }
    for (int k = 0; k < warm; ++k) {
        s.physics.Update(s.scene, dt);
        RefTransformSystemUpdate(s.scene);
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < ticks; ++k) {
        s.physics.Update(s.scene, dt);
        RefTransformSystemUpdate(s.scene);
    }
    const auto t1 = std::chrono::steady_clock::now();
    if (updates) *updates = n;
    return std::chrono::duration<double>(t1 - t0).count();
}

// ---------------------------------------------------------------- CPU-opt baseline (dense SoA, all threads)
// Same synthetic scene and the same arithmetic as orc_bench_tick for non-spinning mass-1 bodies (the euler write-back
// is the identity for them after the first tick, so it is skipped).  world_out (nullable) receives the final world
// matrices so that tests can compare it with the port.  threads <= 0: all hardware threads.
double orc_bench_tick_soa(int shape, int posBox, int bodiesOnRootsOnly, uint64_t n, uint64_t seed, int warm, int ticks,
                          double dt, int threads, int* threads_used, float* world_out)
{
#if defined(_OPENMP)
    if (threads > 0) omp_set_num_threads(threads);
    if (threads_used) *threads_used = omp_get_max_threads();
#else
    if (threads_used) *threads_used = 1;
#endif
    SoaScene s;
    s.n = n;
    s.parent.resize(n);
    s.pos.resize(3 * n); s.euler.resize(3 * n); s.scale.resize(3 * n); s.vel.assign(3 * n, 0.0f);
    s.dynamic.resize(n);
    s.local.resize(16 * n); s.world.resize(16 * n);
    std::vector<float> seed_vel(3 * n);
    for (uint64_t i = 0; i < n; ++i) {
        s.parent[i] = static_cast<int32_t>(synth::parent_of(shape, static_cast<int64_t>(i)));
        synth::trs(seed, static_cast<int64_t>(i), posBox, &s.pos[3 * i], &s.euler[3 * i], &s.scale[3 * i]);
        synth::velocity(seed, static_cast<int64_t>(i), &seed_vel[3 * i]);
        s.dynamic[i] = (!bodiesOnRootsOnly || s.parent[i] < 0) ? 1 : 0;
        if (s.dynamic[i]) {
            // the body-creation tick re-poses the body from its Transform and writes rotationEuler back once
            // (setEulerZYX -> basis -> getEulerZYX, physics_ref.h); afterwards a non-spinning body keeps it
            const bt::Vec3 e = bt::TransformEulerFromMat(bt::MatFromQuat(
                bt::QuatFromTransformEuler(s.euler[3 * i], s.euler[3 * i + 1], s.euler[3 * i + 2])));
            s.euler[3 * i] = e.x; s.euler[3 * i + 1] = e.y; s.euler[3 * i + 2] = e.z;
        }
    }
    s.build_levels();
    const float fdt = static_cast<float>(dt);
    s.tick(fdt, -9.81f); // body creation tick: starts from rest
    s.vel = seed_vel;
    for (int k = 0; k < warm; ++k) s.tick(fdt, -9.81f);
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < ticks; ++k) s.tick(fdt, -9.81f);
    const auto t1 = std::chrono::steady_clock::now();
    if (world_out) std::memcpy(world_out, s.world.data(), 64 * n);
    return std::chrono::duration<double>(t1 - t0).count();
}

} // extern "C"
