"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the
product package (banggameengine_amd) must never import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

ORIENT_IDEAL, ORIENT_QUAT, ORIENT_BASIS = 0, 1, 2
LIBM_DET, LIBM_PLATFORM = 0, 1
SHAPE_FLAT, SHAPE_CHAINS4, SHAPE_SUBTREE64 = 0, 1, 2
POS_SLAB, POS_CUBE = 0, 1
BODY_STATIC, BODY_DYNAMIC, BODY_KINEMATIC, BODY_NONE = 0, 1, 2, 255

_f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile (g++, no reference sources involved)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"])
    return _LIB_PATH


def _ptr(a, dtype):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    return a.ctypes.data_as(C.c_void_p), a


class _Lib:
    def __init__(self):
        build()
        self.l = C.CDLL(_LIB_PATH)
        l = self.l
        l.orc_scene_new.restype = C.c_void_p
        l.orc_scene_free.argtypes = [C.c_void_p]
        l.orc_create_entity.restype = C.c_uint32
        l.orc_create_entity.argtypes = [C.c_void_p]
        l.orc_get_parent.restype = C.c_uint32
        l.orc_count_dirty.restype = C.c_uint64
        l.orc_transform_count.restype = C.c_uint64
        l.orc_pairs.restype = C.c_uint64
        l.orc_bench_tick.restype = C.c_double
        l.orc_bench_tick.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int,
                                     C.c_double, C.POINTER(C.c_uint64)]
        l.orc_bench_tick_soa.restype = C.c_double
        l.orc_bench_tick_soa.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_double,
                                         C.c_int, C.POINTER(C.c_int), C.c_void_p]
        l.orc_trigger_events.restype = C.c_uint64
        l.orc_add_trigger.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_int]
        l.orc_physics_update.argtypes = [C.c_void_p, C.c_double]
        l.orc_set_accumulator.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int]
        l.orc_last_substeps.argtypes = [C.c_void_p]
        l.orc_last_substeps.restype = C.c_int
        l.orc_set_ground_plane.argtypes = [C.c_void_p, C.c_int]
        l.orc_set_friction.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
        l.orc_set_restitution.argtypes = [C.c_void_p, C.c_uint32, C.c_float]
        l.orc_set_static_contacts.argtypes = [C.c_void_p, C.c_int]
        l.orc_get_box_contacts.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        l.orc_get_box_contacts.restype = C.c_int
        l.orc_set_dynamic_contacts.argtypes = [C.c_void_p, C.c_int]
        l.orc_get_dynamic_pairs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        l.orc_get_dynamic_pairs.restype = C.c_int
        l.orc_get_ground_contacts.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
        l.orc_get_ground_contacts.restype = C.c_int
        l.orc_set_physics_options.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_int]
        l.orc_add_rigidbody.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_uint32, C.c_uint32]


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _Lib()
    return _lib.l


def _vp(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


# ----------------------------------------------------------------------------- math hooks
def mtx_srt(scale, euler, pos):
    out = np.empty(16, np.float32)
    s, r, t = (_c(x, np.float32) for x in (scale, euler, pos))
    lib().orc_mtx_srt(_vp(out), _vp(s), _vp(r), _vp(t))
    return out


def mtx_mul(a, b):
    out = np.empty(16, np.float32)
    a, b = _c(a, np.float32), _c(b, np.float32)
    lib().orc_mtx_mul(_vp(out), _vp(a), _vp(b))
    return out


def bx_eval(fn, x):
    """fn: 'cos' | 'sin' | 'floor' evaluated with the bx restatement."""
    x = _c(x, np.float32)
    out = np.empty_like(x)
    lib().orc_bx_eval({"cos": 0, "sin": 1, "floor": 2}[fn], _vp(x), _vp(out), C.c_uint64(x.size))
    return out


def normal_matrices(world):
    """transpose(inverse(world)) per matrix, bx::mtxInverse + bx::mtxTranspose restated."""
    w = _c(world, np.float32).reshape(-1, 16)
    out = np.empty_like(w)
    lib().orc_normal_matrices(_vp(w), _vp(out), C.c_uint64(len(w)))
    return out


def set_libm(which: int):
    lib().orc_set_libm(C.c_int(which))


def libm_eval(fn, a, b=None):
    a = _c(a, np.float32)
    b = _c(b if b is not None else np.zeros_like(a), np.float32)
    out = np.empty_like(a)
    lib().orc_libm_eval({"sin": 0, "cos": 1, "asin": 2, "atan2": 3}[fn], _vp(a), _vp(b), _vp(out), C.c_uint64(a.size))
    return out


def quat_from_transform_euler(e):
    e = _c(e, np.float32)
    q = np.empty(4, np.float32)
    lib().orc_quat_from_transform_euler(_vp(e), _vp(q))
    return q


def transform_euler_from_quat(q):
    q = _c(q, np.float32)
    e = np.empty(3, np.float32)
    lib().orc_transform_euler_from_quat(_vp(q), _vp(e))
    return e


def box_aabb_half_extents(size):
    s = _c(size, np.float32)
    o = np.empty(3, np.float32)
    lib().orc_box_aabb_half_extents(_vp(s), _vp(o))
    return o


def synth_fill(shape, pos_box, seed, first, n, with_vel=True):
    parent = np.empty(n, np.int32)
    pos = np.empty((n, 3), np.float32)
    euler = np.empty((n, 3), np.float32)
    scale = np.empty((n, 3), np.float32)
    vel = np.empty((n, 3), np.float32) if with_vel else None
    lib().orc_synth_fill(C.c_int(shape), C.c_int(pos_box), C.c_uint64(seed), C.c_uint64(first), C.c_uint64(n),
                         _vp(parent), _vp(pos), _vp(euler), _vp(scale), _vp(vel))
    return parent, pos, euler, scale, vel


def bench_tick(shape, pos_box, bodies_on_roots_only, compute_aabbs, n, seed, warm, ticks, dt=1.0 / 120.0):
    """Time the reference-faithful CPU path (1 thread).  Returns (seconds, updates_per_tick)."""
    upd = C.c_uint64(0)
    sec = lib().orc_bench_tick(shape, pos_box, int(bodies_on_roots_only), int(compute_aabbs), n, seed, warm, ticks,
                               float(np.float32(dt)), C.byref(upd))
    return sec, upd.value


def bench_tick_soa(shape, pos_box, bodies_on_roots_only, n, seed, warm, ticks, dt=1.0 / 120.0, threads=0, want_world=False):
    """Time the dense-SoA all-thread CPU path ("CPU-opt").  Returns (seconds, threads_used[, world])."""
    used = C.c_int(0)
    world = np.empty((n, 16), np.float32) if want_world else None
    sec = lib().orc_bench_tick_soa(shape, pos_box, int(bodies_on_roots_only), n, seed, warm, ticks, float(np.float32(dt)),
                                   threads, C.byref(used), _vp(world))
    return (sec, used.value, world) if want_world else (sec, used.value)


# ----------------------------------------------------------------------------- scene session
class RefScene:
    """Thin Python face of the reference-faithful store; method names follow the reference's Scene."""

    def __init__(self):
        self.h = C.c_void_p(lib().orc_scene_new())
        self.n = 0

    def close(self):
        if self.h:
            lib().orc_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- single-entity API (ids as in the reference: 1-based, 0 invalid)
    def CreateEntity(self):
        return lib().orc_create_entity(self.h)

    def DestroyEntity(self, eid):
        lib().orc_destroy_entity(self.h, C.c_uint32(eid))

    def IsAlive(self, eid):
        return bool(lib().orc_is_alive(self.h, C.c_uint32(eid)))

    def AddTransform(self, eid, pos=None, euler=None, scale=None):
        ok = lib().orc_add_transform(self.h, C.c_uint32(eid))
        if ok and (pos is not None or euler is not None or scale is not None):
            self.SetTRS(eid, pos, euler, scale, mark_dirty=True)
        return bool(ok)

    def RemoveTransform(self, eid):
        lib().orc_remove_transform(self.h, C.c_uint32(eid))

    def SetTRS(self, eid, pos=None, euler=None, scale=None, mark_dirty=True):
        p, e, s = _c(pos, np.float32), _c(euler, np.float32), _c(scale, np.float32)
        return bool(lib().orc_set_trs(self.h, C.c_uint32(eid), _vp(p), _vp(e), _vp(s), C.c_int(int(mark_dirty))))

    def MarkDirty(self, eid):
        lib().orc_mark_transform_dirty(self.h, C.c_uint32(eid))

    def SetParent(self, child, parent):
        lib().orc_set_parent(self.h, C.c_uint32(child), C.c_uint32(parent))

    def GetParent(self, child):
        return lib().orc_get_parent(self.h, C.c_uint32(child))

    def AddCollider(self, eid, shape=0, size=(0.5, 0.5, 0.5)):
        s = _c(size, np.float32)
        return bool(lib().orc_add_collider(self.h, C.c_uint32(eid), C.c_int(shape), _vp(s)))

    def RemoveCollider(self, eid):
        lib().orc_remove_collider(self.h, C.c_uint32(eid))

    def AddRigidBody(self, eid, body_type=BODY_DYNAMIC, mass=1.0, layer=1, mask=0xFFFFFFFF):
        return bool(lib().orc_add_rigidbody(self.h, eid, body_type, mass, layer, mask))

    def RemoveRigidBody(self, eid):
        lib().orc_remove_rigidbody(self.h, C.c_uint32(eid))

    def AddTriggerVolume(self, eid, shape=0, size=(0.5, 0.5, 0.5), layer=0, mask=0xFFFFFFFF, one_shot=False, active=True):
        s = _c(size, np.float32)
        return bool(lib().orc_add_trigger(self.h, eid, shape, _vp(s), layer, mask, int(one_shot), int(active)))

    def RemoveTriggerVolume(self, eid):
        lib().orc_remove_trigger(self.h, C.c_uint32(eid))

    def TriggerIsActive(self, eid):
        return bool(lib().orc_trigger_is_active(self.h, C.c_uint32(eid)))

    def TriggerEvents(self):
        """(type, trigger id, other id) rows of the last PhysicsSystemUpdate, sorted; type 0 Enter, 1 Stay, 2 Exit."""
        cap = 1 << 16
        while True:
            out = np.empty((cap, 3), np.uint32)
            k = lib().orc_trigger_events(self.h, _vp(out), C.c_uint64(cap))
            if k <= cap:
                ev = out[:k]
                return ev[np.lexsort((ev[:, 2], ev[:, 1], ev[:, 0]))].copy()
            cap = int(k)

    def MarkBodyDirty(self, eid):
        lib().orc_mark_body_dirty(self.h, C.c_uint32(eid))

    def GetTransform(self, eid):
        pos, euler, scale = (np.empty(3, np.float32) for _ in range(3))
        local, world = np.empty(16, np.float32), np.empty(16, np.float32)
        dirty = C.c_uint8(0)
        ok = lib().orc_get_transform(self.h, C.c_uint32(eid), _vp(pos), _vp(euler), _vp(scale), _vp(local), _vp(world),
                                     C.byref(dirty))
        if not ok:
            return None
        return dict(position=pos, rotationEuler=euler, scale=scale, local=local, world=world, dirty=bool(dirty.value))

    def GetBody(self, eid):
        o, q, v, w, bb = (np.empty(k, np.float32) for k in (3, 4, 3, 3, 6))
        ok = lib().orc_get_body(self.h, C.c_uint32(eid), _vp(o), _vp(q), _vp(v), _vp(w), _vp(bb))
        if not ok:
            return None
        return dict(origin=o, quat=q, linvel=v, angvel=w, aabb=bb)

    def SetVelocity(self, eid, lin, ang=(0, 0, 0)):
        l, a = _c(lin, np.float32), _c(ang, np.float32)
        return bool(lib().orc_set_velocity(self.h, C.c_uint32(eid), _vp(l), _vp(a)))

    # --- systems
    def SetPhysicsOptions(self, gravity_y=-9.81, orient_mode=ORIENT_IDEAL, compute_aabbs=False):
        lib().orc_set_physics_options(self.h, gravity_y, orient_mode, int(compute_aabbs))

    def TransformSystemUpdate(self):
        lib().orc_transform_update(self.h)

    def PhysicsSystemUpdate(self, dt):
        lib().orc_physics_update(self.h, float(dt))

    def SetAccumulator(self, enabled=True, fixed_step=1.0 / 120.0, max_sub_steps=4):
        """Bullet's stepSimulation(dt, max_sub_steps, fixed_step) clock around the sub-steps (resets m_localTime)."""
        lib().orc_set_accumulator(self.h, int(enabled), float(np.float32(fixed_step)), int(max_sub_steps))

    def SetGroundPlane(self, enabled=True):
        """The static plane y = 0 of every reference world (PhysicsSystem.cpp:149-166) with Bullet's contact handling."""
        lib().orc_set_ground_plane(self.h, int(enabled))

    def SetFriction(self, eid, friction):
        lib().orc_set_friction(self.h, eid, float(friction))

    def SetRestitution(self, eid, restitution):
        lib().orc_set_restitution(self.h, eid, float(restitution))

    def SetStaticContacts(self, enabled=True):
        """Dynamic boxes collide with the Static / Kinematic box colliders of the scene (Bullet's btBoxBoxCollisionAlgorithm)."""
        lib().orc_set_static_contacts(self.h, int(enabled))

    def BoxContacts(self, eid):
        """[(other entity id, rows)]: rows[j] = localA.xyz, localB.xyz, normalWorldOnB.xyz, distance, appliedImpulse,
        appliedImpulseLateral1 of the j-th point of the body's manifold with that box; ascending entity id."""
        hdr = np.zeros((4, 2), np.uint32)
        out = np.zeros((4, 4, 12), np.float32)
        n = lib().orc_get_box_contacts(self.h, eid, _vp(hdr), _vp(out))
        return [(int(hdr[k, 0]), out[k, :hdr[k, 1]].copy()) for k in range(min(n, 4))]

    def SetDynamicContacts(self, enabled=True):
        """Dynamic boxes collide with each other; bodies whose fed AABBs overlap form one simulation island (island_ref.h)."""
        lib().orc_set_dynamic_contacts(self.h, int(enabled))

    def DynamicPairs(self):
        """(hdr, points): hdr[k] = lower entity id, higher entity id, number of points of the k-th pair of Dynamic boxes in the
        pair cache (ascending); points[k, j] = the 12 floats BoxContacts documents."""
        n = lib().orc_get_dynamic_pairs(self.h, 0, None, None)
        hdr = np.zeros((max(n, 1), 3), np.uint32)
        out = np.zeros((max(n, 1), 4, 12), np.float32)
        n = lib().orc_get_dynamic_pairs(self.h, n, _vp(hdr), _vp(out))
        return hdr[:n], out[:n]

    def GroundContacts(self, eid):
        """(n, rows): rows[k] = localA.xyz, appliedImpulse, localB.x, distance, localB.z, appliedImpulseLateral1 of the body's k-th contact."""
        out = np.zeros((4, 8), np.float32)
        n = lib().orc_get_ground_contacts(self.h, eid, _vp(out))
        return n, out[:n].copy()

    def LastSubSteps(self):
        return int(lib().orc_last_substeps(self.h))

    def CountDirtyTransforms(self):
        return lib().orc_count_dirty(self.h)

    def GetTransformCount(self):
        return lib().orc_transform_count(self.h)

    # --- bulk (entity index i <-> id i+1)
    def bulk_build(self, parent, pos, euler, scale, has_transform=None, body_type=None, mass=None, size=None,
                   shape=None, layer=None, mask=None):
        n = len(parent)
        self.n = n
        self._keep = [
            _c(parent, np.int32), _c(has_transform, np.uint8), _c(pos, np.float32), _c(euler, np.float32),
            _c(scale, np.float32), _c(body_type, np.uint8), _c(mass, np.float32), _c(size, np.float32),
            _c(shape, np.uint8), _c(layer, np.uint32), _c(mask, np.uint32)]
        ok = lib().orc_bulk_build(self.h, C.c_uint64(n), *[_vp(a) for a in self._keep])
        if not ok:
            raise RuntimeError("orc_bulk_build failed (scene not empty?)")
        return self

    def bulk_set_trs(self, first, pos=None, euler=None, scale=None, mark_dirty=True):
        arrs = [_c(a, np.float32) for a in (pos, euler, scale)]
        n = next(len(a) for a in arrs if a is not None)
        lib().orc_bulk_set_trs(self.h, C.c_uint64(first), C.c_uint64(n), *[_vp(a) for a in arrs], C.c_int(int(mark_dirty)))

    def bulk_world(self, with_local=False):
        n = self.n
        world = np.empty((n, 16), np.float32)
        local = np.empty((n, 16), np.float32) if with_local else None
        dirty = np.empty(n, np.uint8)
        lib().orc_bulk_get_world(self.h, C.c_uint64(n), _vp(world), _vp(local), _vp(dirty))
        return (world, local, dirty) if with_local else (world, dirty)

    def bulk_pose(self):
        n = self.n
        pos, euler = np.empty((n, 3), np.float32), np.empty((n, 3), np.float32)
        lib().orc_bulk_get_pose(self.h, C.c_uint64(n), _vp(pos), _vp(euler))
        return pos, euler

    def bulk_set_velocity(self, lin, ang=None):
        l, a = _c(lin, np.float32), _c(ang, np.float32)
        lib().orc_bulk_set_velocity(self.h, C.c_uint64(len(l)), _vp(l), _vp(a))

    def bulk_bodies(self):
        n = self.n
        o, q, v, w, bb = (np.empty((n, k), np.float32) for k in (3, 4, 3, 3, 6))
        ex = np.empty(n, np.uint8)
        lib().orc_bulk_get_bodies(self.h, C.c_uint64(n), _vp(o), _vp(q), _vp(v), _vp(w), _vp(bb), _vp(ex))
        return dict(origin=o, quat=q, linvel=v, angvel=w, aabb=bb, exists=ex.astype(bool))

    def bulk_activation(self):
        """(state, time): Bullet activation state per entity index (0 = no body, 1 ACTIVE_TAG, 2 ISLAND_SLEEPING,
        3 WANTS_DEACTIVATION, 4 DISABLE_DEACTIVATION) and m_deactivationTime."""
        n = self.n
        st, tm = np.empty(n, np.int32), np.empty(n, np.float32)
        lib().orc_bulk_get_activation(self.h, C.c_uint64(n), _vp(st), _vp(tm))
        return st, tm

    def set_deactivation(self, enabled):
        lib().orc_set_deactivation(self.h, C.c_int(1 if enabled else 0))

    def pairs(self, method="sweep", cap=None):
        cap = cap or max(1024, 64 * self.n)
        out = np.empty((cap, 2), np.uint32)
        k = lib().orc_pairs(self.h, C.c_uint64(self.n), C.c_int(0 if method == "brute" else 1), _vp(out), C.c_uint64(cap))
        if k > cap:
            return self.pairs(method, cap=int(k))
        return out[:k].copy()
