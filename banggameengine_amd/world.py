"""Thin numpy face of the C ABI (include/bge_world.h).  One method per entry point; no logic of its own."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from ._capi import WorldDesc, WorldInfo, check, lib

NO_PARENT = 0xFFFFFFFF
BODY_STATIC, BODY_DYNAMIC, BODY_KINEMATIC, BODY_NONE = 0, 1, 2, 255
SHAPE_BOX, SHAPE_CAPSULE = 0, 1
TICK_PHYSICS, TICK_TRANSFORMS, TICK_BROADPHASE, TICK_ALL, TICK_GATHER_ROOTS, TICK_NORMAL_MATRICES, TICK_AABBS = 1, 2, 4, 3, 8, 16, 32
TICK_BULLET_BASIS = 64
ARRAY_WORLD, ARRAY_ROOT_WORLDS, ARRAY_SLOT_OF_ENTITY, ARRAY_POSITION, ARRAY_PAIRS = 0, 1, 2, 3, 4

# fixed step and gravity of the reference (assets/config/physics.json:2-3)
FIXED_DT = float(np.float32(0.0083333333))
GRAVITY = (0.0, -9.81, 0.0)


def _arr(a, dtype, shape_tail=None):
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape_tail is not None and a.ndim == 2 and a.shape[1] != shape_tail:
        raise ValueError(f"expected (*, {shape_tail}) array, got {a.shape}")
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class World:
    """A device-resident mirror of one reference ``Scene`` (see include/bge_world.h)."""

    def __init__(self, device: int = -1, stream: int | None = None, pair_capacity: int = 0):
        self._h = C.c_void_p()
        desc = WorldDesc(C.sizeof(WorldDesc), device, C.c_void_p(stream) if stream else None, pair_capacity)
        check(lib().bge_world_create(C.byref(desc), C.byref(self._h)))
        self.n = 0

    def close(self):
        if self._h:
            lib().bge_world_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- topology and components
    def set_topology(self, parent, has_transform=None):
        parent = _arr(parent, np.uint32)
        ht = _arr(has_transform, np.uint8)
        self.n = len(parent)
        check(lib().bge_world_set_topology(self._h, self.n, _p(parent), _p(ht)))
        return self

    def upload_trs(self, pos=None, euler=None, scale=None, first=0):
        pos, euler, scale = (_arr(a, np.float32, 3) for a in (pos, euler, scale))
        count = next(len(a) for a in (pos, euler, scale) if a is not None)
        check(lib().bge_world_upload_trs(self._h, first, count, _p(pos), _p(euler), _p(scale)))

    def upload_trs_indexed(self, entity_index, pos=None, euler=None, scale=None):
        """Sparse form of upload_trs: row i belongs to entity entity_index[i] (bge_world_upload_trs_indexed)."""
        idx = _arr(entity_index, np.uint32)
        pos, euler, scale = (_arr(a, np.float32, 3) for a in (pos, euler, scale))
        check(lib().bge_world_upload_trs_indexed(self._h, len(idx), _p(idx), _p(pos), _p(euler), _p(scale)))

    def mark_dirty(self, first=0, count=None):
        check(lib().bge_world_mark_dirty(self._h, first, self.n - first if count is None else count))

    def upload_bodies(self, body_type, mass=None, shape=None, size=None, layer=None, mask=None, first=0):
        t = _arr(body_type, np.uint8)
        check(lib().bge_world_upload_bodies(self._h, first, len(t), _p(t), _p(_arr(mass, np.float32)),
                                            _p(_arr(shape, np.uint8)), _p(_arr(size, np.float32, 3)),
                                            _p(_arr(layer, np.uint32)), _p(_arr(mask, np.uint32))))

    def set_velocities(self, linvel=None, angvel=None, first=0):
        l, a = _arr(linvel, np.float32, 3), _arr(angvel, np.float32, 3)
        count = len(l) if l is not None else len(a)
        check(lib().bge_world_set_velocities(self._h, first, count, _p(l), _p(a)))

    # -- tick
    def tick(self, dt=FIXED_DT, gravity=GRAVITY, flags=TICK_ALL, ticks=1):
        g = (C.c_float * 3)(*gravity)
        check(lib().bge_world_tick_many(self._h, ticks, dt, g, flags))

    def step_simulation(self, dt, max_sub_steps=4, fixed_step=FIXED_DT, gravity=GRAVITY, flags=TICK_PHYSICS) -> int:
        """Bullet's stepSimulation(dt, max_sub_steps, fixed_step) around the world's ticks; returns the sub-steps due."""
        g = (C.c_float * 3)(*gravity)
        n = C.c_int(0)
        check(lib().bge_world_step_simulation(self._h, float(dt), int(max_sub_steps), float(fixed_step), g, flags, C.byref(n)))
        return int(n.value)

    def set_ground_plane(self, enabled=True):
        """The reference's static plane y = 0 with Bullet's contact handling (include/bge_world.h)."""
        check(lib().bge_world_set_ground_plane(self._h, int(enabled)))

    def upload_friction(self, friction, first=0):
        fr = _arr(friction, np.float32)
        check(lib().bge_world_upload_friction(self._h, first, len(fr), _p(fr)))

    def set_static_contacts(self, enabled=True):
        """Dynamic boxes collide with the Static / Kinematic box colliders of the scene (bge_world.h)."""
        check(lib().bge_world_set_static_contacts(self._h, int(enabled)))

    def upload_restitution(self, restitution, first=0):
        r = _arr(restitution, np.float32)
        check(lib().bge_world_upload_restitution(self._h, first, len(r), _p(r)))

    def set_dynamic_contacts(self, enabled=True):
        """Dynamic boxes collide with each other: a persistent manifold per pair, simulation islands of several bodies
        (bge_world_set_dynamic_contacts)."""
        check(lib().bge_world_set_dynamic_contacts(self._h, int(enabled)))

    def download_dynamic_pairs(self):
        """(hdr, points): hdr[k] = lower entity index, higher entity index, points of the k-th pair of Dynamic boxes in the pair
        cache (ascending); points[k, j] = localA.xyz, localB.xyz, normalWorldOnB.xyz, distance, appliedImpulse, appliedImpulseLateral1."""
        total = C.c_uint64(0)
        check(lib().bge_world_download_dynamic_pairs(self._h, 0, None, None, C.byref(total)))
        n = int(total.value)
        hdr = np.zeros((max(n, 1), 3), np.uint32)
        pts = np.zeros((max(n, 1), 4, 12), np.float32)
        if n:
            check(lib().bge_world_download_dynamic_pairs(self._h, n, _p(hdr), _p(pts), C.byref(total)))
        return hdr[:n], pts[:n]

    def download_box_contacts(self, first=0, count=None):
        """(n_manifolds[count], header[count, 4, 2] = (other entity, points), points[count, 4, 4, 12]) — ascending other entity."""
        count = self.n - first if count is None else count
        n = np.zeros(count, np.uint8)
        hdr = np.zeros((count, 4, 2), np.uint32)
        pts = np.zeros((count, 4, 4, 12), np.float32)
        check(lib().bge_world_download_box_contacts(self._h, first, count, _p(n), _p(hdr), _p(pts)))
        return n, hdr, pts

    def download_contacts(self, first=0, count=None):
        """(n, points): n[i] contact points of body i with the ground; points[i, k] = localA.xyz, appliedImpulse, localB.x, distance, localB.z, lateral."""
        count = self.n - first if count is None else count
        n = np.zeros(count, np.uint8)
        pts = np.zeros((count, 4, 8), np.float32)
        check(lib().bge_world_download_contacts(self._h, first, count, _p(n), _p(pts)))
        return n, pts

    def reset_clock(self):
        check(lib().bge_world_reset_clock(self._h))

    def sync(self):
        check(lib().bge_world_sync(self._h))

    def profile_enable(self, mode=1):
        """0 off, 1 one event pair per tick() call, 2 one pair per tick (see include/bge_world.h)."""
        check(lib().bge_world_profile_enable(self._h, int(mode)))

    def profile_read(self):
        """(summed tick-kernel milliseconds, ticks) since the last read; synchronises the stream."""
        ms, n = C.c_double(0), C.c_uint64(0)
        check(lib().bge_world_profile_read(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    # -- results
    def download_world(self, first=0, count=None, out=None):
        count = self.n - first if count is None else count
        out = np.empty((count, 16), np.float32) if out is None else out
        check(lib().bge_world_download_world(self._h, first, count, _p(out)))
        return out

    def download_normal(self, first=0, count=None):
        count = self.n - first if count is None else count
        out = np.empty((count, 16), np.float32)
        check(lib().bge_world_download_normal(self._h, first, count, _p(out)))
        return out

    def download_pose(self, first=0, count=None):
        count = self.n - first if count is None else count
        pos, euler = np.empty((count, 3), np.float32), np.empty((count, 3), np.float32)
        check(lib().bge_world_download_pose(self._h, first, count, _p(pos), _p(euler)))
        return pos, euler

    def download_bodies(self, first=0, count=None):
        count = self.n - first if count is None else count
        v, w, q, bb = (np.empty((count, k), np.float32) for k in (3, 3, 4, 6))
        check(lib().bge_world_download_bodies(self._h, first, count, _p(v), _p(w), _p(q), _p(bb)))
        return dict(linvel=v, angvel=w, quat=q, aabb=bb)

    def download_dirty(self, first=0, count=None):
        count = self.n - first if count is None else count
        d = np.empty(count, np.uint8)
        check(lib().bge_world_download_dirty(self._h, first, count, _p(d)))
        return d.astype(bool)

    def download_activation(self, first=0, count=None):
        """(state, time): bge_activation per entity (0 none, 1 ACTIVE_TAG, 2 ISLAND_SLEEPING, 3 WANTS_DEACTIVATION,
        4 DISABLE_DEACTIVATION) and Bullet's m_deactivationTime."""
        count = self.n - first if count is None else count
        st, tm = np.empty(count, np.uint8), np.empty(count, np.float32)
        check(lib().bge_world_download_activation(self._h, first, count, _p(st), _p(tm)))
        return st, tm

    def set_sleeping(self, linear=0.8, angular=1.0, seconds=2.0):
        check(lib().bge_world_set_sleeping(self._h, linear, angular, seconds))

    # -- sharded broadphase (global pair set across worlds / ranks)
    def set_global_ids(self, ids, first=0):
        ids = _arr(ids, np.uint32)
        check(lib().bge_world_set_global_ids(self._h, first, len(ids), _p(ids)))

    def aabb_bounds(self):
        mn, mx = np.empty(3, np.float32), np.empty(3, np.float32)
        n = C.c_uint64(0)
        check(lib().bge_world_aabb_bounds(self._h, _p(mn), _p(mx), C.byref(n)))
        return mn, mx, int(n.value)

    def axis_histogram(self, axis, lo, hi, bins=4096):
        hist = np.zeros(bins, np.uint64)
        check(lib().bge_world_axis_histogram(self._h, axis, lo, hi, bins, _p(hist)))
        return hist

    def bp_route(self, axis, cuts):
        """cuts: nranks + 1 slab boundaries (the outer two are ignored).  Returns records per destination slab."""
        cuts = _arr(cuts, np.float32)
        counts = np.zeros(len(cuts) - 1, np.uint64)
        check(lib().bge_world_bp_route(self._h, axis, len(cuts) - 1, _p(cuts), _p(counts)))
        return counts

    def bp_pack(self, send_device_ptr):
        check(lib().bge_world_bp_pack(self._h, C.c_void_p(send_device_ptr)))

    def bp_find(self, records_device_ptr, n_records, axis, window_lo, window_hi):
        check(lib().bge_world_bp_find(self._h, C.c_void_p(records_device_ptr), n_records, axis, window_lo, window_hi))

    def bp_exchange(self, axis=2):
        check(lib().bge_world_bp_exchange(self._h, axis))

    def dirty_count(self) -> int:
        v = C.c_uint64(0)
        check(lib().bge_world_dirty_count(self._h, C.byref(v)))
        return int(v.value)

    def pairs(self, cap=None):
        """Sorted (a, b) entity-index pairs of the last BROADPHASE tick."""
        total = C.c_uint64(0)
        cap = int(cap or max(1024, 8 * self.n))
        buf = np.empty((cap, 2), np.uint32)
        check(lib().bge_world_pairs(self._h, _p(buf), cap, C.byref(total)))
        if total.value > cap:
            raise _capi.BgeError(-1, f"{total.value} pairs found but the host buffer holds {cap}")
        out = buf[: total.value]
        order = np.lexsort((out[:, 1], out[:, 0]))
        return out[order].copy()

    def pair_count(self) -> int:
        total = C.c_uint64(0)
        check(lib().bge_world_pairs(self._h, None, 0, C.byref(total)))
        return int(total.value)

    # -- trigger volumes
    def upload_triggers(self, entity_index, shape=None, size=None, layer=None, mask=None, one_shot=None, active=None, keep_order=False):
        """The world processes its triggers in the order of the uploaded array (bge_world.h); unless keep_order is set the set is
        sent in ascending entity order — the order the oracle's ProcessTriggerEvents walks (oracle/physics_ref.h)."""
        e = _arr(entity_index, np.uint32)
        if not keep_order and len(e) > 1:
            order = np.argsort(e, kind="stable")
            pick = lambda a: None if a is None else np.asarray(a)[order]
            e, shape, size, layer, mask, one_shot, active = e[order], pick(shape), pick(size), pick(layer), pick(mask), pick(one_shot), pick(active)
        check(lib().bge_world_upload_triggers(self._h, len(e), _p(e), _p(_arr(shape, np.uint8)), _p(_arr(size, np.float32, 3)),
                                              _p(_arr(layer, np.uint32)), _p(_arr(mask, np.uint32)),
                                              _p(_arr(one_shot, np.uint8)), _p(_arr(active, np.uint8))))

    def trigger_events(self):
        """(type, trigger entity, other entity) rows since the last call, sorted; type 0 Enter, 1 Stay, 2 Exit."""
        total = C.c_uint64(0)
        check(lib().bge_world_trigger_events(self._h, None, 0, C.byref(total)))
        out = np.empty((int(total.value), 3), np.uint32)
        check(lib().bge_world_trigger_events(self._h, _p(out), len(out), C.byref(total)))
        return out[np.lexsort((out[:, 2], out[:, 1], out[:, 0]))].copy() if len(out) else out

    def trigger_active(self, entity_index):
        e = _arr(entity_index, np.uint32)
        out = np.empty(len(e), np.uint8)
        check(lib().bge_world_trigger_active(self._h, len(e), _p(e), _p(out)))
        return out.astype(bool)

    def set_trigger_stay_events(self, enabled=True):
        """Stay records in trigger_events() (the reference's behaviour, the default) or Enter / Exit only (bge_world.h)."""
        check(lib().bge_world_set_trigger_stay_events(self._h, int(bool(enabled))))

    def trigger_diff_stats(self):
        """(ticks whose Enter / Exit difference was taken on the device, ticks that took it on the host, Stay events left out)."""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(lib().bge_world_trigger_diff_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return int(a.value), int(b.value), int(c.value)

    def trigger_query_stats(self):
        """(ghosts that walked the broadphase grid, ghosts tested against every body) in the last tick."""
        a, b = C.c_uint32(0), C.c_uint32(0)
        check(lib().bge_world_trigger_query_stats(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # -- multi-GPU support / device-resident consumers
    def pack_roots(self, dst_device_ptr: int | None = None):
        check(lib().bge_world_pack_roots(self._h, C.c_void_p(dst_device_ptr) if dst_device_ptr else None))

    # native RCCL collective (see include/bge_world.h)
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(lib().bge_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, nranks: int, rank: int, unique_id: bytes, rows_per_rank: int):
        buf = C.create_string_buffer(unique_id, 128)
        check(lib().bge_world_comm_init(self._h, nranks, rank, buf, rows_per_rank))

    def gather_roots(self) -> int:
        """Pack this rank's roots and enqueue the frame's all-gather; returns the device pointer of the table."""
        ptr = C.c_void_p()
        check(lib().bge_world_gather_roots(self._h, C.byref(ptr)))
        return int(ptr.value or 0)

    def download_gathered(self, nranks: int, rows_per_rank: int):
        out = np.empty((nranks, rows_per_rank, 16), np.float32)
        check(lib().bge_world_download_gathered(self._h, _p(out), out.size))
        return out

    def comm_wait(self):
        check(lib().bge_world_comm_wait(self._h))

    def comm_set_mode(self, mode: int):
        """0 = ncclAllGather, 1 = direct send/recv per peer (include/bge_world.h)."""
        check(lib().bge_world_comm_set_mode(self._h, int(mode)))

    def comm_destroy(self):
        check(lib().bge_world_comm_destroy(self._h))

    def device_array(self, which: int):
        ptr, n = C.c_void_p(), C.c_uint64(0)
        check(lib().bge_world_device_array(self._h, which, C.byref(ptr), C.byref(n)))
        return int(ptr.value or 0), int(n.value)

    def info(self) -> dict:
        inf = WorldInfo()
        check(lib().bge_world_get_info(self._h, C.byref(inf)))
        return inf.as_dict()

    # -- convenience used by tests and the bench: build a whole synthetic workload
    def load(self, wl, with_bodies=True):
        self.set_topology(wl.parent)
        self.upload_trs(wl.pos, wl.euler, wl.scale)
        if with_bodies:
            self.upload_bodies(wl.body_type)
        return self


class PinnedArray:
    """numpy view of page-locked host memory (bge_host_alloc); keep the object alive as long as the array is used."""

    def __init__(self, shape, dtype=np.float32):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._p = C.c_void_p()
        check(lib().bge_host_alloc(n, C.byref(self._p)))
        buf = (C.c_char * max(n, 1)).from_address(self._p.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def __del__(self):
        try:
            if self._p:
                lib().bge_host_free(self._p)
                self._p = C.c_void_p()
        except Exception:
            pass


def balanced_cuts(hist, lo, hi, nranks):
    """Host-only: slab boundaries at the k/nranks quantiles of an (all-reduced) axis histogram."""
    hist = _arr(hist, np.uint64)
    cuts = np.zeros(nranks + 1, np.float32)
    check(lib().bge_balanced_cuts(_p(hist), len(hist), lo, hi, nranks, _p(cuts)))
    return cuts


def flatten_topology(parent, has_transform=None):
    """Host-only: the slot / level / pass the library would give each entity, plus layout info."""
    parent = _arr(parent, np.uint32)
    ht = _arr(has_transform, np.uint8)
    n = len(parent)
    slot = np.empty(n, np.uint32)
    level = np.empty(n, np.uint8)
    pas = np.empty(n, np.uint32)
    inf = WorldInfo()
    check(lib().bge_flatten_topology(n, _p(parent), _p(ht), _p(slot), _p(level), _p(pas), C.byref(inf)))
    return slot, level, pas, inf.as_dict()


def partition_subtrees(parent, nranks, has_transform=None):
    """Host-only: rank of every entity (whole subtrees per rank) and the node count per rank."""
    parent = _arr(parent, np.uint32)
    ht = _arr(has_transform, np.uint8)
    n = len(parent)
    rank = np.empty(n, np.uint32)
    load = np.empty(nranks, np.uint64)
    check(lib().bge_partition_subtrees(n, _p(parent), _p(ht), nranks, _p(rank), _p(load)))
    return rank, load
