"""ctypes binding of libbge_world.so (include/bge_world.h).  Fails loudly when the library is missing."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BGE_WORLD_LIB selects another build of the same ABI (A/B timing experiments); default: the in-tree library
_LIB = os.environ.get("BGE_WORLD_LIB") or os.path.join(_HERE, "libbge_world.so")


class BgeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"bge error {code}: {msg}")
        self.code = code


class WorldDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("stream", C.c_void_p),
                ("pair_capacity", C.c_uint64)]


class WorldInfo(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_entities", "n_transforms", "n_slots", "n_tiles", "n_passes", "n_roots",
                                          "n_limbo", "n_bodies", "max_depth")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


# name -> (restype, argtypes); every symbol include/bge_world.h declares
_vp, _u64, _u32, _f = C.c_void_p, C.c_uint64, C.c_uint32, C.c_float
SYMBOLS = {
    "bge_last_error": (C.c_char_p, []),
    "bge_version": (_u32, []),
    "bge_world_create": (C.c_int, [C.POINTER(WorldDesc), C.POINTER(_vp)]),
    "bge_world_destroy": (None, [_vp]),
    "bge_world_set_topology": (C.c_int, [_vp, _u64, _vp, _vp]),
    "bge_world_upload_trs": (C.c_int, [_vp, _u64, _u64, _vp, _vp, _vp]),
    "bge_world_mark_dirty": (C.c_int, [_vp, _u64, _u64]),
    "bge_world_upload_trs_indexed": (C.c_int, [_vp, _u64, _vp, _vp, _vp, _vp]),
    "bge_world_upload_bodies_indexed": (C.c_int, [_vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bge_world_download_world_indexed": (C.c_int, [_vp, _u64, _vp, _vp]),
    "bge_world_download_pose_indexed": (C.c_int, [_vp, _u64, _vp, _vp, _vp]),
    "bge_world_upload_bodies": (C.c_int, [_vp, _u64, _u64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bge_world_set_velocities": (C.c_int, [_vp, _u64, _u64, _vp, _vp]),
    "bge_world_tick": (C.c_int, [_vp, _f, _vp, _u32]),
    "bge_world_tick_many": (C.c_int, [_vp, _u32, _f, _vp, _u32]),
    "bge_world_step_simulation": (C.c_int, [_vp, C.c_double, C.c_int, _f, _vp, _u32, _vp]),
    "bge_world_reset_clock": (C.c_int, [_vp]),
    "bge_world_set_ground_plane": (C.c_int, [_vp, C.c_int]),
    "bge_world_upload_friction": (C.c_int, [_vp, _u64, _u64, _vp]),
    "bge_world_upload_friction_indexed": (C.c_int, [_vp, _u64, _vp, _vp]),
    "bge_world_download_contacts": (C.c_int, [_vp, _u64, _u64, _vp, _vp]),
    "bge_world_set_static_contacts": (C.c_int, [_vp, C.c_int]),
    "bge_world_set_dynamic_contacts": (C.c_int, [_vp, C.c_int]),
    "bge_world_download_dynamic_pairs": (C.c_int, [_vp, _u64, _vp, _vp, _vp]),
    "bge_world_upload_restitution": (C.c_int, [_vp, _u64, _u64, _vp]),
    "bge_world_upload_restitution_indexed": (C.c_int, [_vp, _u64, _vp, _vp]),
    "bge_world_download_box_contacts": (C.c_int, [_vp, _u64, _u64, _vp, _vp, _vp]),
    "bge_world_sync": (C.c_int, [_vp]),
    "bge_world_profile_enable": (C.c_int, [_vp, C.c_int]),
    "bge_world_profile_read": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(_u64)]),
    "bge_world_download_world": (C.c_int, [_vp, _u64, _u64, _vp]),
    "bge_world_download_pose": (C.c_int, [_vp, _u64, _u64, _vp, _vp]),
    "bge_world_download_normal": (C.c_int, [_vp, _u64, _u64, _vp]),
    "bge_world_download_bodies": (C.c_int, [_vp, _u64, _u64, _vp, _vp, _vp, _vp]),
    "bge_world_download_dirty": (C.c_int, [_vp, _u64, _u64, _vp]),
    "bge_world_download_activation": (C.c_int, [_vp, _u64, _u64, _vp, _vp]),
    "bge_world_set_sleeping": (C.c_int, [_vp, C.c_float, C.c_float, C.c_float]),
    "bge_world_dirty_count": (C.c_int, [_vp, C.POINTER(_u64)]),
    "bge_host_alloc": (C.c_int, [_u64, C.POINTER(_vp)]),
    "bge_host_free": (C.c_int, [_vp]),
    "bge_world_set_global_ids": (C.c_int, [_vp, _u64, _u64, _vp]),
    "bge_world_aabb_bounds": (C.c_int, [_vp, _vp, _vp, C.POINTER(_u64)]),
    "bge_world_axis_histogram": (C.c_int, [_vp, C.c_uint32, C.c_float, C.c_float, C.c_uint32, _vp]),
    "bge_balanced_cuts": (C.c_int, [_vp, C.c_uint32, C.c_float, C.c_float, C.c_uint32, _vp]),
    "bge_world_bp_route": (C.c_int, [_vp, C.c_uint32, C.c_uint32, _vp, _vp]),
    "bge_world_bp_pack": (C.c_int, [_vp, _vp]),
    "bge_world_bp_find": (C.c_int, [_vp, _vp, _u64, C.c_uint32, C.c_float, C.c_float]),
    "bge_world_bp_exchange": (C.c_int, [_vp, C.c_uint32]),
    "bge_world_pairs": (C.c_int, [_vp, _vp, _u64, C.POINTER(_u64)]),
    "bge_world_upload_triggers": (C.c_int, [_vp, _u64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "bge_world_trigger_events": (C.c_int, [_vp, _vp, _u64, C.POINTER(_u64)]),
    "bge_world_trigger_active": (C.c_int, [_vp, _u64, _vp, _vp]),
    "bge_world_trigger_query_stats": (C.c_int, [_vp, C.POINTER(_u32), C.POINTER(_u32)]),
    "bge_world_set_trigger_stay_events": (C.c_int, [_vp, C.c_int]),
    "bge_world_trigger_diff_stats": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "bge_world_pack_roots": (C.c_int, [_vp, _vp]),
    "bge_world_device_array": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_u64)]),
    "bge_world_get_info": (C.c_int, [_vp, C.POINTER(WorldInfo)]),
    "bge_comm_unique_id": (C.c_int, [_vp]),
    "bge_world_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _u64]),
    "bge_world_gather_roots": (C.c_int, [_vp, C.POINTER(_vp)]),
    "bge_world_comm_wait": (C.c_int, [_vp]),
    "bge_world_comm_set_mode": (C.c_int, [_vp, C.c_int]),
    "bge_world_download_gathered": (C.c_int, [_vp, _vp, _u64]),
    "bge_world_comm_destroy": (C.c_int, [_vp]),
    "bge_flatten_topology": (C.c_int, [_u64, _vp, _vp, _vp, _vp, _vp, C.POINTER(WorldInfo)]),
    "bge_partition_subtrees": (C.c_int, [_u64, _vp, _vp, _u32, _vp, _vp]),
}

_lib = None


def lib_path() -> str:
    return _LIB


def lib():
    """Load libbge_world.so.  torch is imported first so that the HIP runtime torch bundles
    (same SONAME libamdhip64.so.7) is the single runtime of the process."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise ImportError(f"{_LIB} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is plumbing, the library also loads against /opt/rocm
            pass
        l = C.CDLL(_LIB)
        for name, (res, args) in SYMBOLS.items():
            try:
                fn = getattr(l, name)  # AttributeError if the library does not export a declared symbol
            except AttributeError:
                if os.environ.get("BGE_WORLD_LIB"):  # A/B experiments against an older build: tolerate, fail on use
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise BgeError(rc, lib().bge_last_error().decode("utf-8", "replace"))
