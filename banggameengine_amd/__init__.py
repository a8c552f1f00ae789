"""banggameengine_amd — MI355X-native ECS world tick for BangGameEngine's SandboxCity.

The product is the C-ABI shared library ``libbge_world.so`` (include/bge_world.h, HIP kernels under
``csrc/``).  This Python package is plumbing around it for tests and ``bench.py``: a ctypes binding
(:mod:`._capi`), a thin :class:`World` wrapper over numpy arrays, the synthetic-input generator of
the benchmark configurations and the multi-GPU sharding helpers.  It never falls back to a CPU
implementation: importing works without a GPU, creating a :class:`World` without one raises.
"""
from ._capi import BgeError, lib, lib_path  # noqa: F401
from .world import (BODY_DYNAMIC, BODY_KINEMATIC, BODY_NONE, BODY_STATIC, NO_PARENT, SHAPE_BOX, SHAPE_CAPSULE,  # noqa: F401
                    TICK_AABBS, TICK_ALL, TICK_BULLET_BASIS, TICK_BROADPHASE, TICK_GATHER_ROOTS, TICK_NORMAL_MATRICES, TICK_PHYSICS, TICK_TRANSFORMS, World, flatten_topology,
                    partition_subtrees)

__all__ = ["World", "BgeError", "lib", "lib_path", "flatten_topology", "partition_subtrees"]
