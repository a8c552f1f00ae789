"""Shared test plumbing: run the same scene through the oracle (CPU) and the C-ABI world (GPU)."""
from __future__ import annotations

import numpy as np

from oracle import pyoracle as po

NO_PARENT = 0xFFFFFFFF
DT = float(np.float32(0.0083333333))  # assets/config/physics.json:3 as binary32


def parent_i32(parent_u32):
    p = np.asarray(parent_u32)
    return np.where(p == NO_PARENT, -1, p.astype(np.int64)).astype(np.int32)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bits_equal(got, want, what):
    g, w = bits(got), bits(want)
    if np.array_equal(g, w):
        return
    # NaNs may differ in payload/sign; everything else must match bit for bit
    gn, wn = np.isnan(got), np.isnan(want)
    bad = (g != w) & ~(gn & wn)
    if bad.any():
        idx = np.argwhere(bad)[:5]
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.size} words differ, first at {idx.tolist()}: "
                             f"got {got[tuple(idx[0])]!r} want {want[tuple(idx[0])]!r}")


def matrix_rel_err(got, want):
    """Norm-relative error per 4x4 matrix: max |got-want| / max(|want|)."""
    got = np.asarray(got, np.float64).reshape(-1, 16)
    want = np.asarray(want, np.float64).reshape(-1, 16)
    scale = np.maximum(np.abs(want).max(axis=1), 1e-30)
    return (np.abs(got - want).max(axis=1) / scale).max() if len(got) else 0.0


def build_oracle(wl, orient_mode=po.ORIENT_IDEAL, aabbs=False, has_transform=None, **body_kw):
    ref = po.RefScene()
    ref.SetPhysicsOptions(-9.81, orient_mode, aabbs)
    ref.bulk_build(parent_i32(wl.parent), wl.pos, wl.euler, wl.scale, has_transform=has_transform,
                   body_type=wl.body_type, **body_kw)
    return ref


def run_oracle(ref, wl, ticks, seed_velocity=True, physics=True, angvel=None):
    for k in range(ticks):
        if physics:
            ref.PhysicsSystemUpdate(DT)
        ref.TransformSystemUpdate()
        if k == 0 and seed_velocity and physics:
            ref.bulk_set_velocity(wl.vel, angvel)
    return ref


def run_world(world, wl, ticks, seed_velocity=True, flags=3, angvel=None):
    for k in range(ticks):
        world.tick(dt=DT, flags=flags)
        if k == 0 and seed_velocity and (flags & 1):
            world.set_velocities(wl.vel, angvel)
    return world
