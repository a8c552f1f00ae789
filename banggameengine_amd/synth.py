"""Synthetic inputs of the benchmark configurations, numpy edition.

Counter-based splitmix64: one 24-bit uniform per (seed, entity, field); bit-identical to
oracle/synth.h (tests/test_synth.py).  The reference ships only a 3-entity scene
(assets/scenes/demo.json), so every larger workload is synthetic.
"""
from __future__ import annotations

import numpy as np

FLAT, CHAINS4, SUBTREE64 = 0, 1, 2
SLAB, CUBE = 0, 1
NO_PARENT = np.uint32(0xFFFFFFFF)

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix64(z):
    z = z ^ (z >> np.uint64(30))
    z = z * _M1
    z = z ^ (z >> np.uint64(27))
    z = z * _M2
    z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed: int, entity, field: int):
    """u in [0,1) as float32 for an array of entity indices."""
    with np.errstate(over="ignore"):
        e = np.asarray(entity, dtype=np.uint64)
        h = _mix64(np.uint64(seed) + _GOLD * (np.uint64(16) * e + np.uint64(field + 1)))
    return (h >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)


def _range(seed, e, field, lo, hi):
    lo, hi = np.float32(lo), np.float32(hi)
    return lo + (hi - lo) * uniform(seed, e, field)


def parents(shape: int, first: int, n: int):
    """uint32 parent entity index (global numbering), NO_PARENT for roots."""
    i = np.arange(first, first + n, dtype=np.int64)
    p = np.full(n, -1, np.int64)
    if shape == CHAINS4:
        m = (i % 4) != 0
        p[m] = i[m] - 1
    elif shape == SUBTREE64:
        base = i - (i % 64)
        k = i % 64
        m1 = (k >= 1) & (k < 4)
        m2 = (k >= 4) & (k < 16)
        m3 = k >= 16
        p[m1] = base[m1]
        p[m2] = base[m2] + 1 + (k[m2] - 4) // 4
        p[m3] = base[m3] + 4 + (k[m3] - 16) // 4
    out = p.astype(np.uint32)
    out[p < 0] = NO_PARENT
    return out


def trs(seed: int, first: int, n: int, pos_box: int = SLAB):
    e = np.arange(first, first + n, dtype=np.uint64)
    pos = np.empty((n, 3), np.float32)
    if pos_box == CUBE:
        for a in range(3):
            pos[:, a] = _range(seed, e, a, 0.0, 262.0)
    else:
        pos[:, 0] = _range(seed, e, 0, -250.0, 250.0)
        pos[:, 1] = _range(seed, e, 1, 1.0, 50.0)
        pos[:, 2] = _range(seed, e, 2, -250.0, 250.0)
    euler = np.empty((n, 3), np.float32)
    euler[:, 0] = _range(seed, e, 3, -1.5, 1.5)
    euler[:, 1] = _range(seed, e, 4, -3.1, 3.1)
    euler[:, 2] = _range(seed, e, 5, -3.1, 3.1)
    scale = np.empty((n, 3), np.float32)
    for a in range(3):
        scale[:, a] = _range(seed, e, 6 + a, 0.5, 2.0)
    return pos, euler, scale


def velocity(seed: int, first: int, n: int):
    e = np.arange(first, first + n, dtype=np.uint64)
    v = np.empty((n, 3), np.float32)
    for a in range(3):
        v[:, a] = _range(seed, e, 9 + a, -1.0, 1.0)
    return v


class Workload:
    """One benchmark configuration: topology + TRS + which entities carry a Dynamic body."""

    def __init__(self, name, shape, n, seed, pos_box=SLAB, bodies_on_roots_only=False, first=0):
        self.name, self.shape, self.n, self.seed, self.pos_box = name, shape, int(n), int(seed), pos_box
        self.bodies_on_roots_only = bodies_on_roots_only
        self.first = int(first)
        gp = parents(shape, self.first, self.n)
        # local numbering [0, n): shards start on subtree boundaries, so parents stay inside the shard
        self.parent = np.where(gp == NO_PARENT, NO_PARENT, (gp.astype(np.int64) - self.first).astype(np.uint32))
        self.pos, self.euler, self.scale = trs(seed, self.first, self.n, pos_box)
        self.vel = velocity(seed, self.first, self.n)
        is_root = self.parent == NO_PARENT
        self.body_type = np.where(is_root | (not bodies_on_roots_only), 1, 255).astype(np.uint8)

    @property
    def bytes_per_update(self) -> float:
        """Algorithmic HBM bytes per entity-update."""
        return {FLAT: 140.0, CHAINS4: 113.0, SUBTREE64: (140.0 + 63 * 104.0) / 64.0}[self.shape]


DEFAULT_N = {"flat10k": 10_000, "flat1m": 1_000_000, "chains4": 1_000_000, "cube4m": 4_000_000,
             "chains4_shard": 2_000_000, "subtree64": 2_000_000}


def config(name: str, n: int | None = None, first: int = 0) -> Workload:
    """The BASELINE.json configurations by name."""
    if n is None and name in DEFAULT_N:
        n = DEFAULT_N[name]
    if name == "flat10k":      # configs[0]
        return Workload(name, FLAT, n or 10_000, 0xBA5E0001, first=first)
    if name == "flat1m":       # configs[1]
        return Workload(name, FLAT, n or 1_000_000, 0xBA5E0002, first=first)
    if name == "chains4":      # configs[2]
        return Workload(name, CHAINS4, n or 1_000_000, 0xBA5E0003, bodies_on_roots_only=True, first=first)
    if name == "cube4m":       # configs[3]
        return Workload(name, FLAT, n or 4_000_000, 0xBA5E0004, pos_box=CUBE, first=first)
    if name == "chains4_shard":  # configs[4], variant 5a
        return Workload(name, CHAINS4, n or 2_000_000, 0xBA5E0005, bodies_on_roots_only=True, first=first)
    if name == "subtree64":    # configs[4], variant 5b
        return Workload(name, SUBTREE64, n or 2_000_000, 0xBA5E0005, bodies_on_roots_only=True, first=first)
    raise ValueError(name)
