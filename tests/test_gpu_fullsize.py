"""Parity at BASELINE.json's FULL sizes.  The hash-map C++ oracle needs minutes there, so these tests use the vectorised
numpy-float32 restatement (oracle/np_oracle.py, pinned bit for bit against the C++ oracle in tests/test_np_oracle.py):
positions and velocities after K ticks and EVERY world matrix must match bit for bit; the euler triples the physics
write-back rewrites once (checked bit for bit against the C++ oracle at small sizes) must stay within the round-trip
bound of SURVEY §8 a-11.  K = 120 ticks for configs[1]/[2]."""
import numpy as np
import pytest

import banggameengine_amd as B
from banggameengine_amd import synth
from oracle import np_oracle as npo
from oracle import pyoracle as po

from helpers import DT, assert_bits_equal, build_oracle, run_oracle, run_world

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,n,ticks", [
    ("flat1m", 1_000_000, 120),        # configs[1]
    ("chains4", 1_000_000, 120),       # configs[2]
    ("subtree64", 2_000_000, 30),      # configs[4] 5b, one GPU's shard
    ("chains4_shard", 2_000_000, 30),  # configs[4] 5a, one GPU's shard
    ("cube4m", 4_000_000, 20),         # configs[3]'s bodies (tick without the pair search)
    ("flat1m", 16_000_000, 3),         # configs[4]'s total entity count on one GPU
])
def test_full_size_bitwise(name, n, ticks):
    wl = synth.config(name, n=n)
    dyn = wl.body_type == 1
    with B.World() as w:
        run_world(w.load(wl), wl, ticks)
        pos, euler = w.download_pose()
        vel = w.download_bodies()["linvel"]
        world = w.download_world()
        assert w.dirty_count() == 0
    # integration: tick 0 creates the bodies at rest, the synthetic velocities are seeded after it
    want_pos, want_vel = npo.integrate(wl.pos, np.zeros_like(wl.vel), dyn, 1, DT)
    want_pos, want_vel = npo.integrate(want_pos, wl.vel, dyn, ticks - 1, DT)
    assert_bits_equal(pos, want_pos, "position")
    assert_bits_equal(vel[dyn], want_vel[dyn], "velocity")
    # euler: untouched where there is no body, within the ZYX round-trip bound where the write-back rewrote it
    assert_bits_equal(euler[~dyn], wl.euler[~dyn], "rotationEuler of plain transforms")
    assert np.abs(euler[dyn] - wl.euler[dyn]).max() < 3e-5
    # transform hierarchy: every world matrix, bit for bit, from the downloaded TRS
    want_world = npo.resolve_world(wl.parent, pos, euler, wl.scale)
    assert_bits_equal(world, want_world, "world")
    del world, want_world


def test_bullet_basis_scheme_full_size_against_the_cpp_oracle():
    """BGE_TICK_BULLET_BASIS at configs[1]'s size: 1 M flat Dynamic bodies, 6 ticks, against the hash-map C++ oracle in its
    kOrientBasis mode (a few seconds of CPU): quaternion, position, rotationEuler and every world matrix bit for bit."""
    wl = synth.config("flat1m")
    ticks = 6
    ref = run_oracle(build_oracle(wl, orient_mode=po.ORIENT_BASIS), wl, ticks)
    with B.World() as w:
        run_world(w.load(wl), wl, ticks, flags=B.TICK_ALL | B.TICK_BULLET_BASIS)
        pos, euler = w.download_pose()
        quat = w.download_bodies()["quat"]
        world = w.download_world()
    assert_bits_equal(quat, ref.bulk_bodies()["quat"], "quaternion")
    assert_bits_equal(pos, ref.bulk_pose()[0], "position")
    assert_bits_equal(euler, ref.bulk_pose()[1], "rotationEuler")
    assert_bits_equal(world, ref.bulk_world()[0], "world")


def test_slab_broadphase_full_size_equals_single_world():
    """configs[3]'s 4 M bodies split over 4 worlds (entity i -> shard i % 4: interleaved in space): the union of the slabs'
    pair lists must be exactly the single world's pair list — 12.6 M pairs, no duplicates."""
    from banggameengine_amd import sharding
    n, nshards = 4_000_000, 4
    wl = synth.config("cube4m", n=n)
    with B.World() as w:
        run_world(w.load(wl), wl, 3, flags=B.TICK_ALL | B.TICK_BROADPHASE)
        total = w.pair_count()
        want = w.pairs(cap=total)
    worlds = []
    try:
        for r in range(nshards):
            ids = np.arange(r, n, nshards, dtype=np.uint32)
            s = B.World(pair_capacity=16 * len(ids))
            s.set_topology(np.full(len(ids), 0xFFFFFFFF, np.uint32))
            s.upload_trs(wl.pos[ids], wl.euler[ids], wl.scale[ids])
            s.upload_bodies(wl.body_type[ids])
            s.set_global_ids(ids)
            for k in range(3):
                s.tick(dt=DT, flags=B.TICK_ALL | B.TICK_AABBS)
                if k == 0:
                    s.set_velocities(wl.vel[ids])
            worlds.append(s)
        counts, cuts = sharding.slab_broadphase_local(worlds, axis=2)
        got = np.concatenate([s.pairs(cap=max(s.pair_count(), 1)) for s in worlds])
    finally:
        for s in worlds:
            s.close()
    assert len(got) == total and total > 10_000_000
    key = got[:, 0].astype(np.uint64) << np.uint64(32) | got[:, 1]
    key.sort()
    assert (np.diff(key) != 0).all()
    assert np.array_equal(key, want[:, 0].astype(np.uint64) << np.uint64(32) | want[:, 1])
    assert n <= counts.sum() < 1.05 * n
