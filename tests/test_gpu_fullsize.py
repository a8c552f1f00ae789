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

from helpers import DT, assert_bits_equal, run_world

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
