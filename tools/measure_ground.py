#!/usr/bin/env python3
"""Cost of a tick with the reference's ground plane (PhysicsSystem.cpp:149-166) at scale — the path every scene of the
reference takes: k_ground (collide with y = 0, cached manifold, solver) in front of the tick kernel.

Run on the GPU box:  python tools/measure_ground.py [n_bodies]
n flat Dynamic boxes of mixed size and orientation (default 1 M), three phases, each timed over a batch of ticks with the
wall clock around a synchronised batch and, beside it, the same ticks WITHOUT the plane on an identical world:
  falling   every body far above the plane (k_ground rejects it after the cheap test)
  resting   every body in contact, awake: collide + 10 solver iterations per body and tick
  asleep    after 2 s at rest Bullet's deactivation has put them to sleep: neither collided nor solved
Prints one line per phase: ms per tick with / without the plane, contact counts, sleepers.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import banggameengine_amd as B  # noqa: E402
from banggameengine_amd import synth  # noqa: E402
from banggameengine_amd.world import FIXED_DT  # noqa: E402


def build(wl, size, ground, stream):
    w = B.World(stream=stream.cuda_stream)
    w.set_topology(wl.parent)
    w.upload_trs(wl.pos, wl.euler, wl.scale)
    w.upload_bodies(wl.body_type, size=size, shape=np.zeros(wl.n, np.uint8))
    w.set_ground_plane(ground)
    return w


def timed(w, ticks, flags):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w.tick(dt=FIXED_DT, flags=flags, ticks=ticks)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / ticks * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    flags = B.TICK_ALL | (B.TICK_BULLET_BASIS if os.environ.get("BGE_GROUND_BASIS") else 0)
    rng = np.random.default_rng(5)
    wl = synth.config("flat1m", n=n)
    wl.pos[:, 0] = rng.uniform(-3000, 3000, n).astype(np.float32)
    wl.pos[:, 2] = rng.uniform(-3000, 3000, n).astype(np.float32)
    wl.body_type[:] = 1
    size = rng.uniform(0.2, 0.9, (n, 3)).astype(np.float32)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    out = []
    only = os.environ.get("BGE_GROUND_PHASE")   # one phase, with the plane only (for a profiler run)
    grounds = (True,) if only else (True, False)
    # falling: 500 m up, 100 ticks (they fall 3.5 m)
    wl.pos[:, 1] = rng.uniform(500, 600, n).astype(np.float32)
    for ground in (grounds if only in (None, "falling") else ()):
        with build(wl, size, ground, stream) as w:
            w.tick(dt=FIXED_DT, flags=flags, ticks=5)
            out.append(("falling", ground, timed(w, 100, flags), None))
    # resting: dropped from just above the plane; 150 ticks to land and settle, then timed while still awake (< 2 s)
    wl.pos[:, 1] = (np.abs(size).sum(axis=1) * 0.6 + rng.uniform(0.0, 0.3, n)).astype(np.float32)
    for ground in (grounds if only in (None, "resting", "asleep") else ()):
        with build(wl, size, ground, stream) as w:
            w.tick(dt=FIXED_DT, flags=flags, ticks=150)
            ms = timed(w, 60, flags)
            info = None
            if ground:
                cn, _ = w.download_contacts()
                st, _ = w.download_activation()
                info = f"contacts per body: {np.bincount(cn, minlength=5).tolist()}, asleep {int((st == 2).sum())}"
            out.append(("resting", ground, ms, info))
            if ground:
                w.tick(dt=FIXED_DT, flags=flags, ticks=400)   # > 2 s at rest
                ms = timed(w, 100, flags)
                st, _ = w.download_activation()
                out.append(("asleep", True, ms, f"asleep {int((st == 2).sum())} of {n}"))
    base = {ph: ms for ph, g, ms, _ in out if not g}
    for ph, g, ms, info in out:
        if g:
            ref = base.get(ph, base.get("resting", float("nan")))
            print(f"{n} bodies, {ph:8s}: {ms:.4f} ms per tick with the plane, {ref:.4f} without (+{ms - ref:.4f}); {info or ''}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
