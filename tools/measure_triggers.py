#!/usr/bin/env python3
"""Cost of the trigger pass of a broadphase tick with MANY trigger volumes, with and without the grid look-up.

Run on the GPU box:  python tools/measure_triggers.py [n_triggers] [n_bodies]
Builds BASELINE's cube scene (4 M bodies by default) with n_triggers ghosts 1..6 units wide riding on random bodies, and
times ticks (wall clock around a synchronised batch; a tick with triggers synchronises anyway, its events are for the
host) three ways: no triggers, all ghosts against all bodies (BGE_TRIGGER_GRID_MIN above the count), small ghosts through
the broadphase grid (the default above 64 ghosts).  The two trigger runs must report the same events.
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


def run(wl, trig, grid_min, ticks=8):
    os.environ["BGE_TRIGGER_GRID_MIN"] = str(grid_min)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    flags = B.TICK_ALL | B.TICK_BROADPHASE
    with B.World(stream=stream.cuda_stream) as w:
        w.load(wl)
        if trig is not None:
            w.upload_triggers(*trig)
        w.tick(dt=FIXED_DT)
        w.set_velocities(wl.vel)
        w.tick(dt=FIXED_DT, flags=flags, ticks=2)
        events = [w.trigger_events()] if trig is not None else []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(ticks):
            w.tick(dt=FIXED_DT, flags=flags)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / ticks * 1e3
        if trig is not None:
            events.append(w.trigger_events())   # everything the timed ticks reported (fetched outside the timed region)
        stats = w.trigger_query_stats() if trig is not None else (0, 0)
    return ms, stats, events


def main():
    n_trig = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n = int(sys.argv[2]) if len(sys.argv) > 2 else None
    wl = synth.config("cube4m", n=n)
    rng = np.random.default_rng(3)
    ent = rng.choice(wl.n, n_trig, replace=False).astype(np.uint32)
    size = rng.uniform(0.5, 3.0, (n_trig, 3)).astype(np.float32)
    trig = (ent, np.zeros(n_trig, np.uint8), size, np.full(n_trig, 4, np.uint32), np.full(n_trig, 0xFFFFFFFF, np.uint32),
            np.zeros(n_trig, np.uint8), np.ones(n_trig, np.uint8))
    base, _, _ = run(wl, None, 64)
    brute, sb, eb = run(wl, trig, 1 << 30)
    grid, sg, eg = run(wl, trig, 64)
    same = len(eb) == len(eg) and all(np.array_equal(a, b) for a, b in zip(eb, eg))
    n_events = sum(len(e) for e in eg)
    print(f"{wl.n} bodies, {n_trig} ghosts: tick without triggers {base:.3f} ms; all ghosts against all bodies {brute:.3f} ms "
          f"(+{brute - base:.3f}); through the grid {grid:.3f} ms (+{grid - base:.3f}; {sg[0]} ghosts walked the grid, {sg[1]} "
          f"were tested against every body); {n_events} events, identical in both runs: {same}")
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
