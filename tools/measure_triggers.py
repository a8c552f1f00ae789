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


def run(wl, trig, grid_min, ticks=8, device_diff=True, stay=True):
    os.environ["BGE_TRIGGER_GRID_MIN"] = str(grid_min)
    os.environ["BGE_TRIGGER_DEVICE_DIFF"] = "1" if device_diff else "0"
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    flags = B.TICK_ALL | B.TICK_BROADPHASE
    with B.World(stream=stream.cuda_stream) as w:
        w.load(wl)
        if trig is not None:
            w.upload_triggers(*trig)
            w.set_trigger_stay_events(stay)
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
        if trig is not None:
            stats = stats + w.trigger_diff_stats()
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
    brute, sb, eb = run(wl, trig, 1 << 30, device_diff=False)
    host, sh, eh = run(wl, trig, 64, device_diff=False)
    grid, sg, eg = run(wl, trig, 64)
    lean, sl, el = run(wl, trig, 64, stay=False)
    same = len(eb) == len(eg) == len(eh) and all(np.array_equal(a, b) and np.array_equal(a, c) for a, b, c in zip(eb, eg, eh))
    # without Stay records: what is left must be the Enter / Exit records of the full list, in the same order
    changes = all(np.array_equal(a[a[:, 0] != 1], b) for a, b in zip(eg, el))
    n_events = sum(len(e) for e in eg)
    print(f"{wl.n} bodies, {n_trig} ghosts: tick without triggers {base:.3f} ms; all ghosts against all bodies, host diff {brute:.3f} ms "
          f"(+{brute - base:.3f}); through the grid: host diff {host:.3f} ms (+{host - base:.3f}), device diff {grid:.3f} ms (+{grid - base:.3f}; "
          f"{sg[2]} ticks on the device, {sg[3]} on the host), device diff without Stay records {lean:.3f} ms (+{lean - base:.3f}; {sl[4]} Stay "
          f"events left out); {sg[0]} ghosts walked the grid, {sg[1]} were tested against every body; {n_events} events, identical in the "
          f"three full runs: {same}; Enter / Exit of the lean run identical: {changes}")
    return 0 if same and changes else 1


if __name__ == "__main__":
    sys.exit(main())
