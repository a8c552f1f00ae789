#!/usr/bin/env python3
"""Cost of the Static-collider candidate test at scale: n Dynamic boxes in free fall (awake, far from everything) and K Static boxes elsewhere.
k_ground_select tests every awake Dynamic box against every obstacle's fed AABB (bge_contact.hip) — linear in K per body.

Run on the GPU box:  python tools/measure_obstacles.py [n_dynamic] ; prints ms per tick for K = 0, 10, 100, 1000, 10000.
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


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    rng = np.random.default_rng(2)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    for k in (0, 10, 100, 1000, 10000):
        wl = synth.config("flat1m", n=n + k)
        wl.pos[:n, 0] = rng.uniform(-3000, 3000, n).astype(np.float32)
        wl.pos[:n, 2] = rng.uniform(-3000, 3000, n).astype(np.float32)
        wl.pos[:n, 1] = rng.uniform(500, 600, n).astype(np.float32)
        wl.pos[n:, 0] = rng.uniform(-3000, 3000, k).astype(np.float32)
        wl.pos[n:, 2] = rng.uniform(-3000, 3000, k).astype(np.float32)
        wl.pos[n:, 1] = 2.0
        wl.body_type[:n] = 1
        wl.body_type[n:] = 0
        size = rng.uniform(0.2, 0.9, (n + k, 3)).astype(np.float32)
        with B.World(stream=stream.cuda_stream) as w:
            w.set_topology(wl.parent)
            w.upload_trs(wl.pos, wl.euler, wl.scale)
            w.upload_bodies(wl.body_type, size=size, shape=np.zeros(n + k, np.uint8))
            w.set_ground_plane(True)
            w.set_static_contacts(k > 0)
            w.tick(dt=FIXED_DT, flags=B.TICK_ALL, ticks=5)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            w.tick(dt=FIXED_DT, flags=B.TICK_ALL, ticks=50)
            torch.cuda.synchronize()
            print(f"{n} Dynamic boxes in free fall, {k} Static boxes: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms per tick", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
