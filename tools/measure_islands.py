#!/usr/bin/env python3
"""Cost of the Dynamic-against-Dynamic contact path (bge_world_set_dynamic_contacts; bge_island.hip) at scale, per tick:

  free      n Dynamic boxes in free fall, nobody near anybody: what the sub-step's pair search and its two read-backs cost on top of
            the tick (with the switch off for comparison)
  stacks    n / 2 two-box stacks resting on the plane: n / 2 pairs, n / 2 islands of two
  towers    n / 10 towers of ten: 0.9 n pairs, islands of ten
  heap      ONE pile of h boxes dropped into a pit: a single island — the workgroup solver that keeps Bullet's row order level by level

Run on the GPU box:  python tools/measure_islands.py [n] [h]   (defaults 200000, 2000)
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


def run(name, pos, size, dynamic, warm, ticks, stream):
    n = len(pos)
    wl = synth.config("flat1m", n=n)
    wl.pos[:] = pos
    wl.euler[:] = 0.0
    wl.body_type[:] = 1
    with B.World(stream=stream.cuda_stream, pair_capacity=max(16 * n, 4096)) as w:
        w.set_topology(wl.parent)
        w.upload_trs(wl.pos, wl.euler, wl.scale)
        w.upload_bodies(wl.body_type, size=size, shape=np.zeros(n, np.uint8))
        w.set_ground_plane(True)
        w.set_dynamic_contacts(dynamic)
        w.tick(dt=FIXED_DT, flags=B.TICK_ALL, ticks=warm)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w.tick(dt=FIXED_DT, flags=B.TICK_ALL, ticks=ticks)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / ticks * 1e3
        pairs = len(w.download_dynamic_pairs()[0]) if dynamic else 0
        st, _ = w.download_activation()
        print(f"{name}: {n} bodies, {pairs} pairs, {int((st == 2).sum())} asleep: {ms:.4f} ms per tick", flush=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    only = sys.argv[3] if len(sys.argv) > 3 else ""          # "stacks", "towers", "heap", "free": that scenario alone (profiles)
    rng = np.random.default_rng(4)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    side = int(np.ceil(np.sqrt(n)))
    gx, gz = np.meshgrid(np.arange(side), np.arange(side))
    cell = np.stack([gx.ravel(), gz.ravel()], 1)[:n].astype(np.float32) * 4.0
    half = np.full((n, 3), 0.5, np.float32)
    # free fall
    pos = np.zeros((n, 3), np.float32)
    pos[:, 0], pos[:, 2] = cell[:, 0], cell[:, 1]
    pos[:, 1] = rng.uniform(500, 600, n).astype(np.float32)
    if only in ("", "free"):
        run("free fall, dynamic contacts off", pos, half, False, 5, 50, stream)
        run("free fall, dynamic contacts on ", pos, half, True, 5, 50, stream)
    # two-box stacks
    pos = np.zeros((n, 3), np.float32)
    k = np.arange(n)
    pos[:, 0], pos[:, 2] = cell[k // 2, 0], cell[k // 2, 1]
    pos[:, 1] = 0.5 + 1.0 * (k % 2)
    if only in ("", "stacks"):
        run("two-box stacks                 ", pos, half, True, 30, 50, stream)
    # towers of ten
    pos[:, 0], pos[:, 2] = cell[k // 10, 0], cell[k // 10, 1]
    pos[:, 1] = 0.5 + 1.0 * (k % 10)
    if only in ("", "towers"):
        run("towers of ten                  ", pos, half, True, 30, 50, stream)
    # one heap
    pos = np.zeros((h, 3), np.float32)
    pos[:, 0] = rng.uniform(-4, 4, h)
    pos[:, 2] = rng.uniform(-4, 4, h)
    pos[:, 1] = rng.uniform(0.5, 0.5 + h / 40.0, h)
    hs = rng.uniform(0.2, 0.5, (h, 3)).astype(np.float32)
    if only in ("", "heap"):
        run("one heap                       ", pos, hs, True, 120, 20, stream)
    return 0


if __name__ == "__main__":
    sys.exit(main())
