#!/usr/bin/env python3
"""Measure every BASELINE.json configuration on ONE MI355X, with the CPU port timed beside it.

Prints a markdown table (pasted into DESIGN.md §6) and writes gpurun_out/configs.json.  The timed region is the
device-resident tick (inputs in HBM); two PCIe-inclusive variants are reported separately:
  +D2H  the tick followed by a download of every world matrix (64 B/entity), as the coherent C++ adapter does;
  +H2D  an upload of all TRS (36 B/entity, marks everything dirty) followed by the tick.
"""
import json
import os
import sys
import time


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import banggameengine_amd as B  # noqa: E402
from banggameengine_amd import synth  # noqa: E402
from banggameengine_amd.world import FIXED_DT  # noqa: E402
import bench  # noqa: E402  (the CPU leg goes through bench.py's cpu_baseline / cpu_allcore, the only timed users of oracle/)

CONFIGS = [
    ("configs[0] flat 10k", "flat10k", None, 2000),
    ("configs[1] flat 1M", "flat1m", None, 500),
    ("configs[2] chains-of-4 1M", "chains4", None, 500),
    ("configs[3] flat 4M + broadphase", "cube4m", None, 20),
    ("configs[4] 5a chains-of-4, 2M/GPU shard", "chains4_shard", None, 300),
    ("configs[4] 5b 64-node subtrees, 2M/GPU shard", "subtree64", None, 300),
    ("flat 16M (beyond Infinity Cache)", "flat1m", 16_000_000, 100),
]


def gpu_rate(name, n, steps, extra=None):
    wl = synth.config(name, n=n)
    flags = B.TICK_ALL | (B.TICK_BROADPHASE if name == "cube4m" else 0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    w = B.World(stream=stream.cuda_stream)
    w.load(wl)
    w.tick(dt=FIXED_DT)
    w.set_velocities(wl.vel)
    w.tick(dt=FIXED_DT, flags=flags, ticks=10)
    torch.cuda.synchronize()
    w.profile_enable(1)
    t0 = time.perf_counter()
    w.tick(dt=FIXED_DT, flags=flags, ticks=steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms, ticks = w.profile_read()
    w.profile_enable(0)
    out = dict(n=wl.n, steps=steps, wall_rate=wl.n * steps / dt, stream_ms_per_tick=ms / ticks,
               bytes_per_update=wl.bytes_per_update)
    if name == "cube4m":
        pairs = w.pair_count()
        out["pairs_per_entity"] = pairs / wl.n
        out["bytes_per_update"] = 208.0 + 8.0 * pairs / wl.n
    if extra:
        k = max(3, min(20, steps // 10))
        t0 = time.perf_counter()
        for _ in range(k):
            w.tick(dt=FIXED_DT, flags=flags)
            w.download_world()
        out["rate_with_d2h"] = wl.n * k / (time.perf_counter() - t0)
        from banggameengine_amd.world import PinnedArray
        pinned = PinnedArray((wl.n, 16))
        t0 = time.perf_counter()
        for _ in range(k):
            w.tick(dt=FIXED_DT, flags=flags)
            w.download_world(out=pinned.array)
        out["rate_with_d2h_pinned"] = wl.n * k / (time.perf_counter() - t0)
        t0 = time.perf_counter()
        for _ in range(k):
            w.upload_trs(wl.pos, wl.euler, wl.scale)
            w.tick(dt=FIXED_DT, flags=B.TICK_TRANSFORMS)
        torch.cuda.synchronize()
        out["rate_with_h2d"] = wl.n * k / (time.perf_counter() - t0)
    w.close()
    return wl, out


def cpu_rate(wl, name, seconds=6.0):
    aabb = name == "cube4m"
    # bounded sample: the hash-map port runs ~2-10 M updates/s
    c = bench.cpu_baseline(wl, seconds, aabbs=aabb, max_entities=1_000_000)
    out = dict(n=c["entities"], ticks=c["ticks"], rate=c["value"])
    if not aabb:
        a = bench.cpu_allcore(wl, 3.0, 0, max_entities=1_000_000)
        out.update(soa_rate=a["value"], soa_threads=a["cores"])
    return out


def main():
    rows = []
    for label, name, n, steps in CONFIGS:
        wl, g = gpu_rate(name, n, steps, extra=name in ("flat1m", "flat10k"))
        c = cpu_rate(wl, name)
        g.update(label=label, cpu=c)
        rows.append(g)
        print(f"# {label}: {g['wall_rate']/1e9:.2f} G/s", file=sys.stderr, flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "configs.json"), "w"), indent=1)
    print("| config | entities | GPU entity-updates/s | ms/tick (stream) | algorithmic B/update | achieved GB/s | frac of 8 TB/s | CPU port (1 core) | GPU/CPU port | CPU SoA all cores |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        gbs = r["bytes_per_update"] * r["n"] / (r["stream_ms_per_tick"] * 1e-3) / 1e9
        print(f"| {r['label']} | {r['n']:,} | {r['wall_rate']/1e9:.2f} G | {r['stream_ms_per_tick']:.4f} | "
              f"{r['bytes_per_update']:.1f} | {gbs:,.0f} | {gbs/8000:.3f} | {r['cpu']['rate']/1e6:.2f} M "
              f"({r['cpu']['n']:,} x {r['cpu']['ticks']}) | {r['wall_rate']/r['cpu']['rate']:,.0f}x | "
              + (f"{r['cpu']['soa_rate']/1e6:.0f} M ({r['cpu']['soa_threads']} thr)" if 'soa_rate' in r['cpu'] else "—") + " |")
    for r in rows:
        if "rate_with_d2h" in r:
            print(f"\nPCIe-inclusive, {r['label']}: tick + D2H of all world matrices {r['rate_with_d2h']/1e6:.0f} M updates/s "
                  f"({r['rate_with_d2h_pinned']/1e6:.0f} M into page-locked memory); "
                  f"H2D of all TRS + transform tick {r['rate_with_h2d']/1e6:.0f} M updates/s")


if __name__ == "__main__":
    main()
