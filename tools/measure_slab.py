#!/usr/bin/env python3
"""Cost of the cross-shard (slab) broadphase, rehearsed on ONE MI355X: the scene of BASELINE configs[3] (4 M bodies in a
262 m cube) is split over `nshards` worlds on the same GPU (entity i -> shard i % nshards: interleaved in space, like a
subtree partition), each world updates its AABBs (BGE_TICK_AABBS), then the slab exchange runs with device-to-device copies in
place of the all-to-all.  Reported per phase, summed over the shards (they would run concurrently on separate GPUs):
cuts (bounds + histogram), route (count), pack, find (unpack + sort + windowed pair search); plus the bytes that would cross
xGMI and the pair totals, against the single-world broadphase of the same scene.  Writes gpurun_out/slab.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import banggameengine_amd as B  # noqa: E402
from banggameengine_amd import sharding, synth  # noqa: E402
from banggameengine_amd.world import FIXED_DT, balanced_cuts  # noqa: E402


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, (time.perf_counter() - t0) * 1e3


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    rows = []
    wl = synth.config("cube4m", n=n)
    # single world: the reference point
    with B.World() as w:
        w.load(wl)
        w.tick(dt=FIXED_DT)
        w.set_velocities(wl.vel)
        w.tick(dt=FIXED_DT, flags=B.TICK_ALL | B.TICK_BROADPHASE, ticks=14)   # 15 steps in all, like the shards below
        single_pairs = w.pair_count()
        _, ms = timed(lambda: w.tick(dt=FIXED_DT, flags=B.TICK_ALL | B.TICK_BROADPHASE, ticks=10))
        _, ms_aabb = timed(lambda: w.tick(dt=FIXED_DT, flags=B.TICK_ALL | B.TICK_AABBS, ticks=10))
    single = dict(n=n, ms_tick_with_broadphase=ms / 10, ms_tick_with_aabbs_only=ms_aabb / 10, pairs=single_pairs)
    print(f"# single world: {single}", file=sys.stderr, flush=True)

    for nshards in (2, 4, 8):
        worlds, ids_of = [], []
        for r in range(nshards):
            ids = np.arange(r, n, nshards, dtype=np.uint32)
            w = B.World(pair_capacity=max(16 * len(ids), 1 << 20))
            w.set_topology(np.full(len(ids), 0xFFFFFFFF, np.uint32))
            w.upload_trs(wl.pos[ids], wl.euler[ids], wl.scale[ids])
            w.upload_bodies(wl.body_type[ids])
            w.set_global_ids(ids)
            w.tick(dt=FIXED_DT)
            w.set_velocities(wl.vel[ids])
            for _ in range(14):   # same number of steps as the single world above
                w.tick(dt=FIXED_DT, flags=B.TICK_ALL | B.TICK_AABBS)
            worlds.append(w)
            ids_of.append(ids)
        axis = 2
        phase = dict(cuts=0.0, route=0.0, pack=0.0, find=0.0)
        for rep in range(3):   # last repetition is reported (buffers are sized by then)
            phase = dict(cuts=0.0, route=0.0, pack=0.0, find=0.0)
            (b, hists), ms = timed(lambda: ([w.aabb_bounds() for w in worlds], None))
            lo = min(float(x[0][axis]) for x in b)
            hi = max(float(x[1][axis]) for x in b)
            hist, ms2 = timed(lambda: sum(w.axis_histogram(axis, lo, hi) for w in worlds))
            cuts = balanced_cuts(hist, lo, hi, nshards)
            phase["cuts"] = ms + ms2
            counts, ms = timed(lambda: np.stack([w.bp_route(axis, cuts) for w in worlds]).astype(np.int64))
            phase["route"] = ms
            sends = [torch.empty(max(int(counts[s].sum()), 1) * sharding.RECORD_FLOATS, dtype=torch.float32, device="cuda")
                     for s in range(nshards)]
            _, ms = timed(lambda: [w.bp_pack(t.data_ptr()) for w, t in zip(worlds, sends)])
            phase["pack"] = ms
            recvs = []
            for d in range(nshards):
                parts = []
                for s in range(nshards):
                    off = int(counts[s, :d].sum()) * sharding.RECORD_FLOATS
                    parts.append(sends[s][off: off + int(counts[s, d]) * sharding.RECORD_FLOATS])
                recvs.append(torch.cat(parts))

            def find_all():
                for d, w in enumerate(worlds):
                    wlo, whi = sharding.slab_window(cuts, d)
                    w.bp_find(recvs[d].data_ptr(), int(counts[:, d].sum()), axis, wlo, whi)
            _, ms = timed(find_all)
            phase["find"] = ms
        pairs = sum(w.pair_count() for w in worlds)
        sent = int(counts.sum())
        crossing = int(counts.sum() - np.trace(counts))
        rows.append(dict(nshards=nshards, bodies_per_shard=n // nshards, records_sent=sent, records_per_body=sent / n,
                         bytes_crossing_links=crossing * 48, pairs=pairs, pairs_match_single_world=bool(pairs == single_pairs),
                         per_shard_ms={k: v / nshards for k, v in phase.items()}, summed_ms=phase))
        print(f"# {nshards} shards: {rows[-1]}", file=sys.stderr, flush=True)
        for w in worlds:
            w.close()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(single=single, slabs=rows), open(os.path.join(ROOT, "gpurun_out", "slab.json"), "w"), indent=1)
    print("| shards | bodies / shard | records / body | bytes over links | cuts | route | pack | find | pairs == single world |")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        p = r["per_shard_ms"]
        print(f"| {r['nshards']} | {r['bodies_per_shard']:,} | {r['records_per_body']:.3f} | {r['bytes_crossing_links']/1e6:.0f} MB | "
              f"{p['cuts']:.3f} | {p['route']:.3f} | {p['pack']:.3f} | {p['find']:.3f} | {r['pairs_match_single_world']} |")
    print(f"\nsingle world, {n:,} bodies: {single['ms_tick_with_broadphase']:.3f} ms per tick with the broadphase, "
          f"{single['ms_tick_with_aabbs_only']:.3f} ms with AABBs only, {single['pairs']:,} pairs")


if __name__ == "__main__":
    main()
