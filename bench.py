#!/usr/bin/env python3
"""bench.py — entity-updates/s of the ECS world tick (transform + physics) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-gather] [--no-cpu]

A "step" is one fixed-step tick of the whole (sharded) world: the fused integrate + transform-hierarchy
kernel over every tile, and for N > 1 the packing of this rank's root world matrices plus ONE RCCL
all-gather of them (BASELINE.json configs[4]).  Inputs are synthetic and resident in
HBM before the timed region.  Rank 0 prints one JSON line.

Workloads (BASELINE.json `configs`):
    N = 1 : "flat1m"    — configs[1]: 1,000,000 entities, flat, every entity a Dynamic body
    N > 1 : "subtree64" — configs[4] (variant 5b): 2,000,000 entities PER GPU in 64-node subtrees whose
            roots are Dynamic bodies; shards are whole subtrees; gather of 31,250 root matrices per rank per step.
            ("chains4_shard" = variant 5a, 500,000 roots per rank, is selectable with --workload.)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default=None, help="flat1m | chains4 | subtree64 | chains4_shard | cube4m | flat10k")
    ap.add_argument("--entities", type=int, default=None, help="entities per GPU (default: the configuration's size)")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the per-frame all-gather of roots")
    ap.add_argument("--force-gather", action="store_true", help="run the root all-gather even with one rank (rehearsal)")
    ap.add_argument("--collective", choices=["rccl", "torch"], default="rccl",
                    help="rccl: native ncclAllGather inside libbge_world (side stream, double-buffered); "
                         "torch: torch.distributed.all_gather_into_tensor")
    ap.add_argument("--no-overlap", action="store_true", help="issue the all-gather on the compute stream instead of "
                                                              "double-buffered on a side stream")
    ap.add_argument("--gather-mode", choices=["allgather", "direct"], default=None,
                    help="force the schedule of the native root gather (default: time both before the timed region, keep the faster)")
    ap.add_argument("--bullet-basis", action="store_true",
                    help="BGE_TICK_BULLET_BASIS: Bullet's basis round trip + euler rewrite for every Dynamic body each tick "
                         "(+44 B per body: quaternion in and out, rotationEuler out); default: zero-spin bodies keep theirs")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the cpu_baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the all-core CPU leg (a 1-GPU box's CPU share)")
    return ap.parse_args()


def cpu_baseline(wl, target_seconds, aabbs=False, max_entities=None):
    """Time the reference-faithful CPU path (oracle port, 1 thread) on a bounded sample of the same workload.
    (`aabbs`, `max_entities`: used by tools/measure_configs.py, which times the other configurations through this leg.)"""
    from oracle import pyoracle as po
    n = min(wl.n, max_entities) if max_entities else wl.n
    roots_only = int(wl.bodies_on_roots_only)
    # probe one tick to size the sample, then time `ticks` ticks after one warm tick
    sec, _ = po.bench_tick(wl.shape, wl.pos_box, roots_only, int(aabbs), n, wl.seed, 0, 1)
    ticks = int(max(2, min(120, target_seconds / max(sec, 1e-9))))
    sec, upd = po.bench_tick(wl.shape, wl.pos_box, roots_only, int(aabbs), n, wl.seed, 1, ticks)
    return {
        "entities": n, "ticks": ticks,
        "value": upd * ticks / sec,
        "unit": "entity-updates/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{n} entities x {ticks} ticks of the same workload ({wl.name}); hash-map AoS restatement of "
                  "PhysicsSystem::Update (free bodies) + TransformSystem::Update, 1 thread; the reference itself "
                  "does not build here (bx/bgfx/Bullet absent)",
    }


def cpu_allcore(wl, target_seconds, threads, max_entities=None):
    """BASELINE.md §3 "CPU-opt": the same arithmetic on dense SoA, depth-sorted, all host threads (oracle/soa_ref.h)."""
    from oracle import pyoracle as po
    roots_only = int(wl.bodies_on_roots_only)
    n = min(wl.n, max_entities) if max_entities else wl.n
    sec, threads = po.bench_tick_soa(wl.shape, wl.pos_box, roots_only, n, wl.seed, 1, 2, threads=threads)
    ticks = int(max(3, min(400, target_seconds / max(sec / 2, 1e-9))))
    sec, threads = po.bench_tick_soa(wl.shape, wl.pos_box, roots_only, n, wl.seed, 2, ticks, threads=threads)
    return {
        "value": n * ticks / sec,
        "unit": "entity-updates/s",
        "cores": threads,
        "kind": "port-soa",
        "sample": f"{n} entities x {ticks} ticks ({wl.name}); dense structure-of-arrays rewrite of the same arithmetic, "
                  f"OpenMP on {threads} threads — not how the reference works, reported for a hardware-to-hardware reading",
    }


def load_traffic(workload_name, entities):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/traffic.json), or None.

    The summary holds bytes per entity measured for each workload shape ((2 x FETCH_SIZE + WRITE_SIZE) x 1024 / entities,
    gfx950 correction included); it is scaled to this run's entity count."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        per_entity = t[workload_name]["per_entity"]["total"]
        return per_entity * entities
    except Exception:
        return None


def main():
    args = parse_args()
    # stdout carries exactly ONE line, the result.  Libraries print there too (RCCL writes its version banner to fd 1
    # when the first communicator is created), so fd 1 is pointed at stderr for the duration of the run and the JSON line
    # goes to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size and world_size == 1 and args.gpus > 1:
        print(f"bench.py --gpus {args.gpus} must be launched with torch.distributed.run "
              f"(--nproc-per-node {args.gpus}); running 1 rank", file=sys.stderr)
    n_gpus = world_size
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    if world_size > 1 or args.force_gather:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world_size,
                                device_id=torch.device("cuda", local_rank))

    import banggameengine_amd as B
    from banggameengine_amd import synth
    from banggameengine_amd.world import FIXED_DT, GRAVITY

    name = args.workload or ("flat1m" if n_gpus == 1 else "subtree64")
    base = synth.config(name)
    per_gpu = args.entities or base.n
    if base.shape == synth.SUBTREE64:
        per_gpu -= per_gpu % 64
    elif base.shape == synth.CHAINS4:
        per_gpu -= per_gpu % 4
    wl = synth.config(name, n=per_gpu, first=rank * per_gpu)  # global entity numbering: shard r = [r*per_gpu, (r+1)*per_gpu)
    flags = B.TICK_ALL | (B.TICK_BROADPHASE if name == "cube4m" else 0) | (B.TICK_BULLET_BASIS if args.bullet_basis else 0)

    # a dedicated (non-null) torch stream carries the world's kernels, the events and the collective
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    world = B.World(device=local_rank, stream=stream.cuda_stream)
    world.load(wl)
    world.tick(dt=FIXED_DT, flags=B.TICK_ALL)  # first tick creates the bodies (zero velocity), as in the reference
    world.set_velocities(wl.vel)            # synthetic initial velocities
    info = world.info()

    gather = (n_gpus > 1 and not args.no_gather) or args.force_gather
    n_roots = info["n_roots"]
    collective = "none"
    if gather:
        rows = torch.tensor([n_roots], dtype=torch.int64, device="cuda")
        dist.all_reduce(rows, op=dist.ReduceOp.MAX)
        rows_per_rank = int(rows.item())
        native = args.collective == "rccl"
        if native:
            ok = 1
            try:
                uid = [B.World.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                world.comm_init(n_gpus, rank, uid[0], rows_per_rank)
            except Exception as e:  # transport choice only: the torch path moves the same bytes
                print(f"[bench] rank {rank}: native RCCL init failed ({e})", file=sys.stderr)
                ok = 0
            # the choice of transport must be the same on every rank, or the first collective deadlocks
            agreed = torch.tensor([ok], dtype=torch.int64, device="cuda")
            dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
            if int(agreed.item()) == 1:
                collective = (f"native RCCL gather (libbge_world) of {rows_per_rank} root world matrices per rank per step as "
                              "48-B rows (4x3: a root's fourth column is exactly 0,0,0,1), side stream, ring of 8 buffers")
            else:
                if ok:
                    world.comm_destroy()
                if rank == 0:
                    print("[bench] using torch.distributed for the root gather on every rank", file=sys.stderr)
                native = False
        if not native:
            from banggameengine_amd.sharding import RootTable
            roots = RootTable(n_roots, torch.device("cuda", local_rank), overlap=not args.no_overlap)
            collective = (f"torch.distributed.all_gather_into_tensor (RCCL) of {roots.rows} x 64 B root world matrices per "
                          f"rank per step, {'side stream, double-buffered' if roots.overlap else 'compute stream'}")

    def gather_roots():
        # frame t's roots: packed on the compute stream, gathered on the side stream under frame t+1's tick
        if native:
            world.gather_roots()
        else:
            world.pack_roots(roots.send_buffer().data_ptr())
            roots.gather()

    def step():
        world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=flags)
        if gather:
            gather_roots()

    def barrier():
        if gather:
            world.comm_wait() if native else roots.finish()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    gather_trials = None
    if gather and native and not args.gather_mode:
        # Which schedule moves the root table faster — one ncclAllGather, or a direct send/recv per peer over the xGMI
        # mesh — depends on the RCCL build, the rank count and the message size: measure both on THIS node (untimed
        # tuning, like the warm-up), keep the faster one on every rank (the decision is an all-reduced number).
        gather_trials = {}
        for mode in (0, 1):
            world.comm_set_mode(mode)
            world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=flags | B.TICK_GATHER_ROOTS, ticks=8)
            barrier()
            t_a = time.perf_counter()
            world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=flags | B.TICK_GATHER_ROOTS, ticks=40)
            barrier()
            el = torch.tensor([time.perf_counter() - t_a], dtype=torch.float64, device="cuda")
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            gather_trials[mode] = float(el.item()) / 40 * 1e3
        best = min(gather_trials, key=gather_trials.get)
        world.comm_set_mode(best)
        world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=flags | B.TICK_GATHER_ROOTS, ticks=8)
        barrier()
        collective += (f"; schedule {'direct send/recv per peer' if best else 'ncclAllGather'} chosen by trial "
                       f"(ms per frame: all-gather {gather_trials[0]:.4f}, direct {gather_trials[1]:.4f})")
    elif gather and native and args.gather_mode:
        world.comm_set_mode({"allgather": 0, "direct": 1}[args.gather_mode])
        collective += f"; schedule forced: {args.gather_mode}"
    # Timed region: exactly K steps, no per-step host work.  Without a collective the K tick launches are
    # bracketed by ONE HIP event pair recorded by the library on the launch stream (bge_world_profile_enable(1)),
    # so the roofline figure is the average launch duration inside the timed region, gaps included.
    world.profile_enable(0 if gather else 1)
    t0 = time.perf_counter()
    if not gather:
        world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=flags, ticks=args.steps)
    elif native:
        # one native call enqueues K frames: tick kernels, root packing, ncclAllGather on the side stream
        world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=flags | B.TICK_GATHER_ROOTS, ticks=args.steps)
    else:
        for _ in range(args.steps):
            step()
    barrier()
    t1 = time.perf_counter()
    kernel_note = "one HIP event pair around the K launches of the timed region"
    if gather:
        # the collective and the packing kernel share the stream with the tick kernels: time the tick kernels in a
        # second, instrumented pass of the same K steps (one event pair per tick)
        world.profile_enable(2)
        if native:
            world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=flags | B.TICK_GATHER_ROOTS, ticks=args.steps)
        else:
            for _ in range(args.steps):
                step()
        barrier()
        kernel_note = "event pair per tick in a second, instrumented pass of the same K steps (collective interleaved)"
    kernel_total_ms, kernel_ticks = world.profile_read()
    world.profile_enable(0)
    assert kernel_ticks == args.steps, (kernel_ticks, args.steps)
    kernel_ms = kernel_total_ms / kernel_ticks

    gather_check = None
    if gather and native:
        # Correctness of the collective (outside the timed region): one more frame, then every rank compares EVERY
        # rank's segment of its gathered table with a checksum that rank computed from its own root matrices.
        world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=flags | B.TICK_GATHER_ROOTS)
        table_host = world.download_gathered(n_gpus, rows_per_rank)
        slot_world = world.download_world()
        root_ids = np.flatnonzero(wl.parent == 0xFFFFFFFF)
        mine = np.zeros((rows_per_rank, 16), np.float32)
        mine[: len(root_ids)] = slot_world[root_ids]
        own_ok = np.array_equal(table_host[rank].view(np.uint32), mine.view(np.uint32))
        sums = torch.tensor([int(mine.view(np.uint32).astype(np.uint64).sum() & 0x7FFFFFFFFFFFFFFF)], dtype=torch.int64, device="cuda")
        all_sums = [torch.zeros_like(sums) for _ in range(n_gpus)]
        dist.all_gather(all_sums, sums)
        seen = [int(table_host[r].view(np.uint32).astype(np.uint64).sum() & 0x7FFFFFFFFFFFFFFF) for r in range(n_gpus)]
        ok = own_ok and all(int(all_sums[r].item()) == seen[r] for r in range(n_gpus))
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_check = "ok: every rank holds every rank's root matrices bit for bit" if int(flag.item()) == 1 else "MISMATCH"
        if gather_check == "MISMATCH":
            print(f"[bench] rank {rank}: gathered root table does not match (own segment ok: {own_ok})", file=sys.stderr)

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
    if dist.is_initialized():
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed_s = float(elapsed.item())
    total_entities = per_gpu * n_gpus
    value = total_entities * args.steps / elapsed_s

    if rank == 0:
        bytes_per_update = wl.bytes_per_update
        kernel_name = "k_tick<physics,transforms>"
        if name == "cube4m":
            # 208 B + 8 B per emitted pair; the event pair brackets the tick kernel AND the
            # broadphase kernels of every step, so both sides of the ratio cover the whole step
            bytes_per_update = 208.0 + 8.0 * world.pair_count() / per_gpu
            kernel_name = "k_tick<physics,transforms,aabb> + broadphase (sort, pair search)"
        if args.bullet_basis:
            # every Dynamic body also reads and writes its quaternion (16 + 16 B) and writes rotationEuler (12 B)
            bytes_per_update += 44.0 * float((wl.body_type == 1).mean())
            kernel_name += " [bullet basis]"
        alg_bytes = bytes_per_update * per_gpu  # per step on one GPU
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "entity-updates/sec (transform+physics tick)",
            "value": value,
            "unit": "entity-updates/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_s / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{name}: {per_gpu} entities/GPU x {n_gpus} GPU, "
                            + {"flat1m": "flat hierarchy, every entity a Dynamic body (BASELINE configs[1])",
                               "subtree64": "64-node subtrees (1+3+12+48), roots Dynamic, sharded by subtree "
                                            "(BASELINE configs[4], variant 5b)",
                               "chains4_shard": "depth-4 chains, roots Dynamic, sharded by subtree (configs[4], 5a)",
                               "chains4": "depth-4 chains, roots Dynamic (configs[2])",
                               "cube4m": "flat + AABB broadphase (configs[3])",
                               "flat10k": "flat 10k (configs[0])"}[name],
                "entities_per_gpu": per_gpu,
                "tiles": info["n_tiles"], "passes": info["n_passes"], "roots_per_gpu": n_roots,
                "collective": collective,
                "gather_check": gather_check,
                "gather_schedule_trials_ms": gather_trials,
                "dt": FIXED_DT, "gravity": list(GRAVITY),
                "orientation": "bullet basis round trip every tick" if args.bullet_basis
                               else "quaternion state; a body with zero angular velocity keeps its orientation (DESIGN.md 4.2)",
                "bytes_per_update_algorithmic": bytes_per_update,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kernel_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": load_traffic(name, per_gpu),
                "kernel_ms_per_launch": kernel_ms,
                "timing": kernel_note,
                "algorithmic_bytes_per_launch": alg_bytes,
            },
        }
        if n_gpus == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds)
            out["cpu_allcore"] = cpu_allcore(wl, args.cpu_seconds / 2, args.cpu_threads)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())

    if gather and native:
        world.comm_destroy()
    world.close()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
