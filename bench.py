#!/usr/bin/env python3
"""bench.py — entity-updates/s of the ECS world tick (transform + physics) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-gather] [--no-cpu] [--no-configs]

A "step" is one fixed-step tick of the whole (sharded) world: the fused integrate + transform-hierarchy
kernel over every tile, and for N > 1 the packing of this rank's root world matrices plus ONE RCCL
all-gather of them (BASELINE.json configs[4]).  Inputs are synthetic and resident in
HBM before the timed region.  Rank 0 prints one JSON line.

Workloads (BASELINE.json `configs`):
    N = 1 : "flat1m"    — configs[1]: 1,000,000 entities, flat, every entity a Dynamic body (the headline line); the
            same run then measures the other single-GPU configurations into the line's `configs` array: chains4
            (configs[2]), cube4m (configs[3], with the broadphase), flat 16 M (the working set beyond the Infinity
            Cache: the honest HBM figure) and flat1m in Bullet's own orientation scheme.
    N > 1 : "subtree64" — configs[4] (variant 5b): 2,000,000 entities PER GPU in 64-node subtrees whose
            roots are Dynamic bodies; shards are whole subtrees; gather of 31,250 root matrices per rank per step.  The line
            carries its own anchor: `single_gpu_same_workload` (rank 0's shard ticked alone, before the communicator exists),
            `scaling_efficiency` = value / (N x that), and compute-only against compute + collective ms per step.
            ("chains4_shard" = variant 5a, 500,000 roots per rank, is selectable with --workload.)

Launching.  `python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks ITSELF:
the parent process spawns N children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, 127.0.0.1), never touches the
GPU, relays rank 0's JSON line and exits non-zero when any child does or when the line does not say n_gpus == N.
Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` every process is a rank as before.  A rank
whose WORLD_SIZE disagrees with --gpus refuses to run.  Every phase that can block on a peer (process-group set-up,
communicator init, the schedule trial, the timed region, the gather check) runs under a host-side deadline: a rank
that is still inside the phase when it expires prints which phase (and collective) it was in and exits with status 3,
which brings the whole job down instead of hanging it until the driver's limit.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BGE_TICK_BULLET_BASIS: Bullet reads and rewrites basis and rotationEuler of every Dynamic body each step, mostly with the bits
# they already hold.  The kernel touches quaternion (16 B in, 16 B out) and rotationEuler (12 B out) only for the bodies the
# step can change: 12 % of a scene of non-spinning bodies in steady state (limit cycles of the normalisation, DESIGN.md 4.2;
# all of them in the first ticks after a re-pose, and every spinning body) — 0.12 x 44 B on top of the default scheme's bytes.
BASIS_EXTRA_BYTES = 5.3
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate
STUB = os.environ.get("BGE_BENCH_STUB") == "1"  # tests/test_bench_launcher.py only: control flow over gloo, no GPU work


def parse_args(argv=None):
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
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg and the sampled parity check")
    ap.add_argument("--no-configs", action="store_true", help="N = 1: skip the `configs` array (other single-GPU configurations)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the cpu_baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the all-core CPU leg (a 1-GPU box's CPU share)")
    ap.add_argument("--repeats", type=int, default=5,
                    help="the K-step timed region is run this many times back to back (each bracketed by barrier + synchronize, "
                         "max over ranks); the MEDIAN region is reported — a 20-launch region is 0.5 ms, one cold sample of it is noise")
    ap.add_argument("--deadline", type=float, default=180.0,
                    help="seconds any one blocking phase of a rank may take (and, x3 = 540 s, the launcher's limit for the whole job: "
                         "inside the driver's 600 s)")
    return ap.parse_args(argv)


# --------------------------------------------------------------------------------------------------------- launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Parent of an N-rank run.  Imports nothing that initialises the GPU; children are fresh interpreters."""
    n = args.gpus
    env0 = dict(os.environ)
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env0.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    def stop_all(why):
        print(f"[bench launcher] {why}: stopping all {n} ranks", file=sys.stderr, flush=True)
        for p in procs:
            if p.poll() is None:
                p.terminate()  # exactly the processes started above
        t_end = time.time() + 10
        for p in procs:
            with contextlib.suppress(Exception):
                p.wait(timeout=max(0.1, t_end - time.time()))
            if p.poll() is None:
                p.kill()

    limit = time.time() + 3 * args.deadline
    failed = None
    while failed is None:
        states = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(states) if c not in (None, 0)]
        if bad:
            failed = f"rank {bad[0][0]} exited with status {bad[0][1]}"
            break
        if all(c == 0 for c in states):
            break
        if time.time() > limit:
            failed = f"job still running after {3 * args.deadline:.0f} s"
            break
        time.sleep(0.1)
    if failed:
        stop_all(failed)
        return 1
    reader.join(timeout=10)
    text = (out0[0] if out0 else b"").decode(errors="replace")
    lines = [ln for ln in text.splitlines() if ln.strip()]
    try:
        line = json.loads(lines[-1])
    except Exception:
        print(f"[bench launcher] rank 0 printed no JSON result line (stdout: {text[-300:]!r})", file=sys.stderr)
        return 1
    if line.get("n_gpus") != n:
        print(f"[bench launcher] the result line says n_gpus = {line.get('n_gpus')}, expected {n}", file=sys.stderr)
        return 1
    sys.stdout.write(lines[-1] + "\n")
    sys.stdout.flush()
    return 0


class Deadlines:
    """Host-side watchdog: `with deadlines.phase("what", seconds)` — a rank still inside when the time is up says where
    it was and exits (status 3); the launcher / torchrun then tears the job down."""

    def __init__(self, rank, seconds):
        self.rank, self.seconds = rank, seconds

    @contextlib.contextmanager
    def phase(self, what, seconds=None):
        limit = seconds or self.seconds

        def expire():
            print(f"[bench] rank {self.rank}: DEADLINE — still in '{what}' after {limit:.0f} s; exiting", file=sys.stderr, flush=True)
            os._exit(3)

        t = threading.Timer(limit, expire)
        t.daemon = True
        t.start()
        try:
            yield
        finally:
            t.cancel()


# --------------------------------------------------------------------------------------------------------- CPU legs
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


def parity_sample(world, name, first, ticks_after_velocities, bullet_basis, sample=4096):
    """Part of the cpu_baseline leg (the checker, outside every timed region): the CPU port replays the run's whole tick
    sequence on the first `sample` entities of this rank's shard — whole subtrees; entities of different subtrees do not
    interact — and the device state is compared with it.  BASELINE.md §3: max relative error, norm-relative per matrix /
    position, tolerance 1e-5 (`north_star`); the kernels are built to be bit-identical, which is reported too."""
    import numpy as np
    from banggameengine_amd import synth
    from banggameengine_amd.world import FIXED_DT
    from oracle import pyoracle as po
    sample = min(sample - sample % 64, world.n - world.n % 64) or world.n
    wl = synth.config(name, n=sample, first=first)
    parent_i32 = np.where(wl.parent == 0xFFFFFFFF, -1, wl.parent.astype(np.int64)).astype(np.int32)
    ref = po.RefScene().bulk_build(parent_i32, wl.pos, wl.euler, wl.scale, body_type=wl.body_type)
    ref.SetPhysicsOptions(gravity_y=-9.81, orient_mode=po.ORIENT_BASIS if bullet_basis else po.ORIENT_IDEAL)
    ref.PhysicsSystemUpdate(FIXED_DT)
    ref.TransformSystemUpdate()
    ref.bulk_set_velocity(wl.vel)
    for _ in range(ticks_after_velocities):
        ref.PhysicsSystemUpdate(FIXED_DT)
        ref.TransformSystemUpdate()
    want_w, _ = ref.bulk_world()
    want_p, _ = ref.bulk_pose()
    ref.close()
    got_w = world.download_world(0, sample)
    got_p, _ = world.download_pose(0, sample)

    def rel(got, want):
        num = np.linalg.norm(got.astype(np.float64) - want.astype(np.float64), axis=1)
        den = np.maximum(np.linalg.norm(want.astype(np.float64), axis=1), 1e-30)
        return float(np.max(num / den)) if np.isfinite(num).all() else float("inf")

    err_w, err_p = rel(got_w, want_w), rel(got_p, want_p)
    return {
        "checker": "oracle port (CPU), same tick sequence",
        "sample_entities": int(sample), "ticks": int(ticks_after_velocities + 1),
        "max_rel_err_world": err_w, "max_rel_err_position": err_p, "tolerance": 1e-5,
        "bit_identical": bool(np.array_equal(got_w.view(np.uint32), want_w.view(np.uint32))
                              and np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32))),
        "ok": bool(err_w <= 1e-5 and err_p <= 1e-5),
    }


def load_traffic(workload_name, entities):
    """HBM bytes per launch from the committed rocprofv3 PMC summary (profiles/traffic.json) — NOT measured in this
    run (PMC passes need the profiler) — scaled to this run's entity count; (bytes, provenance) or (None, reason)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        per_entity = t[workload_name]["per_entity"]["total"]
        return per_entity * entities, t.get("_provenance", "profiles/traffic.json (committed rocprofv3 --pmc passes, 2 x FETCH_SIZE + WRITE_SIZE)")
    except Exception:
        return None, "no committed PMC measurement for this workload"


# ------------------------------------------------------------------------------------------------- one configuration
CONFIG_TEXT = {
    "flat1m": "flat hierarchy, every entity a Dynamic body (BASELINE configs[1])",
    "subtree64": "64-node subtrees (1+3+12+48), roots Dynamic, sharded by subtree (BASELINE configs[4], variant 5b)",
    "chains4_shard": "depth-4 chains, roots Dynamic, sharded by subtree (configs[4], 5a)",
    "chains4": "depth-4 chains, roots Dynamic (configs[2])",
    "cube4m": "flat + AABB broadphase (configs[3])",
    "flat10k": "flat 10k (configs[0])",
}


def measure_extra_config(B, synth, torch, label, name, entities, steps, warmup, bullet_basis, check, repeats=3):
    """One more single-GPU configuration, measured like the headline (event pair around the K launches) after it."""
    from banggameengine_amd.world import FIXED_DT, GRAVITY
    wl = synth.config(name, n=entities)
    flags = B.TICK_ALL | (B.TICK_BROADPHASE if name == "cube4m" else 0) | (B.TICK_BULLET_BASIS if bullet_basis else 0)
    stream = torch.cuda.current_stream()
    world = B.World(stream=stream.cuda_stream)
    try:
        world.load(wl)
        world.tick(dt=FIXED_DT, flags=B.TICK_ALL | (B.TICK_BULLET_BASIS if bullet_basis else 0))
        world.set_velocities(wl.vel)
        world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=flags, ticks=warmup)
        walls, kernels = [], []
        for _ in range(repeats):   # median of `repeats` regions of `steps` launches each, like the headline
            torch.cuda.synchronize()
            world.profile_enable(1)
            t0 = time.perf_counter()
            world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=flags, ticks=steps)
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
            ms, ticks = world.profile_read()
            assert ticks == steps
            kernels.append(ms / ticks)
        world.profile_enable(0)
        wall, kernel_ms = sorted(walls)[repeats // 2], sorted(kernels)[repeats // 2]
        bpu = wl.bytes_per_update
        out = {"label": label, "workload": name, "entities": wl.n, "steps": steps, "warmup": warmup, "repeats": repeats}
        if name == "cube4m":
            pairs = world.pair_count()
            bpu = 208.0 + 8.0 * pairs / wl.n
            out["pairs"] = pairs
        if bullet_basis:
            bpu += BASIS_EXTRA_BYTES * float((wl.body_type == 1).mean())
            out["orientation"] = "bullet basis round trip every tick"
        achieved = bpu * wl.n / (kernel_ms * 1e-3) / 1e9
        out.update({
            "value": wl.n * steps / wall, "unit": "entity-updates/s", "ms_per_step": wall / steps * 1e3,
            "kernel_ms_per_launch": kernel_ms, "bytes_per_update_algorithmic": bpu,
            "algorithmic_bytes_per_launch": bpu * wl.n, "achieved_gbs": achieved, "frac": achieved / HBM_PEAK_GBS,
        })
        if check:
            out["parity"] = parity_sample(world, name, 0, warmup + steps * repeats, bullet_basis)
        return out
    finally:
        world.close()


def measure_ground_config(B, synth, torch, entities, check):
    """configs[1] with the reference's ground plane (PhysicsSystem.cpp:149-166): n unit boxes dropped
    from just above y = 0.  Two timed regions on one world: all bodies RESTING awake on their contacts (collide + 10 solver
    iterations per body and tick), and ASLEEP after Bullet's 2 s deactivation time (the state a scene at rest is in).  Wall
    clock around a synchronised batch of ticks (three launches per tick: select, solve, tick).  The checker replays the whole
    tick sequence on the first 4,096 entities with the oracle's contact restatement."""
    import numpy as np
    from banggameengine_amd.world import FIXED_DT, GRAVITY
    wl = synth.config("flat1m", n=entities)
    rng = np.random.default_rng(11)
    wl.pos[:, 1] = (0.9 + rng.uniform(0.0, 0.3, wl.n)).astype(np.float32)   # default collider: a unit box (half extent 0.5 + margin)
    stream = torch.cuda.current_stream()
    world = B.World(stream=stream.cuda_stream)
    script = (("landing", 150, False), ("resting", 60, True), ("falling asleep", 400, False), ("asleep", 100, True))
    try:
        world.load(wl)
        world.set_ground_plane(True)
        phases = []
        for phase, ticks, timed in script:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=B.TICK_ALL, ticks=ticks)
            torch.cuda.synchronize()
            if timed:
                phases.append({"phase": phase, "ticks": ticks, "ms_per_step": (time.perf_counter() - t0) / ticks * 1e3})
        st, _ = world.download_activation()
        cn, _ = world.download_contacts()
        out = {"label": "configs[1] on the reference's ground plane (contacts + solver, SURVEY 8(f) rank 4)", "workload": "flat1m + ground plane",
               "entities": wl.n, "phases": phases, "asleep_at_the_end": int((st == 2).sum()),
               "contacts_per_body_histogram": np.bincount(cn, minlength=5).tolist(),
               "note": "wall clock per tick; the solver is bound by dependent arithmetic, not by HBM: no roofline fraction is quoted"}
        if check:
            from oracle import pyoracle as po
            sample = 4096
            swl = synth.config("flat1m", n=sample)
            swl.pos[:, 1] = wl.pos[:sample, 1]
            parent_i32 = np.where(swl.parent == 0xFFFFFFFF, -1, swl.parent.astype(np.int64)).astype(np.int32)
            ref = po.RefScene().bulk_build(parent_i32, swl.pos, swl.euler, swl.scale, body_type=swl.body_type)
            ref.SetPhysicsOptions(gravity_y=-9.81, orient_mode=po.ORIENT_IDEAL)
            ref.SetGroundPlane(True)
            n_ticks = sum(t for _, t, _ in script)
            for _ in range(n_ticks):
                ref.PhysicsSystemUpdate(FIXED_DT)
                ref.TransformSystemUpdate()
            want_w, _ = ref.bulk_world()
            want_p, _ = ref.bulk_pose()
            ref.close()
            got_w = world.download_world(0, sample)
            got_p, _ = world.download_pose(0, sample)
            same = bool(np.array_equal(got_w.view(np.uint32), want_w.view(np.uint32)) and np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32)))
            err = float(np.max(np.abs(got_p.astype(np.float64) - want_p.astype(np.float64))))
            out["parity"] = {"checker": "oracle port (CPU) with its contact restatement, same tick sequence", "sample_entities": sample,
                             "ticks": n_ticks, "max_abs_err_position": err, "bit_identical": same, "ok": bool(err <= 1e-5)}
        return out
    finally:
        world.close()


def measure_stacks_config(B, synth, torch, check, n=200_000):
    """Dynamic boxes against each other (bge_world_set_dynamic_contacts; after SURVEY 8(f) rank 4): n / 2 two-box stacks resting on
    the plane — n / 2 pairs, n / 2 simulation islands of two, every island through the pair cache, the narrowphase and one solver
    thread — timed awake and resting.  The checker replays the first 512 stacks on the oracle (island_ref.h)."""
    import numpy as np
    from banggameengine_amd.world import FIXED_DT, GRAVITY
    wl = synth.config("flat1m", n=n)
    side = int(np.ceil(np.sqrt(n / 2)))
    k = np.arange(n)
    wl.pos[:, 0] = ((k // 2) % side).astype(np.float32) * 4.0
    wl.pos[:, 2] = ((k // 2) // side).astype(np.float32) * 4.0
    wl.pos[:, 1] = (0.5 + 1.0 * (k % 2)).astype(np.float32)
    wl.euler[:] = 0.0
    wl.body_type[:] = 1
    stream = torch.cuda.current_stream()
    world = B.World(stream=stream.cuda_stream, pair_capacity=16 * n)
    try:
        world.load(wl)
        world.set_ground_plane(True)
        world.set_dynamic_contacts(True)
        world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=B.TICK_ALL, ticks=30)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=B.TICK_ALL, ticks=60)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 60 * 1e3
        hdr, _ = world.download_dynamic_pairs()
        out = {"label": "Dynamic boxes against each other: two-box stacks resting on the plane (pair cache, islands, one solver thread per island)",
               "workload": "two-box stacks + ground plane + dynamic contacts", "entities": n, "pairs": int(len(hdr)), "ms_per_step": ms,
               "value": n / (ms * 1e-3),
               "note": "wall clock per tick, two counter read-backs per sub-step included; sequential impulses inside an island: no roofline fraction is quoted"}
        if check:
            from oracle import pyoracle as po
            sample = 1024
            swl = synth.config("flat1m", n=sample)
            swl.pos[:] = wl.pos[:sample]
            swl.euler[:] = 0.0
            swl.body_type[:] = 1
            parent_i32 = np.where(swl.parent == 0xFFFFFFFF, -1, swl.parent.astype(np.int64)).astype(np.int32)
            ref = po.RefScene().bulk_build(parent_i32, swl.pos, swl.euler, swl.scale, body_type=swl.body_type)
            ref.SetPhysicsOptions(gravity_y=-9.81, orient_mode=po.ORIENT_IDEAL)
            ref.SetGroundPlane(True)
            ref.SetDynamicContacts(True)
            for _ in range(90):
                ref.PhysicsSystemUpdate(FIXED_DT)
                ref.TransformSystemUpdate()
            want_p, _ = ref.bulk_pose()
            want_v = ref.bulk_bodies()["linvel"]
            ref.close()
            got_p, _ = world.download_pose(0, sample)
            got_v = world.download_bodies(0, sample)["linvel"]
            same = bool(np.array_equal(got_p.view(np.uint32), want_p.view(np.uint32)) and np.array_equal(got_v.view(np.uint32), want_v.view(np.uint32)))
            err = float(np.max(np.abs(got_p.astype(np.float64) - want_p.astype(np.float64))))
            out["parity"] = {"checker": "oracle port (CPU) with its island restatement, same tick sequence", "sample_entities": sample, "ticks": 90,
                             "max_abs_err_position": err, "bit_identical": same, "ok": bool(err <= 1e-5)}
        return out
    finally:
        world.close()


# ----------------------------------------------------------------------------------------------------------- a rank
def run_rank(args):
    # stdout carries exactly ONE line, the result.  Libraries print there too (RCCL writes its version banner to fd 1
    # when the first communicator is created), so fd 1 is pointed at stderr for the duration of the run and the JSON line
    # goes to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    import torch.distributed as dist

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world_size:
        # a line that says n_gpus = 1 for a --gpus 8 request (or the reverse) is worse than no line
        print(f"[bench] rank {rank}: --gpus {args.gpus} but WORLD_SIZE = {world_size}; refusing to run", file=sys.stderr)
        return 2
    n_gpus = world_size
    deadlines = Deadlines(rank, args.deadline)
    if STUB:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        assert local_rank < torch.cuda.device_count(), f"LOCAL_RANK {local_rank} but {torch.cuda.device_count()} visible GPU(s)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    def dev_sync():
        if not STUB:
            torch.cuda.synchronize()

    if world_size > 1 or args.force_gather:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        with deadlines.phase("torch.distributed.init_process_group (rendezvous with the other ranks)"):
            if STUB:
                dist.init_process_group(backend="gloo", rank=rank, world_size=world_size)
            else:
                dist.init_process_group(backend="nccl", rank=rank, world_size=world_size, device_id=dev)

    import banggameengine_amd as B
    from banggameengine_amd import synth
    from banggameengine_amd.world import FIXED_DT, GRAVITY
    if STUB:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from bench_stub import StubWorld as WorldClass
    else:
        WorldClass = B.World

    name = args.workload or ("flat1m" if n_gpus == 1 else "subtree64")
    base = synth.config(name, n=64)
    per_gpu = args.entities or synth.DEFAULT_N[name]
    if base.shape == synth.SUBTREE64:
        per_gpu -= per_gpu % 64
    elif base.shape == synth.CHAINS4:
        per_gpu -= per_gpu % 4
    wl = synth.config(name, n=per_gpu, first=rank * per_gpu)  # global entity numbering: shard r = [r*per_gpu, (r+1)*per_gpu)
    basis_flag = B.TICK_BULLET_BASIS if args.bullet_basis else 0
    flags = B.TICK_ALL | (B.TICK_BROADPHASE if name == "cube4m" else 0) | basis_flag

    # a dedicated (non-null) torch stream carries the world's kernels, the events and the collective
    if STUB:
        world = WorldClass()
    else:
        stream = torch.cuda.Stream()
        torch.cuda.set_stream(stream)
        world = WorldClass(device=local_rank, stream=stream.cuda_stream)
    world.load(wl)
    ticks_after_vel = 0  # every tick after set_velocities, for the parity replay

    def tick(tick_flags, ticks=1):
        nonlocal ticks_after_vel
        world.tick(dt=FIXED_DT, gravity=GRAVITY, flags=tick_flags, ticks=ticks)
        ticks_after_vel += ticks

    world.tick(dt=FIXED_DT, flags=B.TICK_ALL | basis_flag)  # first tick creates the bodies (zero velocity), as in the reference
    world.set_velocities(wl.vel)            # synthetic initial velocities
    info = world.info()

    gather = (n_gpus > 1 and not args.no_gather) or args.force_gather
    n_roots = info["n_roots"]

    # The scaling curve's anchor, in the SAME line: before any communicator exists every rank ticks its own
    # shard alone — same workload, same K, same timing (barrier-free wall clock around a synchronised batch + one HIP event pair
    # around the K launches) — so that value / (N x that) can be read without a second run.
    solo = None
    if n_gpus > 1 or args.force_gather:
        with deadlines.phase("single-GPU anchor (this rank's shard, no collective)"):
            if args.warmup:
                tick(flags, args.warmup)
            dev_sync()
            regions, kernels = [], []
            for _ in range(max(1, args.repeats)):
                world.profile_enable(1)
                dev_sync()
                ta = time.perf_counter()
                tick(flags, args.steps)
                dev_sync()
                regions.append(time.perf_counter() - ta)
                k_ms, k_ticks = world.profile_read()
                kernels.append(k_ms / max(k_ticks, 1))
            world.profile_enable(0)
            regions.sort()
            kernels.sort()
            mine = torch.tensor([regions[len(regions) // 2], kernels[len(kernels) // 2]], dtype=torch.float64, device=dev)
            slowest = mine.clone()
            if dist.is_initialized():
                dist.all_reduce(slowest, op=dist.ReduceOp.MAX)
            solo = {"region_s_rank0": float(mine[0]), "kernel_ms_rank0": float(mine[1]),
                    "region_s_slowest_rank": float(slowest[0]), "kernel_ms_slowest_rank": float(slowest[1])}
    collective = "none"
    native = False
    roots = None
    rows_per_rank = 0
    if gather:
        with deadlines.phase("all_reduce(MAX) of the root count"):
            rows = torch.tensor([n_roots], dtype=torch.int64, device=dev)
            dist.all_reduce(rows, op=dist.ReduceOp.MAX)
            rows_per_rank = int(rows.item())
        native = args.collective == "rccl"
        if native:
            ok = 1
            with deadlines.phase("native communicator set-up (broadcast of the RCCL unique id, ncclCommInitRank)"):
                try:
                    uid = [WorldClass.comm_unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(uid, src=0)
                    world.comm_init(n_gpus, rank, uid[0], rows_per_rank)
                except Exception as e:  # transport choice only: the torch path moves the same bytes
                    print(f"[bench] rank {rank}: native RCCL init failed ({e})", file=sys.stderr)
                    ok = 0
                # the choice of transport must be the same on every rank, or the first collective deadlocks
                agreed = torch.tensor([ok], dtype=torch.int64, device=dev)
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
            with deadlines.phase("RootTable set-up (all_gather of root counts)"):
                roots = RootTable(n_roots, dev, overlap=not args.no_overlap)
            collective = (f"torch.distributed.all_gather_into_tensor ({'gloo, STUB' if STUB else 'RCCL'}) of {roots.rows} x 64 B root "
                          f"world matrices per rank per step, {'side stream, double-buffered' if roots.overlap else 'compute stream'}")

    def step():
        tick(flags)
        if gather:
            # frame t's roots: packed on the compute stream, gathered on the side stream under frame t+1's tick
            if native:
                world.gather_roots()
            else:
                world.pack_roots(roots.send_buffer().data_ptr())
                roots.gather()

    def barrier():
        if gather:
            world.comm_wait() if native else roots.finish()
        if dist.is_initialized():
            dist.barrier()
        dev_sync()

    with deadlines.phase("warm-up steps" + (" (first per-frame root gather)" if gather else "")):
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
            with deadlines.phase(f"gather schedule trial, mode {mode} ({'ncclAllGather' if mode == 0 else 'grouped ncclSend/ncclRecv per peer'})"):
                world.comm_set_mode(mode)
                tick(flags | B.TICK_GATHER_ROOTS, 8)
                barrier()
                t_a = time.perf_counter()
                tick(flags | B.TICK_GATHER_ROOTS, 40)
                barrier()
                el = torch.tensor([time.perf_counter() - t_a], dtype=torch.float64, device=dev)
                dist.all_reduce(el, op=dist.ReduceOp.MAX)
                gather_trials[mode] = float(el.item()) / 40 * 1e3
        best = min(gather_trials, key=gather_trials.get)
        with deadlines.phase("settling on the chosen gather schedule"):
            world.comm_set_mode(best)
            tick(flags | B.TICK_GATHER_ROOTS, 8)
            barrier()
        collective += (f"; schedule {'direct send/recv per peer' if best else 'ncclAllGather'} chosen by trial "
                       f"(ms per frame: all-gather {gather_trials[0]:.4f}, direct {gather_trials[1]:.4f})")
    elif gather and native and args.gather_mode:
        world.comm_set_mode({"allgather": 0, "direct": 1}[args.gather_mode])
        collective += f"; schedule forced: {args.gather_mode}"
    # Timed region: exactly K steps, no per-step host work.  Without a collective the K tick launches are
    # bracketed by ONE HIP event pair recorded by the library on the launch stream (bge_world_profile_enable(1)),
    # so the roofline figure is the average launch duration inside the timed region, gaps included.
    # Run R times back to back; every region is exactly K steps between barrier + synchronize on both sides, its time the MAX over
    # ranks; the median region is the line's ms_per_step / value (R and every region's time are in the line).
    repeats = max(1, args.repeats)
    region_s, region_kernel_ms = [], []
    for rep in range(repeats):
        world.profile_enable(0 if gather else 1)
        with deadlines.phase(f"timed region {rep + 1} of {repeats} ({args.steps} steps" + (", one root gather per step)" if gather else ")")):
            barrier()
            t0 = time.perf_counter()
            if not gather:
                tick(flags, args.steps)
            elif native:
                # one native call enqueues K frames: tick kernels, root packing, ncclAllGather on the side stream
                tick(flags | B.TICK_GATHER_ROOTS, args.steps)
            else:
                for _ in range(args.steps):
                    step()
            barrier()
            t1 = time.perf_counter()
        el = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
        if dist.is_initialized():
            with deadlines.phase("all_reduce(MAX) of the elapsed time"):
                dist.all_reduce(el, op=dist.ReduceOp.MAX)
        region_s.append(float(el.item()))
        if not gather:
            k_ms, k_ticks = world.profile_read()
            assert k_ticks == args.steps, (k_ticks, args.steps)
            region_kernel_ms.append(k_ms / k_ticks)
    kernel_note = "one HIP event pair around the K launches of the timed region"
    if gather:
        # the collective and the packing kernel share the stream with the tick kernels: time the tick kernels in a
        # second, instrumented pass of the same K steps (one event pair per tick)
        world.profile_enable(2)
        with deadlines.phase("instrumented second pass (event pair per tick, collective interleaved)"):
            if native:
                tick(flags | B.TICK_GATHER_ROOTS, args.steps)
            else:
                for _ in range(args.steps):
                    step()
            barrier()
        kernel_note = "event pair per tick in a second, instrumented pass of the same K steps (collective interleaved)"
    if gather:
        kernel_total_ms, kernel_ticks = world.profile_read()
        assert kernel_ticks == args.steps, (kernel_ticks, args.steps)
        kernel_ms = kernel_total_ms / kernel_ticks
    else:
        kernel_ms = sorted(region_kernel_ms)[len(region_kernel_ms) // 2]
    world.profile_enable(0)

    gather_check = None
    if gather and native:
        # Correctness of the collective (outside the timed region): one more frame, then every rank compares EVERY
        # rank's segment of its gathered table with a checksum that rank computed from its own root matrices.
        with deadlines.phase("gather check (one more gathered frame + all_gather of checksums)"):
            tick(flags | B.TICK_GATHER_ROOTS)
            table_host = world.download_gathered(n_gpus, rows_per_rank)
            slot_world = world.download_world()
            root_ids = np.flatnonzero(wl.parent == 0xFFFFFFFF)
            mine = np.zeros((rows_per_rank, 16), np.float32)
            mine[: len(root_ids)] = slot_world[root_ids]
            own_ok = np.array_equal(table_host[rank].view(np.uint32), mine.view(np.uint32))
            sums = torch.tensor([int(mine.view(np.uint32).astype(np.uint64).sum() & 0x7FFFFFFFFFFFFFFF)], dtype=torch.int64, device=dev)
            all_sums = [torch.zeros_like(sums) for _ in range(n_gpus)]
            dist.all_gather(all_sums, sums)
            seen = [int(table_host[r].view(np.uint32).astype(np.uint64).sum() & 0x7FFFFFFFFFFFFFFF) for r in range(n_gpus)]
            ok = own_ok and all(int(all_sums[r].item()) == seen[r] for r in range(n_gpus))
            flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        gather_check = "ok: every rank holds every rank's root matrices bit for bit" if int(flag.item()) == 1 else "MISMATCH"
        if gather_check == "MISMATCH":
            print(f"[bench] rank {rank}: gathered root table does not match (own segment ok: {own_ok})", file=sys.stderr)

    elapsed_s = sorted(region_s)[len(region_s) // 2]   # the median region (max over ranks each)
    total_entities = per_gpu * n_gpus
    value = total_entities * args.steps / elapsed_s

    rc = 0
    if rank == 0:
        bytes_per_update = wl.bytes_per_update
        kernel_name = "k_tick<physics,transforms>"
        if name == "cube4m":
            # 208 B + 8 B per emitted pair; the event pair brackets the tick kernel AND the
            # broadphase kernels of every step, so both sides of the ratio cover the whole step
            bytes_per_update = 208.0 + 8.0 * world.pair_count() / per_gpu
            kernel_name = "k_tick<physics,transforms,aabb> + broadphase (sort, pair search)"
        if args.bullet_basis:
            bytes_per_update += BASIS_EXTRA_BYTES * float((wl.body_type == 1).mean())
            kernel_name += " [bullet basis]"
        alg_bytes = bytes_per_update * per_gpu  # per step on one GPU
        achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_source = load_traffic(name + "_basis" if args.bullet_basis else name, per_gpu)
        out = {
            "metric": "entity-updates/sec (transform+physics tick)",
            "value": 0.0 if STUB else value,
            "unit": "entity-updates/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_s / args.steps * 1e3,
            "repeats": repeats,
            "region_ms": [r * 1e3 for r in region_s],
            "timing": f"median of {repeats} back-to-back regions of exactly {args.steps} steps, each between barrier + synchronize, max over ranks",
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "STUB — control-flow test without a GPU, no measurement" if STUB else "synthetic",
            "config": {
                "workload": f"{name}: {per_gpu} entities/GPU x {n_gpus} GPU, " + CONFIG_TEXT[name],
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
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel_ms_per_launch": kernel_ms,
                "timing": kernel_note,
                "algorithmic_bytes_per_launch": alg_bytes,
            },
        }
        if solo is not None:
            solo_value = per_gpu * args.steps / solo["region_s_rank0"]
            out["single_gpu_same_workload"] = {
                "value": 0.0 if STUB else solo_value, "unit": "entity-updates/s",
                "what": f"rank 0's own shard ({per_gpu} entities, {name}) ticked alone for {args.steps} steps before the communicator existed: "
                        f"median of {repeats} regions, same event-pair / wall-clock timing as the line's",
                "ms_per_step": solo["region_s_rank0"] / args.steps * 1e3, "kernel_ms_per_launch": solo["kernel_ms_rank0"],
                "ms_per_step_slowest_rank": solo["region_s_slowest_rank"] / args.steps * 1e3,
                "kernel_ms_per_launch_slowest_rank": solo["kernel_ms_slowest_rank"],
            }
            out["scaling_efficiency"] = None if STUB else value / (n_gpus * solo_value)
            out["compute_only_ms_per_step"] = solo["region_s_slowest_rank"] / args.steps * 1e3
            out["compute_plus_collective_ms_per_step"] = elapsed_s / args.steps * 1e3
        if n_gpus == 1 and not args.no_cpu and not STUB:
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds)
            out["cpu_allcore"] = cpu_allcore(wl, args.cpu_seconds / 2, args.cpu_threads)
    if not args.no_cpu and not STUB and base.shape in (synth.FLAT, synth.CHAINS4, synth.SUBTREE64):
        # every rank checks a sample of ITS shard against the CPU port (outside the timed region); rank 0 reports the worst
        with deadlines.phase("sampled parity check against the CPU port"):
            try:
                par = parity_sample(world, name, rank * per_gpu, ticks_after_vel, args.bullet_basis)
            except Exception as e:  # the checker could not run (not: it ran and disagreed)
                par = {"error": repr(e), "max_rel_err_world": float("nan"), "max_rel_err_position": float("nan"),
                       "bit_identical": False, "ok": None}
            if dist.is_initialized():
                # EVERY rank takes part: a rank whose checker could not run contributes inf and a flag — skipping the
                # collective on that rank alone would leave the others waiting for it until the deadline
                failed = par["ok"] is None
                worst = torch.tensor([float("inf") if failed else par["max_rel_err_world"], float("inf") if failed else par["max_rel_err_position"],
                                      0.0 if par["bit_identical"] else 1.0, 1.0 if failed else 0.0], dtype=torch.float64, device=dev)
                dist.all_reduce(worst, op=dist.ReduceOp.MAX)
                if worst[3].item() != 0.0:
                    par.update(ok=None, ranks_checked=n_gpus)
                    par.setdefault("error", "the checker could not run on another rank")
                else:
                    par.update(max_rel_err_world=float(worst[0]), max_rel_err_position=float(worst[1]),
                               bit_identical=bool(worst[2].item() == 0.0), ranks_checked=n_gpus)
                    par["ok"] = par["max_rel_err_world"] <= 1e-5 and par["max_rel_err_position"] <= 1e-5
        if rank == 0:
            out["parity"] = par
            if par["ok"] is False:
                print("[bench] PARITY FAILURE: the device state is outside 1e-5 of the CPU port", file=sys.stderr)
                rc = 4

    if gather and native:
        world.comm_destroy()
    world.close()

    if rank == 0 and n_gpus == 1 and not args.no_configs and not STUB and name == "flat1m" and not args.entities:
        # The other single-GPU configurations, same process, after the headline's timed region (every
        # single-GPU config in the one driver-timed line).  flat 16 M is the figure without Infinity-Cache help.
        extras = []
        check = not args.no_cpu
        for label, cname, ents, steps, warm, basis in (
                ("configs[2] 1M entities, depth-4 chains (LDS-staged hierarchy)", "chains4", None, 400, 20, False),
                ("configs[4] ONE GPU's shard: 2M entities in 64-node subtrees (variant 5b), no gather — the N = 1 anchor of the "
                 "scaling curve that `--gpus N` measures on this workload", "subtree64", None, 400, 20, False),
                ("configs[3] 4M entities + AABB broadphase", "cube4m", None, 20, 3, False),
                ("flat 16M (2.2 GB working set: beyond the 256 MiB Infinity Cache)", "flat1m", 16_000_000, 40, 5, False),
                ("configs[1] in Bullet's own orientation scheme (BGE_TICK_BULLET_BASIS)", "flat1m", None, 400, 20, True)):
            try:
                extras.append(measure_extra_config(B, synth, torch, label, cname, ents, steps, warm, basis, check))
            except Exception as e:  # the headline stands on its own
                extras.append({"label": label, "workload": cname, "error": repr(e)})
                rc = rc or 5
        try:   # (one more row of SURVEY 8(f))
            extras.append(measure_ground_config(B, synth, torch, None, check))
            if extras[-1].get("parity", {}).get("ok") is False:
                print("[bench] PARITY FAILURE on the ground plane: the device state is outside 1e-5 of the CPU port", file=sys.stderr)
                rc = rc or 4
        except Exception as e:
            extras.append({"label": "configs[1] on the reference's ground plane", "workload": "flat1m + ground plane", "error": repr(e)})
            rc = rc or 5
        try:   # (beyond SURVEY 8(f): Dynamic boxes against each other)
            extras.append(measure_stacks_config(B, synth, torch, check))
            if extras[-1].get("parity", {}).get("ok") is False:
                print("[bench] PARITY FAILURE on the two-box stacks: the device state is outside 1e-5 of the CPU port", file=sys.stderr)
                rc = rc or 4
        except Exception as e:
            extras.append({"label": "Dynamic boxes against each other: two-box stacks", "workload": "two-box stacks + ground plane + dynamic contacts", "error": repr(e)})
            rc = rc or 5
        out["configs"] = extras

    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        with deadlines.phase("final barrier"):
            dist.barrier()
        dist.destroy_process_group()
    return rc


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)  # before anything imports torch or loads libbge_world
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
