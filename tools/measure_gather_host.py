"""Where does a frame with BGE_TICK_GATHER_ROOTS spend its time on ONE rank: host enqueue or GPU?
Run on a GPU box: python tools/measure_gather_host.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import banggameengine_amd as B  # noqa: E402
from banggameengine_amd import synth  # noqa: E402

wl = synth.config("subtree64", n=2_000_000 - 2_000_000 % 64)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
w = B.World(device=0, stream=stream.cuda_stream)
w.load(wl)
w.tick(flags=B.TICK_ALL)
w.set_velocities(wl.vel)
n_roots = w.info()["n_roots"]
uid = B.World.comm_unique_id()
w.comm_init(1, 0, uid, n_roots)
for mode in (0, 1):
    w.comm_set_mode(mode)
    for flags, name in ((B.TICK_ALL, "tick only"), (B.TICK_ALL | B.TICK_GATHER_ROOTS, "tick + gather")):
        w.tick(flags=flags, ticks=50)
        w.sync()
        t0 = time.perf_counter()
        w.tick(flags=flags, ticks=400)
        t1 = time.perf_counter()
        w.sync()
        t2 = time.perf_counter()
        print(f"mode {mode} {name}: host enqueue {1e6 * (t1 - t0) / 400:.1f} us/frame, until done {1e6 * (t2 - t0) / 400:.1f} us/frame", flush=True)
w.comm_destroy()
