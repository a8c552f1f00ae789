"""Stand-in for banggameengine_amd.World used ONLY by tests/test_bench_launcher.py (BGE_BENCH_STUB=1).

It lets the CPU suite drive bench.py's multi-rank CONTROL FLOW — the launcher, the process group, the unique-id
broadcast, the agreed fallback from the native collective to torch.distributed, the schedule trial, the deadlines —
over gloo, where there is neither a GPU nor RCCL.  It computes nothing: every number bench.py prints in this mode
is labelled "STUB" and `value` is 0.  Never imported by the product or by a real bench run.

Environment knobs (tests only):
    BGE_BENCH_STUB_FAIL_INIT_RANK=r   comm_init raises on rank r  -> every rank must fall back to torch.distributed
    BGE_BENCH_STUB_HANG_RANK=r        rank r sleeps forever inside its first gathered tick -> deadlines must fire
"""
from __future__ import annotations

import os
import time

import numpy as np


class StubWorld:
    calls: list

    def __init__(self, device=-1, stream=None, pair_capacity=0):
        self.n = 0
        self.rank = int(os.environ.get("RANK", "0"))
        self.calls = []
        self._prof = 0
        self._ticks = 0
        self._comm = False
        self._wl = None

    # -- what bench.py touches
    def load(self, wl, with_bodies=True):
        self.n = wl.n
        self._wl = wl
        return self

    def tick(self, dt=0.0, gravity=(0, 0, 0), flags=3, ticks=1):
        if (flags & 8) and str(self.rank) == os.environ.get("BGE_BENCH_STUB_HANG_RANK"):
            time.sleep(3600)
        if (flags & 8) and not self._comm:
            raise RuntimeError("stub: BGE_TICK_GATHER_ROOTS without comm_init")
        self.calls.append(("tick", flags, ticks))
        if self._prof:
            self._ticks += ticks

    def set_velocities(self, linvel=None, angvel=None, first=0):
        pass

    def info(self):
        roots = int((self._wl.parent == 0xFFFFFFFF).sum())
        return dict(n_entities=self.n, n_tiles=(self.n + 255) // 256, n_passes=1, n_roots=roots)

    def profile_enable(self, mode=1):
        self._prof = mode
        self._ticks = 0

    def profile_read(self):
        t, self._ticks = self._ticks, 0
        return 1e-3 * max(t, 1), t

    def pair_count(self):
        return 0

    @staticmethod
    def comm_unique_id() -> bytes:
        return bytes(range(128))

    def comm_init(self, nranks, rank, unique_id, rows_per_rank):
        assert unique_id == bytes(range(128)), "the unique id did not survive the broadcast"
        if str(rank) == os.environ.get("BGE_BENCH_STUB_FAIL_INIT_RANK"):
            raise RuntimeError("stub: comm_init fails on this rank")
        self._comm = True
        self.calls.append(("comm_init", nranks, rank, rows_per_rank))

    def comm_set_mode(self, mode):
        self.calls.append(("mode", mode))

    def comm_wait(self):
        pass

    def comm_destroy(self):
        self._comm = False

    def gather_roots(self):
        return 0

    def pack_roots(self, ptr=None):
        pass

    def download_gathered(self, nranks, rows_per_rank):
        return np.zeros((nranks, rows_per_rank, 16), np.float32)

    def download_world(self, first=0, count=None, out=None):
        return np.zeros((self.n if count is None else count, 16), np.float32)

    def download_pose(self, first=0, count=None):
        c = self.n if count is None else count
        return np.zeros((c, 3), np.float32), np.zeros((c, 3), np.float32)

    def sync(self):
        pass

    def close(self):
        pass
