"""Multi-GPU layer: shard a scene by whole scene-graph subtrees, one process per GPU, and keep a node-wide
table of root world matrices with ONE all-gather per frame (BASELINE.json configs[4]).

A subtree's world matrices depend only on its own root chain (src/ecs/TransformSystem.cpp:35 hands a child
nothing but its parent's world), and free rigid bodies integrate independently, so shards never exchange
anything on the data path; the only collective is the gather of the roots' results.  torch.distributed is the
transport ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
from __future__ import annotations

import numpy as np

from .world import NO_PARENT, balanced_cuts, partition_subtrees


def extract_shard(parent, rank_of_entity, rank):
    """Entities of one rank as a self-contained scene.

    Returns (global_ids, local_parent): global_ids are the shard's entities in ascending global order;
    local_parent[k] is the parent of global_ids[k] in shard-local numbering (NO_PARENT for roots).  Whole
    subtrees stay together, so every parent is inside the shard.
    """
    parent = np.asarray(parent, np.uint32)
    rank_of_entity = np.asarray(rank_of_entity)
    ids = np.flatnonzero(rank_of_entity == rank).astype(np.uint32)
    local_of_global = np.full(len(parent), NO_PARENT, np.uint32)
    local_of_global[ids] = np.arange(len(ids), dtype=np.uint32)
    p = parent[ids]
    has = p != NO_PARENT
    local_parent = np.full(len(ids), NO_PARENT, np.uint32)
    local_parent[has] = local_of_global[p[has]]
    if (local_parent[has] == NO_PARENT).any():
        raise ValueError("partition splits a subtree: a parent lives on another rank")
    return ids, local_parent


def shard_scene(parent, nranks, has_transform=None):
    """rank_of_entity + per-rank (global_ids, local_parent)."""
    rank_of_entity, load = partition_subtrees(parent, nranks, has_transform)
    return rank_of_entity, load, [extract_shard(parent, rank_of_entity, r) for r in range(nranks)]


class RootTable:
    """Per-frame all-gather of root world matrices.

    Every rank contributes `n_roots` rows of 16 floats (ranks are padded to the largest count); after
    `gather()` every rank holds the table of all ranks' roots, rank-major.  With `overlap=True` the collective
    is issued on a side stream and double-buffered, so frame t's gather runs under frame t+1's tick.
    """

    def __init__(self, n_roots, device, group=None, overlap=True):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device)
        counts = torch.tensor([n_roots], dtype=torch.int64, device=self.device)
        all_counts = [torch.zeros_like(counts) for _ in range(self.world_size)]
        dist.all_gather(all_counts, counts, group=group)
        self.counts = [int(c.item()) for c in all_counts]
        self.rows = max(max(self.counts), 1)
        self.n_roots = n_roots
        self.cuda = self.device.type == "cuda"
        self.overlap = overlap and self.cuda
        nbuf = 2 if self.overlap else 1
        self.mine = [torch.zeros((self.rows, 16), dtype=torch.float32, device=self.device) for _ in range(nbuf)]
        self.table = [torch.zeros((self.world_size * self.rows, 16), dtype=torch.float32, device=self.device)
                      for _ in range(nbuf)]
        self.frame = 0
        if self.overlap:
            self.comm_stream = torch.cuda.Stream(device=self.device)
            self.done = [None, None]

    def send_buffer(self):
        """The buffer this frame's roots must be packed into (device pointer via .data_ptr())."""
        b = self.frame % len(self.mine)
        if self.overlap and self.done[b] is not None:
            # the gather that last read this buffer must have finished before it is overwritten
            self.torch.cuda.current_stream().wait_event(self.done[b])
        return self.mine[b]

    def gather(self):
        """Issue the frame's all-gather; returns the table it fills."""
        torch, dist = self.torch, self.dist
        b = self.frame % len(self.mine)
        self.frame += 1
        if not self.overlap:
            dist.all_gather_into_tensor(self.table[b], self.mine[b], group=self.group)
            return self.table[b]
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ready)
            dist.all_gather_into_tensor(self.table[b], self.mine[b], group=self.group)
            ev = torch.cuda.Event()
            ev.record(self.comm_stream)
        self.done[b] = ev
        return self.table[b]

    def finish(self):
        """Make the compute stream wait for every outstanding gather (end of the timed region)."""
        if self.overlap:
            for ev in self.done:
                if ev is not None:
                    self.torch.cuda.current_stream().wait_event(ev)

    def rows_of(self, table, rank):
        """The valid rows rank `rank` contributed to a gathered table."""
        return table[rank * self.rows: rank * self.rows + self.counts[rank]]


# ---------------------------------------------------------------------------------------------------------------------
# Sharded broadphase: the global pair set of bodies that live on several ranks (include/bge_world.h, csrc/bge_route.hip).
#
# Subtree sharding interleaves the ranks' bodies in space, so for the pair search the bodies are re-partitioned into
# slabs along one axis, one slab per rank: every body sends a 48-byte record to each slab its AABB extent touches (one
# all-to-all), every rank searches what it received and keeps a pair only where the lower end of the pair's overlap
# interval, max(min_a, min_b), lies in its own slab — present on that rank by construction, and in exactly one slab.
RECORD_FLOATS = 12  # 48-byte record


def uniform_cuts(lo, hi, nranks):
    """nranks + 1 slab boundaries over [lo, hi]; the outer two are placeholders (-inf / +inf in effect).
    Evaluated identically on every rank from the all-reduced extent."""
    lo, hi = float(lo), float(hi)
    if not hi >= lo:
        return np.zeros(nranks + 1, np.float32)
    return np.array([lo + (hi - lo) * (k / nranks) for k in range(nranks + 1)], np.float64).astype(np.float32)


def slab_of(cuts, z):
    """Slab of a coordinate: the number of interior cuts <= z (what the device computes)."""
    cuts = np.asarray(cuts, np.float32)
    z = np.asarray(z, np.float32)
    return (z[..., None] >= cuts[1:-1]).sum(axis=-1).astype(np.int64) if len(cuts) > 2 else np.zeros(z.shape, np.int64)


def slab_window(cuts, rank):
    n = len(cuts) - 1
    return (-np.inf if rank == 0 else float(cuts[rank])), (np.inf if rank == n - 1 else float(cuts[rank + 1]))


def route_numpy(aabb, cuts, axis):
    """Host model of bge_world_bp_route/pack: for each destination slab the indices of the bodies sent there."""
    aabb = np.asarray(aabb, np.float32).reshape(-1, 6)
    lo = slab_of(cuts, aabb[:, axis])
    hi = np.maximum(lo, slab_of(cuts, aabb[:, 3 + axis]))
    return [np.flatnonzero((lo <= d) & (d <= hi)) for d in range(len(cuts) - 1)]


def keep_in_window(aabb_a, aabb_b, axis, window):
    """The dedup rule: a pair belongs to the slab that contains max(min_a, min_b) along the axis."""
    m = np.maximum(np.asarray(aabb_a, np.float32)[:, axis], np.asarray(aabb_b, np.float32)[:, axis])
    return (m >= np.float32(window[0])) & (m < np.float32(window[1]))


def slab_broadphase_local(worlds, axis=2, cuts=None):
    """Several worlds (shards) on ONE GPU in one process: the exchange is a set of device-to-device copies.
    Returns the per-world record counts matrix; afterwards world.pairs() of each world holds its slab's share of the
    global pair set (global ids).  This is the single-GPU rehearsal of SlabBroadphase below."""
    import torch
    n = len(worlds)
    if cuts is None:
        b = [w.aabb_bounds() for w in worlds]
        lo = min(float(x[0][axis]) for x in b if x[2])
        hi = max(float(x[1][axis]) for x in b if x[2])
        hist = sum(w.axis_histogram(axis, lo, hi) for w in worlds)     # the all-reduce
        cuts = balanced_cuts(hist, lo, hi, n)
    counts = np.stack([w.bp_route(axis, cuts) for w in worlds]).astype(np.int64)  # [src][dst]
    sends = []
    for s, w in enumerate(worlds):
        t = torch.empty(max(int(counts[s].sum()), 1) * RECORD_FLOATS, dtype=torch.float32, device="cuda")
        w.bp_pack(t.data_ptr())
        w.sync()
        sends.append(t)
    for d, w in enumerate(worlds):
        parts = []
        for s in range(n):
            off = int(counts[s, :d].sum()) * RECORD_FLOATS
            parts.append(sends[s][off: off + int(counts[s, d]) * RECORD_FLOATS])
        recv = torch.cat(parts) if parts else torch.empty(0, device="cuda")
        torch.cuda.synchronize()
        n_recv = int(counts[:, d].sum())
        wlo, whi = slab_window(cuts, d)
        w.bp_find(recv.data_ptr() if n_recv else 0, n_recv, axis, wlo, whi)
        w.sync()
    return counts, cuts


class SlabBroadphase:
    """One rank's side of the sharded broadphase over torch.distributed ("nccl" = RCCL): call find() after a tick that
    updated the AABBs (TICK_AABBS); world.pairs() then returns this rank's share of the global pair set."""

    def __init__(self, world, axis=2, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.world, self.axis, self.group = torch, dist, world, axis, group
        self.nranks, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.device = torch.device("cuda", torch.cuda.current_device())

    def find(self):
        torch, dist, w, axis = self.torch, self.dist, self.world, self.axis
        mn, mx, nb = w.aabb_bounds()
        ext = torch.tensor([-float(mn[axis]) if nb else -np.inf, float(mx[axis]) if nb else -np.inf],
                           dtype=torch.float32, device=self.device)
        dist.all_reduce(ext, op=dist.ReduceOp.MAX, group=self.group)
        lo, hi = -float(ext[0].item()), float(ext[1].item())
        if hi >= lo:
            hist = torch.from_numpy(w.axis_histogram(axis, lo, hi).astype(np.int64)).to(self.device)
            dist.all_reduce(hist, group=self.group)
            cuts = balanced_cuts(hist.cpu().numpy().astype(np.uint64), lo, hi, self.nranks)
        else:
            cuts = uniform_cuts(lo, hi, self.nranks)
        counts = w.bp_route(axis, cuts).astype(np.int64)
        mine = torch.from_numpy(counts).to(self.device)
        table = torch.empty((self.nranks, self.nranks), dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(table, mine, group=self.group)
        table = table.cpu().numpy()
        recv_counts = table[:, self.rank]
        send = torch.empty(max(int(counts.sum()), 1) * RECORD_FLOATS, dtype=torch.float32, device=self.device)
        recv = torch.empty(max(int(recv_counts.sum()), 1) * RECORD_FLOATS, dtype=torch.float32, device=self.device)
        w.bp_pack(send.data_ptr())
        w.sync()
        dist.all_to_all_single(recv[: int(recv_counts.sum()) * RECORD_FLOATS], send[: int(counts.sum()) * RECORD_FLOATS],
                               output_split_sizes=[int(c) * RECORD_FLOATS for c in recv_counts],
                               input_split_sizes=[int(c) * RECORD_FLOATS for c in counts], group=self.group)
        torch.cuda.current_stream().synchronize()
        wlo, whi = slab_window(cuts, self.rank)
        w.bp_find(recv.data_ptr(), int(recv_counts.sum()), axis, wlo, whi)
        self.keep = (send, recv)  # alive until the search has run
        return table, cuts
