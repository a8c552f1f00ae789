"""Multi-GPU layer: shard a scene by whole scene-graph subtrees, one process per GPU, and keep a node-wide
table of root world matrices with ONE all-gather per frame (BASELINE.json configs[4]).

A subtree's world matrices depend only on its own root chain (src/ecs/TransformSystem.cpp:35 hands a child
nothing but its parent's world), and free rigid bodies integrate independently, so shards never exchange
anything on the data path; the only collective is the gather of the roots' results.  torch.distributed is the
transport ("nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).
"""
from __future__ import annotations

import numpy as np

from .world import NO_PARENT, partition_subtrees


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
