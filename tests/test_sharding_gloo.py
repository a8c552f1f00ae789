"""N > 1 path on CPU: world_size-2 gloo processes.  Each rank owns whole subtrees, ticks its shard (the oracle
stands in for the GPU here — this test checks the sharding and the per-frame root gather, not the kernels),
packs its roots and all-gathers them; the gathered table must equal the unsharded result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NONE = 0xFFFFFFFF


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _scene():
    from banggameengine_amd import synth
    rng = np.random.default_rng(42)
    n = 3000
    parent = np.full(n, NONE, np.uint32)
    for i in range(1, n):
        if rng.random() < 0.9:
            parent[i] = rng.integers(max(0, i - 25), i)
    pos, euler, scale = synth.trs(0xABCD, 0, n)
    vel = synth.velocity(0xABCD, 0, n)
    body = np.where(parent == NONE, 1, 255).astype(np.uint8)
    return parent, pos, euler, scale, vel, body


def _tick_shard(parent, pos, euler, scale, vel, body, ticks):
    from helpers import DT, parent_i32
    from oracle import pyoracle as po
    sc = po.RefScene().bulk_build(parent_i32(parent), pos, euler, scale, body_type=body)
    for k in range(ticks):
        sc.PhysicsSystemUpdate(DT)
        sc.TransformSystemUpdate()
        if k == 0:
            sc.bulk_set_velocity(vel)
    world, _ = sc.bulk_world()
    return world


def _worker(rank, world_size, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from banggameengine_amd import sharding
        parent, pos, euler, scale, vel, body = _scene()
        rank_of, load, shards = sharding.shard_scene(parent, world_size)
        ids, local_parent = shards[rank]
        world = _tick_shard(local_parent, pos[ids], euler[ids], scale[ids], vel[ids], body[ids], ticks=3)
        roots = np.flatnonzero(local_parent == NONE)
        table = sharding.RootTable(len(roots), "cpu", overlap=False)
        for frame in range(2):   # two frames: the buffers are reused
            table.send_buffer()[: len(roots)] = torch.from_numpy(world[roots])
            gathered = table.gather()
        if rank == 0:
            rows, gids = [], []
            for r in range(world_size):
                r_ids, r_parent = shards[r]
                rows.append(table.rows_of(gathered, r).numpy())
                gids.append(r_ids[np.flatnonzero(r_parent == NONE)])
            np.savez(out_path, table=np.concatenate(rows), gids=np.concatenate(gids), load=load)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_root_table_matches_unsharded(tmp_path):
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    z = np.load(out)
    parent, pos, euler, scale, vel, body = _scene()
    full = _tick_shard(parent, pos, euler, scale, vel, body, ticks=3)
    roots = np.flatnonzero(parent == NONE)
    order = np.argsort(z["gids"])
    assert np.array_equal(z["gids"][order], roots)                       # every root exactly once
    assert np.array_equal(z["table"][order].view(np.uint32), full[roots].view(np.uint32))
    assert abs(int(z["load"][0]) - int(z["load"][1])) <= 64              # balanced node counts


def test_extract_shard_rejects_split_subtrees():
    from banggameengine_amd import sharding
    parent = np.array([NONE, 0, 1, NONE], np.uint32)
    with pytest.raises(ValueError):
        sharding.extract_shard(parent, np.array([0, 0, 1, 1]), 1)
    ids, lp = sharding.extract_shard(parent, np.array([0, 0, 0, 1]), 0)
    assert ids.tolist() == [0, 1, 2] and lp.tolist() == [NONE, 0, 1]
