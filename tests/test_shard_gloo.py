"""world_size-2 gloo run of the pair sharding + single gather (the N>1 path of bench.py / the CLI)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lidarregistration_amd import shard


def test_shard_indices_match_distributed_sampler():
    from torch.utils.data.distributed import DistributedSampler
    for P, W in [(7008, 8), (2592, 8), (555, 8), (10, 4), (3, 2), (1, 2), (5, 1)]:
        for r in range(W):
            s = DistributedSampler(list(range(P)), num_replicas=W, rank=r, shuffle=False)
            assert list(iter(s)) == shard.shard_indices(P, W, r)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, P, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    idx = shard.shard_indices(P, world, rank)
    rows = np.zeros((len(idx), shard.ROW))
    for k, i in enumerate(idx):            # a fake "registration": row content is a function of the pair id only
        rows[k, :] = i + np.arange(shard.ROW) / 100.0
    table = shard.gather_rows(rows, P)
    if rank == 0:
        q.put(table)
    dist.barrier()
    dist.destroy_process_group()


def test_gather_rows_world2_gloo():
    P, world = 11, 2                      # odd count -> wrap-around padding on rank 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, P, q)) for r in range(world)]
    for p in procs: p.start()
    table = q.get(timeout=120)
    for p in procs: p.join(timeout=120)
    assert all(p.exitcode == 0 for p in procs)
    expect = np.arange(P)[:, None] + np.arange(shard.ROW)[None, :] / 100.0
    assert table.shape == (P, shard.ROW) and np.allclose(table, expect)


def test_gather_rows_single_rank():
    rows = np.random.default_rng(0).normal(size=(5, shard.ROW))
    assert np.array_equal(shard.gather_rows(rows, 5, world_size=1, rank=0), rows)
