"""Pair sharding across the GPUs of one node and the single result gather.

The reference shards a ``balanced_sets`` pair list round-robin over one OS process per GPU
(``DistributedSampler(num_replicas, rank, shuffle=False)``, Experiments/dataloader/data_loaders.py:111-126;
Experiments/test_parallel.sh:18-20) and merges per-rank ``.npy`` files on disk (Experiments/test.py:257,57-61).
Here: same pair -> rank mapping, one ``torch.distributed`` process per GPU, and ONE collective at the end
(RCCL ``all_gather`` on GPUs, gloo in the CPU tests).  Payload is tiny (rows of 38 float64), so the gather is
latency-bound; no collective sits on the data path.
"""
import math

import numpy as np
import torch
import torch.distributed as dist

ROW = 22 + 16      # stats row of Experiments/test.py:98-100 + the 4x4 transform


def shard_indices(num_pairs, world_size, rank):
    """Indices of this rank: rank, rank+W, rank+2W, ... padded by wrap-around to ceil(P/W) entries
    (what DistributedSampler(shuffle=False, drop_last=False) yields)."""
    per_rank = int(math.ceil(num_pairs / world_size))
    total = per_rank * world_size
    idx = list(range(num_pairs))
    idx += idx[: total - num_pairs] if num_pairs > 0 else []
    while len(idx) < total:                      # world_size > 2*num_pairs
        idx += idx[: total - len(idx)]
    return idx[rank:total:world_size]


def gather_rows(local_rows, num_pairs, world_size=None, rank=None, device=None):
    """All ranks contribute [ceil(P/W), ROW] float64 rows (in shard order); every rank gets back the
    de-interleaved [P, ROW] table in list order, wrap-around padding dropped.  One collective."""
    world_size = dist.get_world_size() if world_size is None else world_size
    rank = dist.get_rank() if rank is None else rank
    local = torch.as_tensor(np.asarray(local_rows, np.float64))
    per_rank = int(math.ceil(num_pairs / world_size))
    assert local.shape == (per_rank, local.shape[1]), (local.shape, per_rank)
    if device is not None:
        local = local.to(device)
    if world_size == 1:
        return local.cpu().numpy()[:num_pairs]
    out = torch.empty((world_size, per_rank, local.shape[1]), dtype=torch.float64, device=local.device)
    dist.all_gather_into_tensor(out.view(-1, local.shape[1]), local.contiguous())
    # out[r, k] is list index k*W + r
    table = out.permute(1, 0, 2).reshape(per_rank * world_size, local.shape[1])
    return table.cpu().numpy()[:num_pairs]
