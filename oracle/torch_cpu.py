"""torch-CPU restatement of the reference's Python path for one pair -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg may import this module (like everything under ``oracle/``).

What ``bench.py`` times as the CPU baseline: the reference's own way of computing a
pair on host cores, with the same cost structure --

  * ``find_nn``: brute-force L2 NN through ``torch.einsum`` in chunks of ``nn_max_n = 250`` query rows, each chunk
    materialising a 250 x N1 distance matrix, ``clamp_min(1e-30).sqrt_()``, ``min(dim=1)``, winner set to ``inf`` and
    ``min`` again for the 2nd neighbour (Experiments/algorithms/matching.py:22-65);
  * ``nn_to_mutual``: the reverse NN of the unique forward targets + an intersection (matching.py:222-239, :67-87; the
    intersection is numpy's sorted-key form of oracle.py, the reference builds two sparse COO matrices);
  * RANSAC + LS refit: the third-party C++ loop of the reference (Open3D / pygcransac, neither installable here) is
    replaced by oracle.c's OpenMP restatement of it (``orc_ransac`` / ``orc_refit``, FR.py:99-139).

The NN result is checked against oracle.c in tests/test_oracle_golden.py (indices agree except where fp32 summation order
breaks a near tie, which the test bounds).
"""
import time

import numpy as np
import torch

from . import oracle as orc

NN_MAX_N = 250          # matching.py:23


def find_nn(F0, F1, return_2nd=False, chunk=NN_MAX_N):
    """matching.py:22-65 on CPU tensors: (corres_idx0, corres_idx1, idx1_2nd or None), int64."""
    F0 = torch.as_tensor(F0, dtype=torch.float32)
    F1 = torch.as_tensor(F1, dtype=torch.float32)
    n0 = F0.shape[0]
    first, second = [], []
    with torch.no_grad():
        for lo in range(0, n0, chunk):
            q = F0[lo:lo + chunk]
            # the reference recomputes the column norms for every chunk (matching.py:29); so does this
            d2 = (q ** 2).sum(dim=1).reshape(-1, 1) + (F1 ** 2).sum(dim=1).reshape(1, -1) - 2 * torch.einsum("ac,bc->ab", q, F1)
            d = d2.clamp_min(1e-30).sqrt_()
            j1 = d.min(dim=1).indices
            first.append(j1)
            if return_2nd:
                d[torch.arange(d.shape[0]), j1] = float("inf")
                second.append(d.min(dim=1).indices)
    idx1 = torch.cat(first).long()
    idx0 = torch.arange(n0).long()
    return idx0, idx1, (torch.cat(second).long() if return_2nd else None)


def nn_to_mutual(F0, F1, idx0, idx1, idx2=None):
    """matching.py:222-239: reverse NN of the unique forward targets, then the pairs present in both directions."""
    F0 = torch.as_tensor(F0, dtype=torch.float32); F1 = torch.as_tensor(F1, dtype=torch.float32)
    uniq1 = torch.unique(idx1)
    _, inv0, _ = find_nn(F1[uniq1], F0, False)
    i, j = orc.torch_intersect(F0.shape[0], F1.shape[0], idx0.numpy(), idx1.numpy(), inv0.numpy(), uniq1.numpy())
    return i, j, (None if idx2 is None else idx2.numpy()[i])


def register_pair(xyz0, xyz1, feats0, feats1, mode="MNN", iters=50000, sample_size=3, thr=0.6, seed=51, confidence=1.0):
    """One pair the way the reference's FR() computes it with --codebase open3D (FR.py:16-119), on host cores.
    Returns dict(T, n_corr, t_nn, t_filter, t_ransac) -- the three times add up to the pair's wall time."""
    t0 = time.perf_counter()
    idx0, idx1, idx2 = find_nn(feats0, feats1, return_2nd=True)
    t1 = time.perf_counter()
    if mode in ("MNN", "MMN"):
        f0, f1, _ = nn_to_mutual(feats0, feats1, idx0, idx1, idx2)
    elif mode == "no_filter":
        f0, f1 = idx0.numpy(), idx1.numpy()
    else:
        raise AssertionError("torch_cpu.register_pair: mode must be MNN or no_filter")
    t2 = time.perf_counter()
    src = np.ascontiguousarray(np.asarray(xyz0, np.float32)[f0]); tgt = np.ascontiguousarray(np.asarray(xyz1, np.float32)[f1])
    T, info = orc.ransac(src, tgt, iters, sample_size, True, thr, seed, confidence)
    if info["best_h"] >= 0:
        T, _ = orc.refit(xyz0, xyz1, idx1.numpy().astype(np.int32), T, thr)
    t3 = time.perf_counter()
    return dict(T=T, n_corr=len(f0), t_nn=t1 - t0, t_filter=t2 - t1, t_ransac=t3 - t2, idx1=idx1.numpy(), idx2=idx2.numpy())
