"""Synthetic FCGF-like pairs: there are no datasets or FCGF weights here.

A pair is two clouds with a planted rigid motion and partial overlap ``rho``; descriptors are
32-d unit vectors (the reference L2-normalises FCGF output, Experiments/misc/fcgf.py:387-391)
whose partner rows are noisy copies (noise scale ``s``), so that feature-space NN recovers a
controllable fraction of true correspondences.
"""
import numpy as np


def _unit(x):
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)


def _partner_perm(n0, n1, rho, rng):
    """Which rows of cloud 1 are partners of which rows of cloud 0 (K = floor(rho*min(n0,n1)))."""
    k = int(rho * min(n0, n1))
    src_rows = rng.permutation(n0)[:k]
    order1 = rng.permutation(n1)
    return k, src_rows, order1


def make_features(n0, n1, d=32, rho=0.5, s=1.2, seed=51):
    rng = np.random.default_rng(seed)
    k, src_rows, order1 = _partner_perm(n0, n1, rho, rng)
    F0 = _unit(rng.standard_normal((n0, d)))
    F1 = np.empty((n1, d), np.float32)
    F1[order1[:k]] = _unit(F0[src_rows] + s * rng.standard_normal((k, d)) / np.sqrt(d))
    F1[order1[k:]] = _unit(rng.standard_normal((n1 - k, d)))
    return F0, F1


def random_motion(rng, max_t_xy=30.0):
    yaw = rng.uniform(-np.pi, np.pi)
    roll, pitch = np.radians(rng.uniform(-2, 2, 2))
    cz, sz = np.cos(yaw), np.sin(yaw)
    cy, sy = np.cos(pitch), np.sin(pitch)
    cx, sx = np.cos(roll), np.sin(roll)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = [rng.uniform(-max_t_xy, max_t_xy), rng.uniform(-max_t_xy, max_t_xy), rng.uniform(-0.5, 0.5)]
    return T


def make_clouds(n0, n1, rho=0.5, seed=51, clustered=False, noise=0.05):
    """xyz0 [n0,3] f32, xyz1 [n1,3] f32, T_gt 4x4 f64 with xyz1[partner] ~ R xyz0 + t.

    Uses the same partner permutation as ``make_features`` for the same (n0, n1, rho, seed).
    ``clustered`` concentrates x,y around a few centres so that GPF grid cells are unevenly filled.
    """
    rng = np.random.default_rng(seed)
    k, src_rows, order1 = _partner_perm(n0, n1, rho, rng)
    rng = np.random.default_rng(seed + 1000003)

    def box(n):
        if clustered:
            c = rng.uniform(-60, 60, (6, 2))
            xy = c[rng.integers(0, 6, n)] + rng.normal(0, 12.0, (n, 2))
            xy = np.clip(xy, -80, 80)
        else:
            xy = rng.uniform(-80, 80, (n, 2))
        return np.concatenate([xy, rng.uniform(-3, 5, (n, 1))], axis=1)

    xyz0 = box(n0)
    T = random_motion(rng)
    xyz1 = np.empty((n1, 3))
    xyz1[order1[:k]] = xyz0[src_rows] @ T[:3, :3].T + T[:3, 3] + rng.normal(0, noise, (k, 3))
    xyz1[order1[k:]] = box(n1 - k)
    return xyz0.astype(np.float32), xyz1.astype(np.float32), T


def make_pair(N=30000, D=32, rho=0.5, s=1.2, seed=51, N1=None, clustered=False):
    """dict(xyz0, xyz1, feats0, feats1, T_gt) -- BASELINE config #2 uses the defaults."""
    N1 = N if N1 is None else N1
    F0, F1 = make_features(N, N1, D, rho, s, seed)
    xyz0, xyz1, T = make_clouds(N, N1, rho, seed, clustered=clustered)
    return dict(xyz0=xyz0, xyz1=xyz1, feats0=F0, feats1=F1, T_gt=T)


def make_pair_dev(N=30000, D=32, rho=0.5, s=1.2, seed=51, device=None, noise=0.05, T_gt=None):
    """The recipe of ``make_pair`` drawn with torch's generator on `device` (milliseconds instead of a quarter second per
    30k-point pair, so a benchmark can hold hundreds of distinct pairs): same distributions, different numbers.
    T_gt: plant this 4x4 motion instead of a random one (the list-driven surrogate: a row's ground truth and overlap).
    Returns dict(xyz0, xyz1, feats0, feats1) of float32 device tensors and T_gt (4x4 float64 numpy)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    k = int(rho * N)

    def unit(x):
        return x / x.norm(dim=1, keepdim=True)

    def box(n):
        xy = torch.rand((n, 2), generator=g, device=device) * 160 - 80
        z = torch.rand((n, 1), generator=g, device=device) * 8 - 3
        return torch.cat([xy, z], dim=1)

    src_rows = torch.randperm(N, generator=g, device=device)[:k]
    order1 = torch.randperm(N, generator=g, device=device)
    F0 = unit(torch.randn((N, D), generator=g, device=device))
    F1 = torch.empty((N, D), device=device)
    F1[order1[:k]] = unit(F0[src_rows] + s * torch.randn((k, D), generator=g, device=device) / np.sqrt(D))
    F1[order1[k:]] = unit(torch.randn((N - k, D), generator=g, device=device))
    T = random_motion(np.random.default_rng(seed + 1000003)) if T_gt is None else np.asarray(T_gt, np.float64).reshape(4, 4)
    Tt = torch.from_numpy(T).to(device=device, dtype=torch.float32)
    xyz0 = box(N)
    xyz1 = torch.empty((N, 3), device=device)
    xyz1[order1[:k]] = xyz0[src_rows] @ Tt[:3, :3].T + Tt[:3, 3] + noise * torch.randn((k, 3), generator=g, device=device)
    xyz1[order1[k:]] = box(N - k)
    return dict(xyz0=xyz0.contiguous(), xyz1=xyz1.contiguous(), feats0=F0.contiguous(), feats1=F1.contiguous(), T_gt=T)
