"""RANSAC / Kabsch / refit operators on top of liblidarreg.so.

``GC_RANSAC`` keeps the reference's call shape (Experiments/algorithms/GC_RANSAC.py:8-55);
``RANSAC_registration`` keeps FR.py:122-139's.  Both run the same HIP kernels (hypothesis
generation, lane-per-hypothesis scoring, Kabsch) -- see csrc/lr_ransac.hip.
"""
import ctypes
from time import time

import numpy as np
import torch

from . import _ext
from .matching import _device, _f32, _i32, _stream, workspace

DEFAULT_SEED = 51          # Experiments/test.py:357
PRECHECK = {"NONE": 0, "ELC": 1, "SPRT": 2}      # --fast_rejection (test.py:306) -> lr_ransac_params.use_elc


def ransac_params(iters, sample_size=3, use_elc=True, thr=0.6, seed=DEFAULT_SEED, confidence=1.0, batch=0, sampler=0, prosac_growth=0,
                  scoring=0, local_opt=0, lo_rounds=0, lo_trials=0, lo_max_calls=0, min_iters=0):
    """confidence < 1 enables the early exit between batches of `batch` hypothesis ids (0 -> 1024, 8192, 65536, ...: eightfold); sampler 1 = PROSAC
    (correspondences best quality first, growth parameter T_N = prosac_growth, 0 -> 100000), 2 = uniform with unique indices; scoring 1 = MSAC,
    2 = MSAC at GC-RANSAC's truncated threshold (3/2 thr)^2; local_opt 1 = GC-RANSAC's local optimisation + final iterated least squares,
    2 = the latter only; lo_rounds / lo_trials / lo_max_calls / min_iters: 0 = the defaults of gcransac_python.cpp:513-517 (lidarreg.h)."""
    return _ext.RansacParams(int(sample_size), int(use_elc), np.float32(float(thr) * float(thr)), int(iters), int(seed),
                             float(confidence), int(batch), int(sampler), int(prosac_growth), int(scoring), int(local_opt),
                             int(lo_rounds), int(lo_trials), int(lo_max_calls), int(min_iters))


def ransac_dev(src, tgt, iters, sample_size=3, use_elc=True, thr=0.6, seed=DEFAULT_SEED, confidence=1.0, batch=0, sampler=0,
               prosac_growth=0, scoring=0, local_opt=0, want_mask=False, lo_rounds=0, lo_trials=0, lo_max_calls=0, min_iters=0):
    """RANSAC over M correspondences src[i] <-> tgt[i] ([M,3]).  Returns (T 4x4 float64 numpy, info dict); with want_mask the
    info holds the inlier mask of the returned model (what pygcransac.findRigidTransform returns next to the pose)."""
    src, tgt = _f32(src), _f32(tgt)
    m = src.shape[0]
    ws = workspace(max(m, 1), 1, iters)
    T = torch.empty(16, dtype=torch.float64, device=src.device)
    res = torch.zeros(ctypes.sizeof(_ext.RansacResult), dtype=torch.uint8, device=src.device)
    p = ransac_params(iters, sample_size, use_elc, thr, seed, confidence, batch, sampler, prosac_growth, scoring, local_opt,
                      lo_rounds, lo_trials, lo_max_calls, min_iters)
    _ext.check(_ext.lib().lr_ransac(ws.handle, src.data_ptr(), tgt.data_ptr(), m, None, ctypes.byref(p),
                                     T.data_ptr(), res.data_ptr(), _stream()))
    mask = None
    if want_mask:
        # the estimator's own inlier set: same enqueue, no synchronisation in between (scoring 2 tests against the truncated threshold)
        mask = torch.zeros(max(m, 1), dtype=torch.uint8, device=src.device)
        nin = torch.zeros(1, dtype=torch.int32, device=src.device)
        _ext.check(_ext.lib().lr_inlier_mask(ws.handle, src.data_ptr(), tgt.data_ptr(), m, T.data_ptr(), ctypes.c_float(p.effective_thr2()),
                                              mask.data_ptr(), nin.data_ptr(), _stream()))
    r = _ext.RansacResult.from_buffer_copy(res.cpu().numpy().tobytes())          # (the one synchronisation of the call)
    info = dict(best_h=r.best_h, best_count=r.best_count, best_ssq=r.best_ssq, n_valid=r.n_valid, n_ids=r.n_ids)
    if want_mask:
        if r.best_h < 0:          # no model (pygcransac returns pose None there): the mask of nothing
            info["mask"] = np.zeros(m, bool); info["n_inliers"] = 0
        else:
            info["mask"] = mask[:m].cpu().numpy().astype(bool); info["n_inliers"] = int(nin.item())
    return T.cpu().numpy().reshape(4, 4), info


def refit_dev(xyz0, xyz1, idx1, T, thr=0.6):
    """FR.py:99-111: LS refit over the original NN pairs within thr of T.  Returns (T 4x4, n_inliers)."""
    xyz0, xyz1, idx1 = _f32(xyz0), _f32(xyz1), _i32(idx1)
    n0 = xyz0.shape[0]
    ws = workspace(n0, 1)
    Tin = torch.as_tensor(np.ascontiguousarray(T, np.float64).reshape(16)).to(xyz0.device)
    Tout = torch.empty(16, dtype=torch.float64, device=xyz0.device)
    n = torch.zeros(1, dtype=torch.int32, device=xyz0.device)
    _ext.check(_ext.lib().lr_refit(ws.handle, xyz0.data_ptr(), n0, xyz1.data_ptr(), idx1.data_ptr(), Tin.data_ptr(),
                                    float(thr) * float(thr), Tout.data_ptr(), n.data_ptr(), _stream()))
    return Tout.cpu().numpy().reshape(4, 4), int(n.item())


def icp_dev(xyz0, xyz1, T_init, max_dist=0.6, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
    """Point-to-point ICP refinement (Experiments/test.py:183-189).  Returns (T 4x4 float64 numpy, info dict)."""
    xyz0, xyz1 = _f32(xyz0), _f32(xyz1)
    ws = workspace(xyz0.shape[0], xyz1.shape[0])
    Tin = torch.as_tensor(np.ascontiguousarray(T_init, np.float64).reshape(16)).to(xyz0.device)
    Tout = torch.empty(16, dtype=torch.float64, device=xyz0.device)
    res = torch.zeros(ctypes.sizeof(_ext.IcpResult), dtype=torch.uint8, device=xyz0.device)
    _ext.check(_ext.lib().lr_icp(ws.handle, xyz0.data_ptr(), xyz0.shape[0], xyz1.data_ptr(), xyz1.shape[0], Tin.data_ptr(),
                                  float(max_dist), int(max_iter), float(rel_fitness), float(rel_rmse), Tout.data_ptr(), res.data_ptr(),
                                  _stream()))
    r = _ext.IcpResult.from_buffer_copy(res.cpu().numpy().tobytes())
    return Tout.cpu().numpy().reshape(4, 4), dict(fitness=r.fitness, inlier_rmse=r.inlier_rmse, n_corr=r.n_corr, iterations=r.iterations)


def kabsch_dev(P, Q, w=None):
    """Least-squares rigid fit Q ~ R P + t (models/common.py:7-45).  Returns 4x4 float64 numpy."""
    dev = _device()
    P = torch.as_tensor(np.ascontiguousarray(P, np.float64)).to(dev)
    Q = torch.as_tensor(np.ascontiguousarray(Q, np.float64)).to(dev)
    wt = None if w is None else torch.as_tensor(np.ascontiguousarray(w, np.float64)).to(dev)
    T = torch.empty(16, dtype=torch.float64, device=dev)
    _ext.check(_ext.lib().lr_kabsch(P.data_ptr(), Q.data_ptr(), None if wt is None else wt.data_ptr(), P.shape[0],
                                     T.data_ptr(), _stream()))
    return T.cpu().numpy().reshape(4, 4)


def GC_RANSAC(A, B, distance_threshold, num_iterations, args, match_quality, return_mask=False):
    """GC_RANSAC.py:8-55: (pose 4x4 column-vector convention, elapsed seconds) -- the same estimator ``FR(codebase="GC")``
    runs: 3-point samples drawn without repetition, pre-verification by ``args.fast_rejection`` ("ELC" edge-length check |
    "SPRT" sequential probability ratio test over the first 256 pairs | "NONE"),
    PROSAC when ``args.prosac`` (pairs sorted by -match_quality, GC_RANSAC.py:39-43), MSAC scoring at GC-RANSAC's truncated
    threshold, GC-RANSAC's local optimisation unless ``args.GC_LO`` is False and a pre-verification is selected (GC_RANSAC.py:36-37;
    gcransac_python.cpp:518-521 honours the switch only in those branches), the final iterated least squares, confidence
    ``args.GC_conf``.  A non-zero ``args.spatial_coherence_weight`` raises (only the reference's default 0 is built).
    With return_mask the inlier mask pygcransac returns next to the pose is appended (in the caller's pair order)."""
    fast_rejection = getattr(args, "fast_rejection", "ELC")
    assert fast_rejection in PRECHECK, "unknown fast_rejection"
    if float(getattr(args, "spatial_coherence_weight", 0.0)) != 0.0:
        raise NotImplementedError("--spatial_coherence_weight != 0 is not implemented on the HIP path")
    use_elc = PRECHECK[fast_rejection]
    A = np.ascontiguousarray(A, np.float32); B = np.ascontiguousarray(B, np.float32)
    prosac = bool(getattr(args, "prosac", False)) and match_quality is not None
    order = None
    if prosac:
        fd = -np.asarray(match_quality, np.float32)
        fd[np.isnan(fd)] = np.inf
        order = np.argsort(fd, kind="stable")                # sort from best quality to worst (GC_RANSAC.py:40-43)
        A, B = A[order], B[order]
    start_time = time()
    T, info = ransac_dev(A, B, num_iterations, sample_size=3, use_elc=use_elc, thr=distance_threshold,
                         seed=getattr(args, "seed", DEFAULT_SEED), confidence=getattr(args, "GC_conf", 0.999),      # GC_RANSAC.py:26
                         sampler=1 if prosac else 2, scoring=2,                 # MSAC at the truncated threshold, as pygcransac scores models
                         local_opt=1 if (getattr(args, "GC_LO", True) or fast_rejection == "NONE") else 2, want_mask=return_mask)
    if info["best_h"] < 0:
        T = np.eye(4)                                       # GC_RANSAC.py:51-52
    elapsed = time() - start_time
    if return_mask:
        mask = info["mask"]
        if order is not None:
            unsorted = np.zeros_like(mask); unsorted[order] = mask; mask = unsorted
        return T, elapsed, mask
    return T, elapsed


def RANSAC_registration(pcd0, pcd1, idx0, idx1, distance_threshold, num_iterations, args):
    """FR.py:122-139 (Open3D registration_ransac_based_on_correspondence, ransac_n=4, edge-length checker)."""
    p0 = np.asarray(pcd0.points, np.float32)[np.asarray(idx0)]
    p1 = np.asarray(pcd1.points, np.float32)[np.asarray(idx1)]
    T, info = ransac_dev(p0, p1, num_iterations, sample_size=getattr(args, "ransac_n", 4), use_elc=True,
                         thr=distance_threshold, seed=getattr(args, "seed", DEFAULT_SEED),
                         confidence=getattr(args, "o3d_conf", 0.9995))                                               # FR.py:136
    return T if info["best_h"] >= 0 else np.eye(4)
