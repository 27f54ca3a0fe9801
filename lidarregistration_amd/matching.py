"""Host-side mirror of the reference's ``Experiments/algorithms/matching.py`` on top of liblidarreg.so.

Same function names, argument order and return arity as the reference (including the return-arity
quirk of ``nn_to_mutual``, matching.py:233-239), so callers such as ``FR`` and the TEASER wrapper can
import this module instead.  Inputs are torch tensors (moved to the current HIP device if needed);
index outputs are int64 CPU tensors like the reference's ``.cpu()`` results.  All arithmetic happens
in the HIP kernels; torch only provides device memory and the stream.
"""
from time import time

import numpy as np
import torch

from . import _ext

_WS = {}


def _device():
    if not torch.cuda.is_available():
        raise _ext.LidarRegError("no HIP device visible: lidarregistration_amd has no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _stream():
    return torch.cuda.current_stream().cuda_stream


def workspace(n0, n1, iters=0, dim=32):
    """Cached per-device workspace large enough for (n0, n1, iters)."""
    dev = torch.cuda.current_device()
    ws = _WS.get(dev)
    if ws is None or not ws.handle or not ws.fits(n0, n1, iters) or ws.dim != dim:
        grow = (ws.max_n0, ws.max_n1, ws.max_iters) if ws is not None else (0, 0, 0)
        if ws is not None:
            _WS.pop(dev, None)
            torch.cuda.synchronize()
            ws.close()
        ws = _ext.Workspace(max(n0, grow[0]), max(n1, grow[1]), dim, max(iters, grow[2], 1))
        _WS[dev] = ws
    return ws


def _f32(t):
    return torch.as_tensor(t).to(device=_device(), dtype=torch.float32).contiguous()


def _i32(t):
    return torch.as_tensor(t).to(device=_device(), dtype=torch.int32).contiguous()


def _ptr(t):
    return None if t is None else t.data_ptr()


# ----------------------------------------------------------------------------- device-level primitives

def nn_top2_dev(F0, F1, want_2nd=True, want_dist=False):
    """(idx1, idx2, s1, s2) as device tensors (int32 / float32); unused ones are None."""
    F0, F1 = _f32(F0), _f32(F1)
    n0, n1, d = F0.shape[0], F1.shape[0], F0.shape[1]
    ws = workspace(n0, n1, dim=d)
    dev = F0.device
    idx1 = torch.empty(n0, dtype=torch.int32, device=dev)
    idx2 = torch.empty(n0, dtype=torch.int32, device=dev) if want_2nd else None
    s1 = torch.empty(n0, dtype=torch.float32, device=dev) if want_dist else None
    s2 = torch.empty(n0, dtype=torch.float32, device=dev) if (want_dist and want_2nd) else None
    _ext.check(_ext.lib().lr_nn_top2(ws.handle, F0.data_ptr(), n0, F1.data_ptr(), n1, d,
                                      idx1.data_ptr(), _ptr(idx2), _ptr(s1), _ptr(s2), _stream()))
    return idx1, idx2, s1, s2


def mutual_dev(F0, F1, idx1, idx2=None):
    """(is_bb uint8 [n0], out_idx0, out_idx1, out_idx2 or None) device tensors, survivors in ascending idx0."""
    F0, F1, idx1 = _f32(F0), _f32(F1), _i32(idx1)
    idx2 = None if idx2 is None else _i32(idx2)
    n0, n1, d = F0.shape[0], F1.shape[0], F0.shape[1]
    ws = workspace(n0, n1, dim=d)
    dev = F0.device
    is_bb = torch.empty(n0, dtype=torch.uint8, device=dev)
    o0 = torch.empty(n0, dtype=torch.int32, device=dev)
    o1 = torch.empty(n0, dtype=torch.int32, device=dev)
    o2 = torch.empty(n0, dtype=torch.int32, device=dev) if idx2 is not None else None
    cnt = torch.zeros(1, dtype=torch.int32, device=dev)
    _ext.check(_ext.lib().lr_nn_to_mutual(ws.handle, F0.data_ptr(), n0, F1.data_ptr(), n1, d, idx1.data_ptr(), _ptr(idx2),
                                           is_bb.data_ptr(), o0.data_ptr(), o1.data_ptr(), _ptr(o2), cnt.data_ptr(), _stream()))
    m = int(cnt.item())
    return is_bb, o0[:m], o1[:m], (o2[:m] if o2 is not None else None)


# ----------------------------------------------------------------------------- reference-shaped API

def find_nn(F0, F1, return_2nd=False):
    """matching.py:22-65."""
    idx1, idx2, _, _ = nn_top2_dev(F0, F1, want_2nd=return_2nd)
    corres_idx0 = torch.arange(idx1.shape[0]).long()
    corres_idx1 = idx1.long().cpu()
    if return_2nd:
        return corres_idx0, corres_idx1, idx2.long().cpu()
    return corres_idx0, corres_idx1, None


def find_2nn(fcgf_feats0, fcgf_feats1):
    """matching.py:6-19.  The reference runs the NN twice to bill only the 2nd-NN surcharge; the fused
    kernel has no such surcharge, so the by-product time is reported as 0."""
    corres_idx0, corres_idx1, idx1_2nd = find_nn(fcgf_feats0, fcgf_feats1, return_2nd=True)
    return corres_idx0, corres_idx1, idx1_2nd, 0.0


def nn_to_mutual(feats0, feats1, corres_idx0, corres_idx1, idx1_2nd=None, force_return_2nd=False):
    """matching.py:222-239 (relies, like the reference, on corres_idx0 == arange(N0))."""
    _, o0, o1, o2 = mutual_dev(feats0, feats1, corres_idx1, idx1_2nd)
    final_corres_idx0, final_corres_idx1 = o0.long().cpu(), o1.long().cpu()
    if idx1_2nd is not None:
        return final_corres_idx0, final_corres_idx1, o2.long().cpu()
    elif force_return_2nd:
        return final_corres_idx0, final_corres_idx1, None
    else:
        return final_corres_idx0, final_corres_idx1


def mark_best_buddies(fcgf_feats0, fcgf_feats1, corres_idx0, corres_idx1):
    """matching.py:207-220."""
    is_bb, _, _, _ = mutual_dev(fcgf_feats0, fcgf_feats1, corres_idx1)
    is_bb = is_bb.bool().cpu().numpy()
    return is_bb, is_bb.sum()


def calc_distance_ratio_in_feature_space(fcgf_feats0, fcgf_feats1, corres_idx0, corres_idx1, idx1_2nd):
    """matching.py:89-98 (returns a device tensor, as the reference does)."""
    F0, F1 = _f32(fcgf_feats0), _f32(fcgf_feats1)
    i0, i1, i2 = _i32(corres_idx0), _i32(corres_idx1), _i32(idx1_2nd)
    out = torch.empty(i0.shape[0], dtype=torch.float32, device=F0.device)
    _ext.check(_ext.lib().lr_feat_ratio(F0.data_ptr(), F1.data_ptr(), F0.shape[1], i0.shape[0],
                                         i0.data_ptr(), i1.data_ptr(), i2.data_ptr(), out.data_ptr(), _stream()))
    return out


def Grid_Prioritized_Filter(fcgf_feats0, fcgf_feats1, corres_idx0, corres_idx1, idx1_2nd, xyz0, args, BB_first=False):
    """matching.py:100-205, both forms (``BB_first=True`` is what the reference's TEASER wrapper calls)."""
    F0, F1 = _f32(fcgf_feats0), _f32(fcgf_feats1)
    i1, i2 = _i32(corres_idx1), _i32(idx1_2nd)
    xyz = _f32(xyz0)
    n0, n1, d = F0.shape[0], F1.shape[0], F0.shape[1]
    ws = workspace(n0, n1, dim=d)
    dev = F0.device
    o0 = torch.empty(n0, dtype=torch.int32, device=dev); o1 = torch.empty_like(o0); o2 = torch.empty_like(o0)
    sc = torch.empty(n0, dtype=torch.float32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    if BB_first:
        _ext.check(_ext.lib().lr_gpf_bb_first(ws.handle, F0.data_ptr(), n0, F1.data_ptr(), n1, d, i1.data_ptr(), i2.data_ptr(),
                                               xyz.data_ptr(), int(args.GPF_grid_wid), float(args.GPF_max_matches),
                                               o0.data_ptr(), o1.data_ptr(), o2.data_ptr(), sc.data_ptr(), cnt.data_ptr(),
                                               cnt[1:].data_ptr(), _stream()))
        m, has_score = [int(v) for v in cnt.cpu()]
        return (o0[:m].long().cpu(), o1[:m].long().cpu(), o2[:m].long().cpu(),
                corres_idx0, corres_idx1, idx1_2nd, sc[:m] if has_score else None)
    _ext.check(_ext.lib().lr_gpf(ws.handle, F0.data_ptr(), n0, F1.data_ptr(), n1, d, i1.data_ptr(), i2.data_ptr(),
                                  xyz.data_ptr(), int(args.GPF_grid_wid), float(args.GPF_factor),
                                  o0.data_ptr(), o1.data_ptr(), o2.data_ptr(), sc.data_ptr(), cnt.data_ptr(), _stream()))
    m = int(cnt[0].item())
    return (o0[:m].long().cpu(), o1[:m].long().cpu(), o2[:m].long().cpu(),
            corres_idx0, corres_idx1, idx1_2nd, sc[:m])


def measure_inlier_ratio(corres_idx0, corres_idx1, pcd0, pcd1, T_gt, voxel_size):
    """matching.py:241-249 (statistics only, outside the timed region; plain numpy on the host)."""
    i0 = np.asarray(corres_idx0); i1 = np.asarray(corres_idx1)
    p = np.asarray(pcd0.points, np.float64); q = np.asarray(pcd1.points, np.float64)
    T = np.asarray(T_gt, np.float64)
    pt = p @ T[:3, :3].T + T[:3, 3]
    if len(i0) == 0:
        return 0.0
    dist2 = np.sum((pt[i0, :] - q[i1, :]) ** 2, axis=1)
    return float((dist2 < (2 * voxel_size) ** 2).sum()) / len(dist2)
