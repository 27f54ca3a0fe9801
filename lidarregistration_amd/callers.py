"""The two sibling callers of the same hot path in the reference tree, as thin wrappers over FR():

* ``FCGF_FAST/net/RANSAC.py:137-194``  ``FCGF_RANSAC_tester.RANSAC``: mutual-NN + Open3D RANSAC (4-point, edge-length checker,
  500k iterations, confidence 0.9999, ``:226-243``) + LS refit over the original NN pairs (``:178-190``).
* ``DGR/core/deep_global_registration.py:461-565``  ``register_FCGF``: all NN pairs (``MUTUAL_ONLY = False``), the same RANSAC
  (``:60-76``), then an inverse-feature-distance weighted Procrustes refit over the inliers (``:519-537``), then ICP (``:556-563``).
"""
import numpy as np

from . import FR as fr
from .matching import measure_inlier_ratio


class _Args:
    codebase = "open3D"; ransac_n = 4; o3d_conf = 0.9999; iters = 500 * 10 ** 3; GPF_factor = 2.0; GPF_grid_wid = 10

    def __init__(self, **kw):
        self.__dict__.update(kw)


def FCGF_FAST_RANSAC(A, B, A_feat, B_feat, gt_motion, iters=500 * 10 ** 3, seed=51):
    """-> (T, elapsed_time, pcd0, pcd1, GT_inlier_ratio), the tuple of FCGF_FAST/net/RANSAC.py:194."""
    T, elapsed, pcd0, pcd1, _, ir_init, _, _ = fr.FR(A, B, A_feat, B_feat, _Args(mode="MNN", refit=1, iters=iters, seed=seed), gt_motion)
    return T, elapsed, pcd0, pcd1, ir_init


def DGR_register_FCGF(xyz0, xyz1, feats0, feats1, iters=500 * 10 ** 3, seed=51, T_gt=None, use_icp=True):
    """-> {'base': T, 'w_icp': T_icp} like DGR/core/deep_global_registration.py:554-565; features are given, not extracted.
    `use_icp` (the reference's ``self.use_icp = True``, :93): 'w_icp' is 'base' refined by point-to-point ICP with the correspondence
    distance 2 * voxel_size (:556-563, `registration_icp(pcd0, pcd1, voxel_size * 2, T, PointToPoint)`) -- lr_icp; else 'w_icp' = 'base'."""
    from .ransac import icp_dev
    T_gt = np.eye(4) if T_gt is None else T_gt
    T, *_ = fr.FR(xyz0, xyz1, feats0, feats1, _Args(mode="no_filter", refit=3, iters=iters, seed=seed), T_gt)
    res = {"base": T, "w_icp": T}
    if use_icp:
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        x0 = torch.as_tensor(np.asarray(xyz0), dtype=torch.float32).to(dev).contiguous(); x1 = torch.as_tensor(np.asarray(xyz1), dtype=torch.float32).to(dev).contiguous()
        res["w_icp"], _ = icp_dev(x0, x1, T, max_dist=2 * fr.VOXEL_SIZE)
    return res
