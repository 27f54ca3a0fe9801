"""The two sibling callers of the same hot path in the reference tree, as thin wrappers over FR():

* ``FCGF_FAST/net/RANSAC.py:137-194``  ``FCGF_RANSAC_tester.RANSAC``: mutual-NN + Open3D RANSAC (4-point, edge-length checker,
  500k iterations, confidence 0.9999, ``:226-243``) + LS refit over the original NN pairs (``:178-190``).
* ``DGR/core/deep_global_registration.py:461-565``  ``register_FCGF``: all NN pairs (``MUTUAL_ONLY = False``), the same RANSAC
  (``:60-76``), then an inverse-feature-distance weighted Procrustes refit over the inliers (``:519-537``).
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


def DGR_register_FCGF(xyz0, xyz1, feats0, feats1, iters=500 * 10 ** 3, seed=51, T_gt=None):
    """-> {'base': T, 'w_icp': T} like DGR/core/deep_global_registration.py:550 (ICP off); features are given, not extracted."""
    T_gt = np.eye(4) if T_gt is None else T_gt
    T, *_ = fr.FR(xyz0, xyz1, feats0, feats1, _Args(mode="no_filter", refit=3, iters=iters, seed=seed), T_gt)
    return {"base": T, "w_icp": T}
