"""``pygcransac.findRigidTransform`` -- the reference's native call shape on liblidarreg.so.

The reference's native boundary for RANSAC is ``pygcransac.findRigidTransform(x1y1z1, x2y2z2, **params) -> (pose, mask)``
(call sites Experiments/algorithms/GC_RANSAC.py:12-22,46-49 and Experiments/baseline_scripts/baseline_3DMatch.py:107-116;
native side ``findRigidTransform_``, GC-RANSAC/src/pygcransac/src/gcransac_python.cpp:404-624).  This module keeps that
signature, its sentinel overloading and its output conventions, so the reference's own ``GC_RANSAC.py`` can import it
unchanged (``Experiments/pygcransac.py`` re-exports it under the name the reference imports):

* ``use_sprt`` means "perform fast rejection" (GC_RANSAC.py:29-32); with it, a NEGATIVE ``min_inlier_ratio_for_sprt`` selects
  the edge-length check instead of SPRT (GC_RANSAC.py:33-34 <-> gcransac_python.cpp:500);
* a NON-ZERO ``neighborhood`` means "no local optimisation" (GC_RANSAC.py:36-37 <-> gcransac_python.cpp:418-423) -- honoured only
  in the two branches with a pre-verification (:518-521, :553-556); without one the optimisation always runs (:571-591);
* ``sampler`` 0 = uniform, 1 = PROSAC (the points must come best quality first), anything else is an error: message on
  stderr, no model (:463-479);
* settings the wrapper hard-codes: at most 20 local optimisations and no exit before 20 iterations with a pre-verification,
  50 / 50 without (:513-517, :579-583) -- the library's defaults for ``lo_max_calls`` / ``min_iters``;
* the pose comes back in ROW-vector convention (the transpose of ``T`` with ``q = R p + t``; GC_RANSAC.py:55 transposes it
  back), float64, or ``None`` when no model was found; ``mask`` is a bool array over the M input pairs, the inliers of
  the returned model at the estimator's own (truncated) threshold (:594-603).

Behind it: ``lr_ransac`` (3-point samples without repetition, MSAC at the truncated threshold, local optimisation, final
iterated least squares) + ``lr_inlier_mask``.  Deviations, stated once: the hypothesis stream is this library's (Philox keyed by
``seed``, a module attribute -- the native signature has no seed); ``spatial_coherence_weight != 0`` raises
``NotImplementedError`` (the graph-cut term is not built; the reference runs it at 0), which also makes ``neighborhood_size``
(the cell size of the FLANN graph, :444-446) unused; SPRT starts from the wrapper's own 0.1 (other values raise).
"""
import sys

import numpy as np

from . import ransac as _ransac

seed = _ransac.DEFAULT_SEED      # hypothesis stream of the next calls (the reference seeds numpy / torch globally, test.py:357)


def findRigidTransform(x1y1z1, x2y2z2, threshold=1.0, conf=0.99, spatial_coherence_weight=0.975, max_iters=10000, use_sprt=True,
                       min_inlier_ratio_for_sprt=0.1, sampler=1, neighborhood=0, neighborhood_size=20.0):
    """(pose 4x4 float64 in row-vector convention or None, mask bool[M]).  Defaults as upstream's binding (both call sites in
    the reference pass the arguments that matter by keyword)."""
    A = np.ascontiguousarray(x1y1z1, np.float32); B = np.ascontiguousarray(x2y2z2, np.float32)
    if A.ndim != 2 or A.shape[1] != 3 or A.shape != B.shape:
        raise ValueError("findRigidTransform: x1y1z1 and x2y2z2 must both be [M,3]")
    m = A.shape[0]
    if float(spatial_coherence_weight) != 0.0:
        raise NotImplementedError("spatial_coherence_weight != 0 is not implemented on the HIP path (the reference runs GC-RANSAC at 0)")
    do_local_optimization = int(neighborhood) == 0                       # gcransac_python.cpp:418-423
    sampler = int(sampler)
    if sampler not in (0, 1):                                            # :463-479: message, zero inliers -> no pose
        print("Unknown sampler identifier: %d. The accepted samplers are 0 (uniform sampling), 1 (PROSAC sampling)" % sampler, file=sys.stderr)
        return None, np.zeros(m, bool)
    if use_sprt:
        if float(min_inlier_ratio_for_sprt) < 0.0:
            precheck = _ransac.PRECHECK["ELC"]                           # :500-532
        else:
            if abs(float(min_inlier_ratio_for_sprt) - 0.1) > 1e-12:
                raise NotImplementedError("SPRT starts from min_inlier_ratio_for_sprt = 0.1 on the HIP path (the value both call sites of the reference pass)")
            precheck = _ransac.PRECHECK["SPRT"]                          # :534-568
        local_opt = 1 if do_local_optimization else 2                    # :518-521, :553-556 (2: the final least squares only)
    else:
        precheck = _ransac.PRECHECK["NONE"]                              # :571-591: do_local_optimization is never read
        local_opt = 1
    T, info = _ransac.ransac_dev(A, B, int(max_iters), sample_size=3, use_elc=precheck, thr=float(threshold), seed=seed, confidence=float(conf),
                                 sampler=1 if sampler == 1 else 2, scoring=2, local_opt=local_opt, want_mask=True)
    if info["best_h"] < 0 or info["n_inliers"] == 0:
        return None, np.zeros(m, bool)
    return np.ascontiguousarray(T.T), info["mask"]
