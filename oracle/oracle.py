"""CPU oracle for the registration hot path -- TEST INFRASTRUCTURE, not product code.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product package (``lidarregistration_amd``) never does and fails loudly when
its HIP library is missing.

The functions keep the names, argument order and return arity of the reference's
``Experiments/algorithms/matching.py`` so that parity tests read like calls into the reference;
tensors are plain numpy arrays here.  Arithmetic-critical inner loops (feature-space distances,
RANSAC scoring, Kabsch) live in ``oracle.c`` where every fma is explicit; the index/bookkeeping
logic (mutual intersection, Grid-Prioritized Filter) is restated in numpy below, each function
citing the reference lines it follows (paths relative to the reference tree).

Parity status: NN / MNN / best-buddies / ratio / GPF / Kabsch / metric are pinned by golden vectors
captured from the importable reference (``tests/golden/make_golden.py``), the composed ``FR()`` pipeline by
fixture G11.  The RANSAC loop and its GC options (PROSAC, SPRT, local optimisation), ICP and
``sparse_quantize`` are third-party (Open3D 0.13.0 / pygcransac 0.1 / MinkowskiEngine 0.5.4, neither
vendored nor installable here): PARITY UNPINNED for those; see DESIGN.md section 5.
"""
import ctypes
import os
import subprocess
from copy import deepcopy

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i32p = ctypes.POINTER(ctypes.c_int32)


class RansacParams(ctypes.Structure):
    _fields_ = [("sample_size", ctypes.c_int32), ("use_elc", ctypes.c_int32),
                ("thr2", ctypes.c_float), ("iters", ctypes.c_int32), ("seed", ctypes.c_uint64),
                ("confidence", ctypes.c_float), ("batch", ctypes.c_int32),
                ("sampler", ctypes.c_int32), ("prosac_growth", ctypes.c_int32), ("scoring", ctypes.c_int32), ("local_opt", ctypes.c_int32),
                ("lo_rounds", ctypes.c_int32), ("lo_trials", ctypes.c_int32), ("lo_max_calls", ctypes.c_int32), ("min_iters", ctypes.c_int32)]


class RansacResult(ctypes.Structure):
    _fields_ = [("best_h", ctypes.c_int64), ("best_count", ctypes.c_uint32),
                ("best_ssq", ctypes.c_uint64), ("n_valid", ctypes.c_int64), ("n_ids", ctypes.c_int64)]


def build(force=False):
    """Compile oracle.c -> liboracle.so (gcc)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.orc_num_threads.restype = ctypes.c_int
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _p(a, t):
    return a.ctypes.data_as(t)


# ----------------------------------------------------------------------------- NN (a1, a2)

def row_norms(F):
    F = _f32(F)
    out = np.empty(F.shape[0], np.float32)
    lib().orc_row_norms(_p(F, c_f32p), F.shape[0], F.shape[1], _p(out, c_f32p))
    return out


def nn_top2(F0, F1):
    """(idx1, idx2, s1, s2) int32/float32 -- raw form of matching.py:25-41."""
    F0, F1 = _f32(F0), _f32(F1)
    n0, n1, d = F0.shape[0], F1.shape[0], F0.shape[1]
    idx1 = np.empty(n0, np.int32); idx2 = np.empty(n0, np.int32)
    s1 = np.empty(n0, np.float32); s2 = np.empty(n0, np.float32)
    lib().orc_nn_top2(_p(F0, c_f32p), n0, _p(F1, c_f32p), n1, d,
                      _p(idx1, c_i32p), _p(idx2, c_i32p), _p(s1, c_f32p), _p(s2, c_f32p))
    return idx1, idx2, s1, s2


def find_nn(F0, F1, return_2nd=False):
    """matching.py:22-65."""
    idx1, idx2, _, _ = nn_top2(F0, F1)
    corres_idx0 = np.arange(len(idx1), dtype=np.int64)
    corres_idx1 = idx1.astype(np.int64)
    if return_2nd:
        return corres_idx0, corres_idx1, idx2.astype(np.int64)
    return corres_idx0, corres_idx1, None


def find_2nn(fcgf_feats0, fcgf_feats1):
    """matching.py:6-19 (the timing by-product is 0.0 here)."""
    i0, i1, i2 = find_nn(fcgf_feats0, fcgf_feats1, return_2nd=True)
    return i0, i1, i2, 0.0


# ----------------------------------------------------------------------------- mutual (a3-a5)

def torch_intersect(Na, Nb, i_ab, j_ab, i_ba, j_ba):
    """matching.py:67-87: entries whose summed sparse value is exactly 2, in coalesce (i, j) order."""
    keys = np.concatenate([np.asarray(i_ab, np.int64) * Nb + np.asarray(j_ab, np.int64),
                           np.asarray(i_ba, np.int64) * Nb + np.asarray(j_ba, np.int64)])
    uniq, cnt = np.unique(keys, return_counts=True)
    both = uniq[cnt == 2]
    return both // Nb, both % Nb


def nn_to_mutual(feats0, feats1, corres_idx0, corres_idx1, idx1_2nd=None, force_return_2nd=False):
    """matching.py:222-239, including its return-arity quirk."""
    uniq_inds_1 = np.unique(corres_idx1)
    _, inv_corres_idx0, _ = find_nn(np.asarray(feats1)[uniq_inds_1, :], feats0, False)
    inv_corres_idx1 = uniq_inds_1
    final_corres_idx0, final_corres_idx1 = torch_intersect(
        np.asarray(feats0).shape[0], np.asarray(feats1).shape[0],
        corres_idx0, corres_idx1, inv_corres_idx0, inv_corres_idx1)
    if idx1_2nd is not None:
        idx1_2nd = np.asarray(idx1_2nd)[final_corres_idx0]
        return final_corres_idx0, final_corres_idx1, idx1_2nd
    elif force_return_2nd:
        return final_corres_idx0, final_corres_idx1, None
    else:
        return final_corres_idx0, final_corres_idx1


def mark_best_buddies(fcgf_feats0, fcgf_feats1, corres_idx0, corres_idx1):
    """matching.py:207-220."""
    bb_idx0, bb_idx1 = nn_to_mutual(fcgf_feats0, fcgf_feats1, corres_idx0, corres_idx1)
    P = 1 + np.max(corres_idx0)
    is_bb = np.isin(P * np.asarray(corres_idx1) + np.asarray(corres_idx0), P * bb_idx1 + bb_idx0)
    return is_bb, is_bb.sum()


# ----------------------------------------------------------------------------- ratio (a6)

def calc_distance_ratio_in_feature_space(fcgf_feats0, fcgf_feats1, corres_idx0, corres_idx1, idx1_2nd):
    """matching.py:89-98."""
    F0, F1 = _f32(fcgf_feats0), _f32(fcgf_feats1)
    i0, i1, i2 = _i32(corres_idx0), _i32(corres_idx1), _i32(idx1_2nd)
    out = np.empty(len(i0), np.float32)
    lib().orc_feat_ratio(_p(F0, c_f32p), _p(F1, c_f32p), F0.shape[1], len(i0),
                         _p(i0, c_i32p), _p(i1, c_i32p), _p(i2, c_i32p), _p(out, c_f32p))
    return out


# ----------------------------------------------------------------------------- GPF (a7)

def gpf_water_fill(max_per_quad, TOTAL_NUM):
    """matching.py:154-179: approximate water-filling by bisection on the cap height."""
    def apply_height(height):
        is_dwarf = max_per_quad < height
        return is_dwarf * max_per_quad + (~is_dwarf) * height

    max_height = TOTAL_NUM
    min_height = 0
    curr_height = (max_height + min_height) / 2
    while np.abs(max_height - min_height) > 2:
        cur_total = apply_height(curr_height).sum()
        if cur_total == TOTAL_NUM:
            break
        elif cur_total < TOTAL_NUM:
            min_height = curr_height
        elif cur_total > TOTAL_NUM:
            max_height = curr_height
        curr_height = (max_height + min_height) / 2
    return apply_height(np.round(curr_height))


def _to_quads(X, GRID_WID):
    """matching.py:136-141 in float32, as torch evaluates it."""
    X = X.astype(np.float32)
    m, M = X.min(), X.max()
    den = np.float32(np.float32(M - m) + np.float32(1e-3))
    X_ = ((X - m) / den).astype(np.float32)
    return np.floor(np.float32(GRID_WID) * X_)


def Grid_Prioritized_Filter(fcgf_feats0, fcgf_feats1, corres_idx0, corres_idx1, idx1_2nd, xyz0, args, BB_first=False):
    """matching.py:100-205."""
    corres_idx0_orig = deepcopy(corres_idx0)
    corres_idx1_orig = deepcopy(corres_idx1)
    idx1_2nd_orig = deepcopy(idx1_2nd)
    GRID_WID = args.GPF_grid_wid

    if BB_first:
        TOTAL_NUM = args.GPF_max_matches
        corres_idx0, corres_idx1, idx1_2nd = nn_to_mutual(fcgf_feats0, fcgf_feats1, corres_idx0, corres_idx1,
                                                         idx1_2nd, force_return_2nd=True)
        if TOTAL_NUM >= corres_idx0.shape[0]:
            return corres_idx0, corres_idx1, idx1_2nd, corres_idx0_orig, corres_idx1_orig, idx1_2nd_orig, None
    else:
        is_bb, num_bb = mark_best_buddies(fcgf_feats0, fcgf_feats1, corres_idx0, corres_idx1)
        TOTAL_NUM = args.GPF_factor * num_bb

    feat_dist = calc_distance_ratio_in_feature_space(fcgf_feats0, fcgf_feats1, corres_idx0, corres_idx1, idx1_2nd)
    m, M = feat_dist.min(), feat_dist.max()
    norm_feat_dist = ((feat_dist - m) / np.float32(M - m)).astype(np.float32)
    if not BB_first:
        norm_feat_dist[is_bb] -= np.float32(1)

    xyz0 = np.asarray(xyz0, dtype=np.float32)
    quadrant_i = _to_quads(xyz0[corres_idx0, 0], GRID_WID)
    quadrant_j = _to_quads(xyz0[corres_idx0, 1], GRID_WID)
    cell = (quadrant_i * GRID_WID + quadrant_j).astype(np.int64)
    max_per_quad = np.bincount(cell, minlength=GRID_WID * GRID_WID).astype(np.float64).reshape(GRID_WID, GRID_WID)

    per_quad = gpf_water_fill(max_per_quad, TOTAL_NUM)

    keep = np.zeros(len(norm_feat_dist), dtype=bool)
    for qi in range(GRID_WID):
        for qj in range(GRID_WID):
            extra_per_quad = int(per_quad[qi, qj])
            if extra_per_quad > 0:
                is_cand = cell == (qi * GRID_WID + qj)
                if per_quad[qi, qj] == max_per_quad[qi, qj]:
                    keep[is_cand] = True
                else:
                    is_cand_inds = is_cand.nonzero()[0]
                    # torch.argsort is not stable; ties are resolved towards the lower position here
                    order = np.argsort(norm_feat_dist[is_cand], kind="stable")
                    keep[is_cand_inds[order[:extra_per_quad]]] = True

    corres_idx0 = corres_idx0[keep]
    corres_idx1 = corres_idx1[keep]
    norm_feat_dist = norm_feat_dist[keep]
    idx1_2nd = idx1_2nd[keep] if idx1_2nd is not None else None
    return corres_idx0, corres_idx1, idx1_2nd, corres_idx0_orig, corres_idx1_orig, idx1_2nd_orig, norm_feat_dist


# ----------------------------------------------------------------------------- voxel de-duplication (f2)

def sparse_quantize(coordinates, return_index=True):
    """ME.utils.sparse_quantize(coordinates, return_index=True) as the reference's loaders use it
    (dataloader/generic_balanced_loader.py:62-63).  MinkowskiEngine 0.5.4 (Requirements/conda_GC_full.yml:106) is not vendored:
    PARITY UNPINNED.  Its CPU path floors the coordinates and inserts the rows one after the other into a hash map, so the
    kept row of every occupied cell is its first one, and the kept indices come out in ascending order."""
    cells = np.floor(np.asarray(coordinates, np.float64)).astype(np.int64)
    _, first = np.unique(cells, axis=0, return_index=True)
    sel = np.sort(first)
    return (cells[sel].astype(np.int32), sel.astype(np.int64)) if return_index else cells[sel].astype(np.int32)


# ----------------------------------------------------------------------------- stats (a8)

def measure_inlier_ratio(corres_idx0, corres_idx1, xyz0, xyz1, T_gt, voxel_size):
    """matching.py:241-249 on raw (N,3) arrays instead of Open3D clouds."""
    p = np.asarray(xyz0, np.float64)
    q = np.asarray(xyz1, np.float64)
    T = np.asarray(T_gt, np.float64)
    pt = p @ T[:3, :3].T + T[:3, 3]
    dist2 = np.sum((pt[corres_idx0, :] - q[corres_idx1, :]) ** 2, axis=1)
    is_close = dist2 < (2 * voxel_size) ** 2
    return float(is_close.sum()) / len(is_close)


# ----------------------------------------------------------------------------- Kabsch (a13)

def kabsch(P, Q, w=None):
    """Least-squares rigid fit Q ~ R P + t (models/common.py:7-45).  Returns 4x4 float64."""
    P = np.ascontiguousarray(P, np.float64); Q = np.ascontiguousarray(Q, np.float64)
    T = np.empty(16, np.float64)
    wp = None
    if w is not None:
        w = np.ascontiguousarray(w, np.float64)
        wp = _p(w, c_f64p)
    lib().orc_kabsch_points(_p(P, c_f64p), _p(Q, c_f64p), wp, P.shape[0], _p(T, c_f64p))
    return T.reshape(4, 4)


def elc(src, tgt, sample):
    src, tgt, s = _f32(src), _f32(tgt), _i32(sample)
    return bool(lib().orc_elc(_p(src, c_f32p), _p(tgt, c_f32p), _p(s, c_i32p), len(s)))


def philox(seed, h):
    out = (ctypes.c_uint32 * 4)()
    lib().orc_philox(ctypes.c_uint64(seed), ctypes.c_uint64(h), out)
    return np.array(out[:], np.uint32)


# ----------------------------------------------------------------------------- RANSAC (a10) + refit (a11)

def _params(sample_size, use_elc, thr, iters, seed, confidence=1.0, batch=0, sampler=0, prosac_growth=0, scoring=0, local_opt=0,
            lo_rounds=0, lo_trials=0, lo_max_calls=0, min_iters=0):
    # (the library rejects these with LR_EINVAL, lr_ransac.hip: the checker refuses them too instead of clamping silently)
    if not (0 <= int(lo_trials) <= 20 and int(lo_rounds) >= 0 and int(lo_max_calls) >= 0 and int(min_iters) >= 0):
        raise ValueError("lo_rounds, lo_trials (<= 20), lo_max_calls and min_iters must be >= 0 (0 = default)")
    return RansacParams(sample_size, int(use_elc), np.float32(float(thr) * float(thr)), iters, seed, confidence, batch,
                        int(sampler), int(prosac_growth), int(scoring), int(local_opt), int(lo_rounds), int(lo_trials), int(lo_max_calls),
                        int(min_iters))


def prosac_order(feat_dist):
    """GC_RANSAC.py:39-43: ord = argsort(-match_quality) with match_quality = -feat_dist (FR.py:80); ties by index, NaN last."""
    fd = np.asarray(feat_dist, np.float32).copy()
    fd[np.isnan(fd)] = np.inf
    return np.argsort(fd, kind="stable")


def hypothesis(src, tgt, h, sample_size=3, use_elc=True, thr=0.6, seed=51, sampler=0, prosac_growth=0):
    src, tgt = _f32(src), _f32(tgt)
    T = np.empty(16, np.float64); s = np.zeros(4, np.int32)
    p = _params(sample_size, use_elc, thr, 0, seed, sampler=sampler, prosac_growth=prosac_growth)
    ok = lib().orc_hypothesis(_p(src, c_f32p), _p(tgt, c_f32p), src.shape[0], ctypes.byref(p),
                              ctypes.c_uint64(h), _p(T, c_f64p), _p(s, c_i32p))
    return bool(ok), T.reshape(4, 4), s[:sample_size]


TRUNCATED_THR2 = float(np.float32(0.6 * 0.6) * np.float32(2.25))      # scoring = 2 at the reference's 0.6 m: the float the loops test against


def score(src, tgt, T, thr=0.6, thr2=None):
    src, tgt = _f32(src), _f32(tgt)
    T = np.ascontiguousarray(T, np.float64)
    c = ctypes.c_uint32(); q = ctypes.c_uint64()
    lib().orc_score(_p(src, c_f32p), _p(tgt, c_f32p), src.shape[0], _p(T, c_f64p),
                    ctypes.c_float(np.float32(float(thr) * float(thr)) if thr2 is None else np.float32(thr2)), ctypes.byref(c), ctypes.byref(q))
    return c.value, q.value


def ransac(src, tgt, iters, sample_size=3, use_elc=True, thr=0.6, seed=51, confidence=1.0, batch=0, sampler=0, prosac_growth=0,
           scoring=0, local_opt=0, lo_rounds=0, lo_trials=0, lo_max_calls=0, min_iters=0, sequential=False):
    """RANSAC over M correspondences src[i] <-> tgt[i] (sampler 1: PROSAC, pairs best quality first; 2: uniform with unique
    indices; scoring 1: MSAC, 2: MSAC at GC-RANSAC's truncated threshold; local_opt 1: GC-RANSAC's local optimisation + final
    iterated least squares, 2: the latter only).  sequential=True: the per-iteration reference mode (orc_ransac_seq) instead of the
    batched loop the HIP kernels implement.  Returns (T 4x4 float64, info dict)."""
    src, tgt = _f32(src), _f32(tgt)
    T = np.empty(16, np.float64)
    p = _params(sample_size, use_elc, thr, iters, seed, confidence, batch, sampler, prosac_growth, scoring, local_opt, lo_rounds, lo_trials,
                lo_max_calls, min_iters)
    r = RansacResult()
    fn = lib().orc_ransac_seq if sequential else lib().orc_ransac
    fn(_p(src, c_f32p), _p(tgt, c_f32p), src.shape[0], ctypes.byref(p), _p(T, c_f64p), ctypes.byref(r))
    return T.reshape(4, 4), dict(best_h=r.best_h, best_count=r.best_count, best_ssq=r.best_ssq, n_valid=r.n_valid, n_ids=r.n_ids)


def refit(xyz0, xyz1, idx1, T, thr=0.6, feats0=None, feats1=None):
    """FR.py:99-111 (with feats: DGR's inverse-feature-distance weighted form).  Returns (T 4x4 float64, inlier count)."""
    xyz0, xyz1, idx1 = _f32(xyz0), _f32(xyz1), _i32(idx1)
    Tin = np.ascontiguousarray(T, np.float64).reshape(16)
    Tout = np.empty(16, np.float64)
    if feats0 is not None:
        F0, F1 = _f32(feats0), _f32(feats1)
        n = lib().orc_refit_weighted(_p(xyz0, c_f32p), xyz0.shape[0], _p(xyz1, c_f32p), _p(idx1, c_i32p), _p(Tin, c_f64p),
                                     ctypes.c_double(float(thr) * float(thr)), _p(Tout, c_f64p), _p(F0, c_f32p), _p(F1, c_f32p))
        return Tout.reshape(4, 4), n
    n = lib().orc_refit(_p(xyz0, c_f32p), xyz0.shape[0], _p(xyz1, c_f32p), _p(idx1, c_i32p),
                        _p(Tin, c_f64p), ctypes.c_double(float(thr) * float(thr)), _p(Tout, c_f64p))
    return Tout.reshape(4, 4), n


# ----------------------------------------------------------------------------- ICP (f1)

def icp(src, tgt, T_init, max_dist=0.6, max_iter=30, rel_fitness=1e-6, rel_rmse=1e-6):
    """Point-to-point ICP as Experiments/test.py:183-189 calls Open3D's registration_icp.  Returns (T, info)."""
    src, tgt = _f32(src), _f32(tgt)
    Tin = np.ascontiguousarray(T_init, np.float64).reshape(16)
    Tout = np.empty(16, np.float64); out = np.zeros(4, np.float64)
    lib().orc_icp(_p(src, c_f32p), src.shape[0], _p(tgt, c_f32p), tgt.shape[0], _p(Tin, c_f64p), ctypes.c_double(max_dist),
                  int(max_iter), ctypes.c_double(rel_fitness), ctypes.c_double(rel_rmse), _p(Tout, c_f64p), _p(out, c_f64p))
    return Tout.reshape(4, 4), dict(fitness=out[0], inlier_rmse=out[1], n_corr=int(out[2]), iterations=int(out[3]))


# ----------------------------------------------------------------------------- metric (a15)

def rotation_error_deg(T, T_gt):
    """libs/loss.py:44,48 evaluated in float64."""
    R, Rg = np.asarray(T, np.float64)[:3, :3], np.asarray(T_gt, np.float64)[:3, :3]
    c = np.clip((np.trace(R.T @ Rg) - 1) / 2.0, -1, 1)
    return float(np.degrees(np.arccos(c)))


def translation_error_cm(T, T_gt):
    """libs/loss.py:45,49."""
    return float(np.linalg.norm(np.asarray(T, np.float64)[:3, 3] - np.asarray(T_gt, np.float64)[:3, 3]) * 100)


# ----------------------------------------------------------------------------- whole pair (a9)

def lo_sample(seed, call, rnd, trial, n):
    """The 21 distinct positions of [0, n) the local optimisation draws for (call, round, trial)."""
    pos = np.zeros(21, np.int32)
    lib().orc_lo_sample(ctypes.c_uint64(seed), int(call), int(rnd), int(trial), int(n), _p(pos, c_i32p))
    return pos


def register_pair(xyz0, xyz1, feats0, feats1, mode="MNN", iters=50000, sample_size=3, use_elc=True,
                  thr=0.6, seed=51, args=None, refit_on_orig=True, confidence=1.0, batch=0, prosac=False, scoring=0,
                  local_opt=0, unique=False):
    """FR.py:16-119 with the open3D-codebase ordering: NN -> filter -> RANSAC -> LS refit on the
    original NN pairs.  Returns dict(T, idx0, idx1, idx1_orig, ransac=info)."""
    idx0, idx1, idx2, _ = find_2nn(feats0, feats1)
    idx1_orig = idx1
    feat_dist = None
    if mode in ("MNN", "MMN"):
        f0, f1, f2 = nn_to_mutual(feats0, feats1, idx0, idx1, idx2, force_return_2nd=True)
    elif mode == "GPF":
        f0, f1, f2, _, _, _, feat_dist = Grid_Prioritized_Filter(feats0, feats1, idx0, idx1, idx2, xyz0, args)
    elif mode == "no_filter":
        f0, f1, f2 = idx0, idx1, idx2
    else:
        raise AssertionError("unknown mode")
    src = _f32(xyz0)[f0]; tgt = _f32(xyz1)[f1]
    if prosac:
        # FR.py:73-80 + GC_RANSAC.py:39-43: pairs sorted best match quality first, PROSAC sampler
        if feat_dist is None:
            feat_dist = calc_distance_ratio_in_feature_space(feats0, feats1, f0, f1, f2)
        order = prosac_order(feat_dist)
        T, info = ransac(src[order], tgt[order], iters, sample_size, use_elc, thr, seed, confidence, batch, sampler=1, scoring=scoring,
                         local_opt=local_opt)
    else:
        T, info = ransac(src, tgt, iters, sample_size, use_elc, thr, seed, confidence, batch, sampler=2 if unique else 0, scoring=scoring,
                         local_opt=local_opt)
    n_ref = 0
    if refit_on_orig == 2 and info["best_h"] >= 0:
        # GC codebase: final least squares over the inliers among the filtered pairs
        Tr, n_ref = refit(src, tgt, np.arange(len(src)), T, thr)
        T = Tr
    elif refit_on_orig == 3 and info["best_h"] >= 0:
        T, n_ref = refit(xyz0, xyz1, idx1_orig, T, thr, feats0, feats1)
    elif refit_on_orig and info["best_h"] >= 0:
        T, n_ref = refit(xyz0, xyz1, idx1_orig, T, thr)
    return dict(T=T, idx0=f0, idx1=f1, idx1_orig=idx1_orig, ransac=info, n_refit=n_ref)
