"""On-disk formats either side of the hot path.

* ``balanced_sets/<set>/<phase>.txt`` pair lists of the reference (header ``session_ind i j mot0..mot15 trans_x trans_y
  trans_z roll pitch yaw overlap overlap_symmetric``; read at Experiments/dataloader/balanced/ApolloSouthbay.py:99-100,
  GT = columns 3..18, :142).
* ``coarse_motions.txt`` per-pair 4x4 outputs (writer FCGF_FAST/test.py:86-106: header, one row per pair, row-major,
  ``%.16f``, sorted by source index then stably by session).
* a feature cache (new; the reference computes FCGF on the fly): ``<dir>/<session>_<idx>.npz`` with ``xyz`` [N,3] float32
  and ``feats`` [N,32] float32, so real data can be registered without MinkowskiEngine.
* the reference's cloud cache: ``<cache_dir>/<session>_<idx>.npy``, the raw scan as an [N,3] float64 array
  (Experiments/dataloader/balanced/ApolloSouthbay.py:148-158; written by ``load_PC(..., cache_file)``).
"""
import os

import numpy as np

DATASET_NAMES = {"A": "ApolloSouthbay", "B": "NuScenes_boston", "S": "NuScenes_singapore", "K": "KITTI_10m", "L": "LyftLEVEL5"}
HEADER = "session_ind source_ind target_ind " + " ".join(f"mot{k}" for k in range(16))


def read_pair_list(path):
    """-> dict(session, src, tgt [P] int64, T_gt [P,4,4] float64, overlap [P] float64 or None)."""
    with open(path) as f:
        header = f.readline().split()
    rows = np.loadtxt(path, skiprows=1, ndmin=2)
    assert header[:3] == ["session_ind", "i", "j"] and header[3] == "mot0", f"unexpected header in {path}"
    overlap = rows[:, header.index("overlap")] if "overlap" in header else None
    return dict(session=rows[:, 0].astype(np.int64), src=rows[:, 1].astype(np.int64), tgt=rows[:, 2].astype(np.int64),
                T_gt=rows[:, 3:19].reshape(-1, 4, 4).copy(), overlap=overlap)


def write_coarse_motions(path, session, src, tgt, T):
    """FCGF_FAST/test.py:86-106 byte format."""
    session, src, tgt = (np.asarray(a, np.int64) for a in (session, src, tgt))
    T = np.asarray(T, np.float64).reshape(-1, 16)
    o1 = np.argsort(src)
    o0 = np.argsort(session[o1], kind="stable")
    order = o1[o0]
    with open(path, "w") as f:
        f.write(HEADER + "\n")
        for i in order:
            f.write("%d %d %d " % (session[i], src[i], tgt[i]) + " ".join("%.16f" % v for v in T[i]) + "\n")


def read_coarse_motions(path):
    rows = np.loadtxt(path, skiprows=1, ndmin=2)
    return rows[:, :3].astype(np.int64), rows[:, 3:19].reshape(-1, 4, 4)


def ref_cloud_path(cache_dir, session, idx):
    return os.path.join(cache_dir, "%d_%d.npy" % (session, idx))             # ApolloSouthbay.py:148,154


def load_ref_cloud(cache_dir, session, idx):
    """One scan of the reference's cloud cache: [N,3] float64 (extra columns, e.g. intensity, are dropped)."""
    path = ref_cloud_path(cache_dir, session, idx)
    if not os.path.isfile(path):
        raise FileNotFoundError(f"{path}: not in the cloud cache (the reference fills it from the raw dataset, which is not part of this repo)")
    a = np.load(path)
    assert a.ndim == 2 and a.shape[1] >= 3, f"{path}: expected an [N,3] array, got {a.shape}"
    return np.ascontiguousarray(a[:, :3], np.float64)


def save_ref_cloud(cache_dir, session, idx, xyz):
    os.makedirs(cache_dir, exist_ok=True)
    np.save(ref_cloud_path(cache_dir, session, idx), np.asarray(xyz, np.float64))


def cache_path(cache_dir, session, idx):
    return os.path.join(cache_dir, "%d_%d.npz" % (session, idx))


def load_cloud(cache_dir, session, idx):
    z = np.load(cache_path(cache_dir, session, idx))
    return np.ascontiguousarray(z["xyz"], np.float32), np.ascontiguousarray(z["feats"], np.float32)


def save_cloud(cache_dir, session, idx, xyz, feats):
    os.makedirs(cache_dir, exist_ok=True)
    np.savez(cache_path(cache_dir, session, idx), xyz=np.asarray(xyz, np.float32), feats=np.asarray(feats, np.float32))
