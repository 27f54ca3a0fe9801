"""Time of lr_ransac with / without GC-RANSAC's local optimisation and final polish (one pair, planted correspondences with noise)."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarregistration_amd import ransac, _ext
from lidarregistration_amd.matching import workspace, _stream


def planted(m, rho, noise, seed):
    rng = np.random.default_rng(seed)
    src = rng.uniform(-50, 50, (m, 3)).astype(np.float32)
    ang = 0.3
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    tgt = src.astype(np.float64) @ R.T + np.array([2.0, -1.0, 0.3]) + rng.normal(0, noise, (m, 3))
    out = rng.random(m) > rho
    tgt[out] = rng.uniform(-50, 50, (int(out.sum()), 3))
    return torch.from_numpy(src).cuda(), torch.from_numpy(tgt.astype(np.float32)).cuda()


for m in (15717, 30000):
    src, tgt = planted(m, 0.4, 0.05, 3)
    ws = workspace(m, 1, 50000)
    T = torch.empty(16, dtype=torch.float64, device="cuda"); res = torch.zeros(ctypes.sizeof(_ext.RansacResult), dtype=torch.uint8, device="cuda")
    for lo in (0, 2, 1):
        p = ransac.ransac_params(50000, 3, 1, 0.6, 51, 0.999, 0, 1, 0, 1, lo)
        def call():
            _ext.check(_ext.lib().lr_ransac(ws.handle, src.data_ptr(), tgt.data_ptr(), m, None, ctypes.byref(p), T.data_ptr(), res.data_ptr(), _stream()))
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): call()
        e1.record(); torch.cuda.synchronize()
        r = _ext.RansacResult.from_buffer_copy(res.cpu().numpy().tobytes())
        print(f"M={m} local_opt={lo}: {e0.elapsed_time(e1) / 10 * 1e3:8.0f} us per call   best_count {r.best_count} ssq {r.best_ssq} n_ids {r.n_ids}")
