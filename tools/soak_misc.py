"""One-off soak of the stages around RANSAC: GPF lists, ratio, refit, ICP, voxel de-duplication against the oracle with random
sizes and parameters (not part of the suite)."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarregistration_amd import matching, ransac, synth, voxel
from oracle import oracle
from tests.conftest import Args
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t = torch.from_numpy
def H(x): return x.cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
t0 = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(3000 + case)
    N = int(rng.choice([rng.integers(10, 300), rng.integers(300, 4000), rng.integers(4000, 12000)]))
    N1 = max(2, int(N * rng.uniform(0.5, 1.5)))
    p = synth.make_pair(N=N, N1=N1, rho=float(rng.uniform(0.1, 0.9)), s=float(rng.uniform(0.4, 0.95)), seed=case, clustered=bool(rng.integers(2)))
    F0, F1, X0, X1 = p["feats0"], p["feats1"], p["xyz0"], p["xyz1"]
    e0, e1, e2, _ = oracle.find_2nn(F0, F1)
    # ---- GPF (both orders) and the ratio
    a = Args(GPF_factor=float(rng.choice([0.05, 0.3, 0.5, 1.0 / 3.0, 0.7, 1.5, 2.0])), GPF_grid_wid=int(rng.choice([2, 3, 7, 10, 16, 23, 64])))
    for bb in (False, True):
        g = matching.Grid_Prioritized_Filter(t(F0), t(F1), t(e0), t(e1), t(e2), t(X0), a, BB_first=bb)
        o = oracle.Grid_Prioritized_Filter(F0, F1, e0, e1, e2, X0, a, BB_first=bb)
        for x, y in zip(g[:2], o[:2]):
            assert np.array_equal(H(x), H(y)), ("gpf", case, N, N1, vars(a), bb)
    r = matching.calc_distance_ratio_in_feature_space(t(F0), t(F1), t(e0), t(e1), t(e2))
    ro = oracle.calc_distance_ratio_in_feature_space(F0, F1, e0, e1, e2)
    assert np.array_equal(H(r), H(ro)), ("ratio", case)
    # ---- refit and ICP from a perturbed ground truth
    Tn = p["T_gt"].copy(); Tn[:3, 3] += rng.normal(0, 0.15, 3)
    T, n = ransac.refit_dev(X0, X1, e1, Tn); Te, ne = oracle.refit(X0, X1, e1, Tn)
    assert n == ne and np.abs(T - Te).max() <= 1e-9, ("refit", case, n, ne)
    md = float(rng.choice([0.3, 0.6, 1.2])); mi = int(rng.choice([1, 5, 30]))
    Ti, ii = ransac.icp_dev(X0, X1, Tn, max_dist=md, max_iter=mi); Tie, iie = oracle.icp(X0, X1, Tn, max_dist=md, max_iter=mi)
    assert ii["n_corr"] == iie["n_corr"] and ii["iterations"] == iie["iterations"] and np.abs(Ti - Tie).max() <= 1e-9, ("icp", case, ii, iie)
    # ---- voxel de-duplication
    vs = float(rng.choice([0.05, 0.3, 1.0, 5.0]))
    pts = np.concatenate([X0, X0[rng.integers(N, size=N // 3)] + rng.normal(0, 0.01, (N // 3, 3)).astype(np.float32)]).astype(np.float64)
    c, i = voxel.sparse_quantize(pts / vs); co, io = oracle.sparse_quantize(pts / vs)
    assert np.array_equal(H(c), H(co)) and np.array_equal(H(i), H(io)), ("voxel", case, N, vs)
print(f"misc soak ok: {n_cases} cases in {time.time() - t0:.0f} s")
