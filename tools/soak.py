"""One-off soak: many random NN / mutual / FR cases against the oracle on a poisoned workspace (not part of the suite)."""
import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarregistration_amd import matching, synth, FR
from oracle import oracle
from tests.conftest import Args
from tests.test_gpu_fuzz import _features
t = torch.from_numpy
ws = matching.workspace(9000, 9000, 20000)
t0 = time.time(); n_cases = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    rng = np.random.default_rng(5000 + seed)
    for kind in ("correlated", "duplicates", "scales", "random"):
        n0, n1 = int(rng.integers(1, 8000)), int(rng.integers(1, 8000))
        F0, F1 = _features(rng, n0, n1, kind)
        e0, e1, e2, _ = oracle.find_2nn(F0, F1)
        m = oracle.nn_to_mutual(F0, F1, e0, e1, e2)
        ws.poison(int(rng.integers(256)))
        i0, i1, i2, _ = matching.find_2nn(t(F0), t(F1))
        assert np.array_equal(i1.numpy(), e1) and (n1 < 2 or np.array_equal(i2.numpy(), e2)), (seed, kind, n0, n1)
        ws.poison(int(rng.integers(256)))
        g = matching.nn_to_mutual(t(F0), t(F1), t(e0), t(e1), t(e2))
        assert all(np.array_equal(a.numpy(), b) for a, b in zip(g, m)), (seed, kind, n0, n1)
        n_cases += 1
    if seed % 6 == 0:
        N = int(rng.integers(1500, 6000)); mode = ("MNN", "GPF", "no_filter")[seed % 3]; cb = ("GC", "open3D")[(seed // 3) % 2]
        p = synth.make_pair(N=N, rho=0.5, s=0.9, seed=seed, clustered=(mode == "GPF"))
        a = Args(mode=mode, codebase=cb, iters=1200, GPF_factor=0.5, prosac=(cb == "GC"))
        ns = 3 if cb == "GC" else 4
        e = oracle.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode=mode, iters=1200, sample_size=ns, seed=51, args=a,
                                 confidence=a.GC_conf if cb == "GC" else a.o3d_conf, refit_on_orig=0 if cb == "GC" else 1,
                                 prosac=(cb == "GC"), scoring=2 if cb == "GC" else 0, local_opt=1 if cb == "GC" else 0)
        ws.poison(int(rng.integers(256)))
        T = FR.FR(t(p["xyz0"]), t(p["xyz1"]), t(p["feats0"]), t(p["feats1"]), a, p["T_gt"])[0]
        np.testing.assert_allclose(T, e["T"], rtol=0, atol=1e-9)
        n_cases += 1
print("soak ok:", n_cases, "cases in %.0f s" % (time.time() - t0))
