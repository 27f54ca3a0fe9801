"""One-off NN soak at larger sizes: single calls and 3-pair batched calls (ragged), four descriptor distributions incl. duplicate-heavy
ones (candidate-store overflow -> exact full scan), against the oracle's exhaustive search (not part of the suite)."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarregistration_amd import matching, FR, _ext
from oracle import oracle
from tests.conftest import Args
from tests.test_gpu_fuzz import _features
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t = torch.from_numpy
t0 = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(11000 + case)
    trio = []
    for k in range(3):
        n0, n1 = int(rng.integers(1000, 40000)), int(rng.integers(1000, 40000))
        kind = ("correlated", "duplicates", "scales", "random")[int(rng.integers(4))]
        F0, F1 = _features(rng, n0, n1, kind)
        e0, e1, e2, _ = oracle.find_2nn(F0, F1)
        i0, i1, i2, _ = matching.find_2nn(t(F0), t(F1))
        assert np.array_equal(i1.numpy(), e1) and np.array_equal(i2.numpy(), e2), ("single", case, k, kind, n0, n1)
        m = oracle.nn_to_mutual(F0, F1, e0, e1, e2)
        g = matching.nn_to_mutual(t(F0), t(F1), t(e0), t(e1), t(e2))
        assert all(np.array_equal(a.numpy(), b) for a, b in zip(g, m)), ("mutual", case, k, kind, n0, n1)
        trio.append((F0, F1, e1, e2, m))
    # the three pairs as one batched call (MNN): NN lists and mutual lists per pair
    a = Args(mode="MNN", codebase="open3D", iters=200, ransac_n=3, o3d_conf=1.0)
    params = FR.pair_params(a)
    dev = []
    for F0, F1, *_ in trio:
        x0 = torch.rand(F0.shape[0], 3, device="cuda"); x1 = torch.rand(F1.shape[0], 3, device="cuda")
        dev.append((x0, x1, t(F0).cuda(), t(F1).cuda()))
    nmax = max(max(d[2].shape[0], d[3].shape[0]) for d in dev)
    ws = _ext.Workspace(nmax, nmax, 32, 200, max_pairs=3)
    ws.poison(int(rng.integers(256)))
    out = FR.register_batch_dev(dev, params, ws=ws).cpu().numpy()
    for k, (F0, F1, e1, e2, m) in enumerate(trio):
        n0 = F0.shape[0]
        r = _ext.PairResult.from_buffer_copy(out[k].tobytes())
        bufs = [torch.empty(n0, dtype=torch.int32, device="cuda") for _ in range(4)]
        _ext.check(_ext.lib().lr_workspace_lists_at(ws.handle, k, n0, *[b.data_ptr() for b in bufs], None))
        nn1, nn2, c0, c1 = [b.cpu().numpy() for b in bufs]
        assert np.array_equal(nn1, e1) and np.array_equal(nn2, e2), ("batch nn", case, k)
        assert r.n_corr == len(m[0]) and np.array_equal(c0[:r.n_corr], m[0]) and np.array_equal(c1[:r.n_corr], m[1]), ("batch mutual", case, k)
    print(case, [(x[0].shape[0], x[1].shape[0]) for x in trio], f"{time.time() - t0:.0f}s", flush=True)
print(f"NN soak ok: {n_cases} x 3 pairs in {time.time() - t0:.0f} s")
