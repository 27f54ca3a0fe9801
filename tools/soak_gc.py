"""One-off soak of the GC-RANSAC path (local optimisation, polish, SPRT, PROSAC, early exit) against the oracle: random sizes,
inlier ratios, noise levels and flags (not part of the suite)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarregistration_amd import ransac, synth
from oracle import oracle
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
t0 = time.time()
for case in range(n_cases):
    rng = np.random.default_rng(9000 + case)
    n = int(rng.choice([rng.integers(4, 60), rng.integers(60, 3000), rng.integers(3000, 40000)]))
    inl, noise = float(rng.uniform(0.03, 0.95)), float(rng.choice([0.0, 0.02, 0.1, 0.3]))
    src = np.concatenate([rng.uniform(-80, 80, (n, 2)), rng.uniform(-3, 5, (n, 1))], 1).astype(np.float32)
    T = synth.random_motion(rng)
    tgt = (src.astype(np.float64) @ T[:3, :3].T + T[:3, 3] + rng.normal(0, noise, (n, 3))).astype(np.float32)
    bad = rng.random(n) > inl
    tgt[bad] = np.concatenate([rng.uniform(-80, 80, (bad.sum(), 2)), rng.uniform(-3, 5, (bad.sum(), 1))], 1)
    kw = dict(sample_size=3, seed=int(rng.integers(1 << 30)), sampler=int(rng.choice([1, 2])), scoring=int(rng.choice([0, 1, 2])),
              local_opt=int(rng.choice([1, 1, 2])), confidence=float(rng.choice([1.0, 0.999, 0.99])), batch=int(rng.choice([0, 0, 512, 4096])),
              use_elc=int(rng.choice([0, 1, 1, 2])), lo_rounds=int(rng.choice([0, 0, 1, 3])), lo_trials=int(rng.choice([0, 0, 1, 7])),
              lo_max_calls=int(rng.choice([0, 0, 1, 2])), min_iters=int(rng.choice([0, 0, 600])))
    iters = int(rng.choice([300, 3000, 20000]))
    Tg, info = ransac.ransac_dev(src, tgt, iters, **kw)
    Te, einfo = oracle.ransac(src, tgt, iters, **kw)
    assert info == einfo and np.array_equal(Tg, Te), (case, n, inl, noise, kw, iters, info, einfo)
print(f"GC soak ok: {n_cases} cases in {time.time() - t0:.0f} s")
