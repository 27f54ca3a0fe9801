"""Parity of the HIP path (through the C ABI) against the oracle and the golden vectors.  Needs an MI355X."""
import numpy as np
import pytest

from lidarregistration_amd import synth
from tests.conftest import Args, gc_oracle_kwargs, golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lr():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    from lidarregistration_amd import FR, _ext, matching, ransac
    _ext.lib()                      # fails loudly if liblidarreg.so is missing
    class NS: pass
    ns = NS(); ns.FR = FR; ns.matching = matching; ns.ransac = ransac; ns.torch = torch; ns.ext = _ext
    return ns


def _bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


# ----------------------------------------------------------------------------- NN (a1/a2)
@pytest.mark.parametrize("n0,n1,seed", [(2048, 2048, 1), (3000, 2500, 2), (251, 499, 3), (250, 250, 4), (33, 65, 5),
                                       (1, 40, 6), (5000, 1029, 7), (4100, 9000, 8)])
def test_nn_top2_bit_exact(lr, oracle, n0, n1, seed):
    F0, F1 = synth.make_features(n0, n1, 32, 0.5, 1.0, seed)
    i1, i2, s1, s2 = lr.matching.nn_top2_dev(F0, F1, want_2nd=True, want_dist=True)
    o1, o2, os1, os2 = oracle.nn_top2(F0, F1)
    assert np.array_equal(i1.cpu().numpy(), o1)
    assert np.array_equal(i2.cpu().numpy(), o2)
    assert np.array_equal(_bits(s1.cpu().numpy()), _bits(os1))      # distances bit for bit
    assert np.array_equal(_bits(s2.cpu().numpy()), _bits(os2))


def test_nn_golden_vectors(lr):
    g = golden("g1_find_nn.npz")
    for tag in "abcd":
        n0, n1, d, seed = [int(v) for v in g[f"{tag}_shape"]]
        F0, F1 = synth.make_features(n0, n1, d, 0.5, 1.0, seed)
        c0, c1, c2 = lr.matching.find_nn(lr.torch.from_numpy(F0), lr.torch.from_numpy(F1), return_2nd=True)
        assert c1.dtype == lr.torch.int64 and not c1.is_cuda
        assert np.array_equal(c0.numpy(), np.arange(n0))
        assert np.array_equal(c1.numpy(), g[f"{tag}_idx1"]) and np.array_equal(c2.numpy(), g[f"{tag}_idx2"])
        d0, d1, none = lr.matching.find_nn(lr.torch.from_numpy(F0), lr.torch.from_numpy(F1), return_2nd=False)
        assert none is None and np.array_equal(d1.numpy(), c1.numpy())


def test_nn_exact_ties_and_duplicates(lr, oracle):
    # every candidate row appears three times -> exact distance ties everywhere -> first index must win,
    # and the 2nd/3rd candidates tie after sqrt, which drives every row through the exact fix-up kernel
    F0, F1 = synth.make_features(700, 300, 32, 0.5, 1.0, 11)
    F1 = np.concatenate([F1, F1, F1], 0)
    i1, i2, s1, s2 = lr.matching.nn_top2_dev(F0, F1, want_2nd=True, want_dist=True)
    o1, o2, os1, os2 = oracle.nn_top2(F0, F1)
    assert np.array_equal(i1.cpu().numpy(), o1) and np.array_equal(i2.cpu().numpy(), o2)
    assert (o1 < 300).all() and np.array_equal(o2, o1 + 300)
    # identical clouds: distance clamps at 1e-30 -> sqrt -> 1e-15; self match first
    i1, i2, s1, _ = lr.matching.nn_top2_dev(F0, F0, want_2nd=True, want_dist=True)
    o1, o2, os1, _ = oracle.nn_top2(F0, F0)
    assert np.array_equal(i1.cpu().numpy(), o1) and np.array_equal(i2.cpu().numpy(), o2)
    assert np.array_equal(_bits(s1.cpu().numpy()), _bits(os1))


def test_nn_near_ties_sqrt_rounding(lr, oracle):
    # candidates that differ by one ulp-scale perturbation: d2 differs, sqrt(d2) often does not
    rng = np.random.default_rng(5)
    F0, F1 = synth.make_features(512, 64, 32, 0.5, 1.0, 21)
    reps = []
    for k in range(40):
        P = F1.copy()
        P[:, k % 32] = np.nextafter(P[:, k % 32], np.float32(2.0 * (k % 2) - 1.0))
        reps.append(P)
    F1 = np.concatenate(reps[::-1], 0).astype(np.float32)
    i1, i2, _, _ = lr.matching.nn_top2_dev(F0, F1, want_2nd=True)
    o1, o2, _, _ = oracle.nn_top2(F0, F1)
    assert np.array_equal(i1.cpu().numpy(), o1) and np.array_equal(i2.cpu().numpy(), o2)


def test_nn_full_size_30k(lr, oracle):
    p = synth.make_pair(N=30000, seed=51)
    i1, i2, s1, s2 = lr.matching.nn_top2_dev(p["feats0"], p["feats1"], want_2nd=True, want_dist=True)
    o1, o2, os1, os2 = oracle.nn_top2(p["feats0"], p["feats1"])
    assert np.array_equal(i1.cpu().numpy(), o1) and np.array_equal(i2.cpu().numpy(), o2)
    assert np.array_equal(_bits(s1.cpu().numpy()), _bits(os1))
    # ... and against the REFERENCE itself at this size (tests/golden/make_golden_full.py: matching.py find_2nn / nn_to_mutual /
    # calc_distance_ratio_in_feature_space / Grid_Prioritized_Filter run on this very pair)
    g = golden("g12_full_30k.npz")
    assert np.array_equal(i1.cpu().numpy(), g["idx1"]) and np.array_equal(i2.cpu().numpy(), g["idx2"])
    t = lr.torch.from_numpy
    F0, F1 = t(p["feats0"]), t(p["feats1"])
    i0 = lr.torch.arange(30000)
    m0, m1, m2 = lr.matching.nn_to_mutual(F0, F1, i0, i1.cpu().long(), i2.cpu().long())
    assert len(m0) == 15794
    assert np.array_equal(m0.numpy(), g["mnn_idx0"]) and np.array_equal(m1.numpy(), g["mnn_idx1"]) and np.array_equal(m2.numpy(), g["mnn_idx2"])
    r = lr.matching.calc_distance_ratio_in_feature_space(F0, F1, m0, m1, m2).cpu().numpy()
    np.testing.assert_allclose(r, g["mnn_ratio"], rtol=2e-6, atol=0)
    from tests.test_oracle_golden import prosac_order_consistent
    assert prosac_order_consistent(oracle.prosac_order(r), g["mnn_ratio"], g["mnn_prosac_order"])
    for tag, factor in (("gpf20", 2.0), ("gpf05", 0.5)):
        k = lr.matching.Grid_Prioritized_Filter(F0, F1, i0, i1.cpu().long(), i2.cpu().long(), t(p["xyz0"]), Args(GPF_factor=factor))
        assert np.array_equal(k[0].numpy(), g[f"{tag}_idx0"]) and np.array_equal(k[1].numpy(), g[f"{tag}_idx1"])


# ----------------------------------------------------------------------------- mutual / ratio / GPF (a3-a7)
@pytest.fixture(scope="module")
def filt(oracle):
    g = golden("g2_filters.npz")
    n0, n1, d, seed = [int(v) for v in g["shape"]]
    F0, F1 = synth.make_features(n0, n1, d, 0.5, 1.0, seed)
    xyz0, xyz1, T_gt = synth.make_clouds(n0, n1, 0.5, seed, clustered=True)
    i0, i1, i2, _ = oracle.find_2nn(F0, F1)
    return dict(g=g, F0=F0, F1=F1, xyz0=xyz0, xyz1=xyz1, T_gt=T_gt, i0=i0, i1=i1, i2=i2)


def test_nn_to_mutual_golden_and_arity(lr, filt):
    t = lr.torch.from_numpy
    g = filt["g"]
    m0, m1, m2 = lr.matching.nn_to_mutual(t(filt["F0"]), t(filt["F1"]), t(filt["i0"]), t(filt["i1"]), t(filt["i2"]))
    assert np.array_equal(m0.numpy(), g["mnn_idx0"]) and np.array_equal(m1.numpy(), g["mnn_idx1"])
    assert np.array_equal(m2.numpy(), g["mnn_idx2"])
    assert len(lr.matching.nn_to_mutual(t(filt["F0"]), t(filt["F1"]), t(filt["i0"]), t(filt["i1"]))) == 2
    r = lr.matching.nn_to_mutual(t(filt["F0"]), t(filt["F1"]), t(filt["i0"]), t(filt["i1"]), None, force_return_2nd=True)
    assert len(r) == 3 and r[2] is None
    is_bb, num_bb = lr.matching.mark_best_buddies(t(filt["F0"]), t(filt["F1"]), t(filt["i0"]), t(filt["i1"]))
    assert np.array_equal(is_bb, g["is_bb"]) and int(num_bb) == int(g["num_bb"])


@pytest.mark.parametrize("n0,n1,seed", [(1000, 3000, 31), (3000, 1000, 32), (257, 255, 33),
                                       (8200, 8193, 34), (1023, 8195, 35)])      # (around the 1024 x 8 elements of a rank-kernel round)
def test_nn_to_mutual_ragged_vs_oracle(lr, oracle, n0, n1, seed):
    F0, F1 = synth.make_features(n0, n1, 32, 0.5, 0.8, seed)
    i0, i1, i2, _ = oracle.find_2nn(F0, F1)
    e0, e1, e2 = oracle.nn_to_mutual(F0, F1, i0, i1, i2)
    t = lr.torch.from_numpy
    m0, m1, m2 = lr.matching.nn_to_mutual(t(F0), t(F1), t(i0), t(i1), t(i2))
    assert np.array_equal(m0.numpy(), e0) and np.array_equal(m1.numpy(), e1) and np.array_equal(m2.numpy(), e2)


def _mutual_vs_oracle(lr, oracle, F0, F1):
    i0, i1, i2, _ = oracle.find_2nn(F0, F1)
    e0, e1, e2 = oracle.nn_to_mutual(F0, F1, i0, i1, i2)
    t = lr.torch.from_numpy
    m0, m1, m2 = lr.matching.nn_to_mutual(t(F0), t(F1), t(i0), t(i1), t(i2))
    assert np.array_equal(m0.numpy(), e0) and np.array_equal(m1.numpy(), e1) and np.array_equal(m2.numpy(), e2)
    return len(e0)


def test_nn_to_mutual_reverse_ordering_edge_cases(lr, oracle):
    """The reverse pass orders rows / columns by forward NN distance and prunes column tiles: inputs that stress it."""
    rng = np.random.default_rng(77)
    # (a) every query points at the same target: one row in the reverse pass, 1999 cloud-1 points left out
    F1 = rng.standard_normal((2000, 32)).astype(np.float32); F1 /= np.linalg.norm(F1, axis=1, keepdims=True)
    F0 = np.tile(F1[123], (700, 1)) + 1e-3 * rng.standard_normal((700, 32)).astype(np.float32)
    assert _mutual_vs_oracle(lr, oracle, F0.astype(np.float32), F1) == 1
    # (b) all forward distances identical (a permuted copy): the distance range is a single point, one sort bucket
    F0 = rng.standard_normal((1500, 32)).astype(np.float32)
    perm = rng.permutation(1500)
    assert _mutual_vs_oracle(lr, oracle, F0, F0[perm].copy()) == 1500
    # (c) exact duplicates on both sides: ties everywhere, resolved by index order in both directions
    base = rng.standard_normal((300, 32)).astype(np.float32)
    F0 = np.concatenate([base, base, base[:100]]); F1 = np.concatenate([base[::-1], base[:50]])
    _mutual_vs_oracle(lr, oracle, F0, F1)
    # (d) widely spread distances (clusters at very different scales) and a far outlier
    F0 = np.concatenate([1e-3 * rng.standard_normal((400, 32)), rng.standard_normal((400, 32)), 40.0 * rng.standard_normal((400, 32))]).astype(np.float32)
    F1 = (F0[rng.permutation(1200)[:900]] * (1 + 0.05 * rng.standard_normal((900, 1)))).astype(np.float32)
    F1[0] = 3000.0
    _mutual_vs_oracle(lr, oracle, F0, F1)
    # (e) more rows than one block on the reverse side, few columns
    F0, F1 = synth.make_features(40, 2600, 32, 0.5, 0.8, 91)
    _mutual_vs_oracle(lr, oracle, F0, F1)
    # (f) one sort bucket holding nearly a whole cloud on both sides -- far more than the ordering kernel's copy queue takes in a round (round 6:
    #     nn16_rev_order_kernel; elements past the queue are moved by their own thread) -- next to ordinary points: 9 000 copies of one
    #     descriptor in cloud 0, 7 000 of another in cloud 1
    a = rng.standard_normal((1, 32)).astype(np.float32); b = rng.standard_normal((1, 32)).astype(np.float32)
    F0 = np.concatenate([np.repeat(a, 9000, axis=0), rng.standard_normal((800, 32)).astype(np.float32)])
    F1 = np.concatenate([rng.standard_normal((700, 32)).astype(np.float32), np.repeat(b, 7000, axis=0), a + 1e-3])
    _mutual_vs_oracle(lr, oracle, F0, F1)


def test_ratio_bit_exact(lr, oracle, filt):
    r = lr.matching.calc_distance_ratio_in_feature_space(filt["F0"], filt["F1"], filt["i0"], filt["i1"], filt["i2"])
    e = oracle.calc_distance_ratio_in_feature_space(filt["F0"], filt["F1"], filt["i0"], filt["i1"], filt["i2"])
    assert np.array_equal(_bits(r.cpu().numpy()), _bits(e))
    np.testing.assert_allclose(r.cpu().numpy(), filt["g"]["ratio_nn"], rtol=2e-6)


@pytest.mark.parametrize("m", [1, 7, 31, 33, 255, 256, 257, 1000, 5003])
@pytest.mark.parametrize("dim", [32, 16, 5])
def test_ratio_ragged_counts_bit_exact(lr, oracle, m, dim):
    """Every count around the kernel's 32-pair rounds and 256-pair blocks (dim 32: eight lanes per pair, running sums handed from lane
    to lane), repeated and scattered rows, and the generic widths: the bits of matching.py:77's ratio as the oracle computes it."""
    rng = np.random.default_rng(1000 * dim + m)
    F0 = rng.standard_normal((3000, dim)).astype(np.float32); F1 = rng.standard_normal((2500, dim)).astype(np.float32)
    i0 = rng.integers(0, 3000, m).astype(np.int64); i1 = rng.integers(0, 2500, m).astype(np.int64); i2 = rng.integers(0, 2500, m).astype(np.int64)
    if m > 3:
        i2[1] = i1[1]; F1[i1[2]] = F0[i0[2]]          # ratio exactly 1, and a zero numerator
    r = lr.matching.calc_distance_ratio_in_feature_space(F0, F1, i0, i1, i2)
    e = oracle.calc_distance_ratio_in_feature_space(F0, F1, i0, i1, i2)
    assert np.array_equal(_bits(r.cpu().numpy()), _bits(e))


@pytest.mark.parametrize("k", [0, 1, 2, 3, 4, 5, 6])
def test_gpf_golden(lr, oracle, filt, k):
    g = filt["g"]
    factor, wid = g[f"gpf{k}_cfg"]
    a = Args(GPF_grid_wid=int(wid), GPF_factor=float(factor))
    t = lr.torch.from_numpy
    out = lr.matching.Grid_Prioritized_Filter(t(filt["F0"]), t(filt["F1"]), t(filt["i0"]), t(filt["i1"]), t(filt["i2"]),
                                              t(filt["xyz0"]), a)
    assert np.array_equal(out[0].numpy(), g[f"gpf{k}_idx0"])
    assert np.array_equal(out[1].numpy(), g[f"gpf{k}_idx1"])
    assert np.array_equal(out[2].numpy(), g[f"gpf{k}_idx2"])
    e = oracle.Grid_Prioritized_Filter(filt["F0"], filt["F1"], filt["i0"], filt["i1"], filt["i2"], filt["xyz0"], a)
    assert np.array_equal(_bits(out[6].cpu().numpy()), _bits(e[6]))


@pytest.mark.parametrize("k", [0, 1, 2])
def test_gpf_bb_first_golden(lr, filt, k):
    g = filt["g"]
    a = Args(GPF_max_matches=int(g[f"gpfbb{k}_cap"]))
    t = lr.torch.from_numpy
    out = lr.matching.Grid_Prioritized_Filter(t(filt["F0"]), t(filt["F1"]), t(filt["i0"]), t(filt["i1"]), t(filt["i2"]),
                                              t(filt["xyz0"]), a, BB_first=True)
    assert np.array_equal(out[0].numpy(), g[f"gpfbb{k}_idx0"]) and np.array_equal(out[1].numpy(), g[f"gpfbb{k}_idx1"])
    assert (out[6] is not None) == bool(g[f"gpfbb{k}_has_score"])
    if out[6] is not None:
        np.testing.assert_allclose(out[6].cpu().numpy(), g[f"gpfbb{k}_score"], rtol=0, atol=3e-6)


@pytest.mark.parametrize("factor,wid,seed", [(0.3, 10, 41), (0.05, 3, 42), (1.5, 16, 43), (0.37, 64, 44), (1.0 / 3.0, 23, 45), (0.7, 2, 46)])
def test_gpf_vs_oracle(lr, oracle, factor, wid, seed):
    n0, n1 = 4000, 3500
    F0, F1 = synth.make_features(n0, n1, 32, 0.5, 1.0, seed)
    xyz0, _, _ = synth.make_clouds(n0, n1, 0.5, seed, clustered=True)
    i0, i1, i2, _ = oracle.find_2nn(F0, F1)
    a = Args(GPF_grid_wid=wid, GPF_factor=factor)
    e = oracle.Grid_Prioritized_Filter(F0, F1, i0, i1, i2, xyz0, a)
    t = lr.torch.from_numpy
    out = lr.matching.Grid_Prioritized_Filter(t(F0), t(F1), t(i0), t(i1), t(i2), t(xyz0), a)
    assert np.array_equal(out[0].numpy(), e[0]) and np.array_equal(out[1].numpy(), e[1]) and np.array_equal(out[2].numpy(), e[2])
    assert len(e[0]) < n0


# ----------------------------------------------------------------------------- Kabsch / RANSAC / refit
@pytest.mark.parametrize("k", range(6))
def test_kabsch_golden(lr, oracle, k):
    g = golden("g7_kabsch.npz")
    P, Q, w = g[f"k{k}_P"], g[f"k{k}_Q"], g[f"k{k}_w"]
    T = lr.ransac.kabsch_dev(P, Q, w if len(w) else None)
    np.testing.assert_allclose(T, oracle.kabsch(P, Q, w if len(w) else None), rtol=0, atol=1e-12)
    np.testing.assert_allclose(T, g[f"k{k}_T_procrustes"], rtol=0, atol=2e-5)


def _planted(n=4000, inlier=0.3, seed=3):
    rng = np.random.default_rng(seed)
    src = np.concatenate([rng.uniform(-80, 80, (n, 2)), rng.uniform(-3, 5, (n, 1))], 1).astype(np.float32)
    T = synth.random_motion(rng)
    tgt = (src.astype(np.float64) @ T[:3, :3].T + T[:3, 3] + rng.normal(0, 0.05, (n, 3))).astype(np.float32)
    bad = rng.random(n) > inlier
    tgt[bad] = np.concatenate([rng.uniform(-80, 80, (bad.sum(), 2)), rng.uniform(-3, 5, (bad.sum(), 1))], 1)
    return src, tgt, T


@pytest.mark.parametrize("ns,elc,n,iters,seed", [(3, True, 4000, 5000, 51), (4, True, 3000, 5000, 7), (3, False, 1500, 700, 9),
                                                 (4, False, 900, 300, 1), (3, True, 17000, 20000, 123)])
def test_ransac_same_winner_as_oracle(lr, oracle, ns, elc, n, iters, seed):
    src, tgt, T_gt = _planted(n=n, seed=seed)
    T, info = lr.ransac.ransac_dev(src, tgt, iters, sample_size=ns, use_elc=elc, seed=seed)
    Te, einfo = oracle.ransac(src, tgt, iters, sample_size=ns, use_elc=elc, seed=seed)
    assert info == einfo                                   # same hypothesis id, inlier count, error sum, #valid
    assert np.array_equal(T, Te)                           # fp64 minimal-sample model, bit for bit
    assert oracle.rotation_error_deg(T, T_gt) < 1.0


def test_ransac_degenerate_inputs(lr, oracle):
    rng = np.random.default_rng(0)
    src = rng.uniform(-80, 80, (300, 3)).astype(np.float32)
    tgt = (rng.uniform(1e3, 2e3, (300, 3)) * np.array([1, 50, 1000])).astype(np.float32)
    T, info = lr.ransac.ransac_dev(src, tgt, 200, thr=1e-4)
    Te, einfo = oracle.ransac(src, tgt, 200, thr=1e-4)
    assert info == einfo and np.array_equal(T, Te)
    # all correspondences identical: every sample is degenerate, Kabsch must still return finite numbers
    src1 = np.tile(src[:1], (50, 1)); tgt1 = np.tile(tgt[:1], (50, 1))
    T, info = lr.ransac.ransac_dev(src1, tgt1, 64)
    Te, einfo = oracle.ransac(src1, tgt1, 64)
    assert info == einfo and np.array_equal(np.isfinite(T), np.isfinite(Te))


@pytest.mark.parametrize("n,n_in,iters,elc", [(40, 12, 300, False), (200, 150, 3000, False), (64, 20, 2000, True), (500, 3, 4000, False), (500, 3, 30000, False)])
def test_ransac_equal_counts_are_decided_by_the_error_sum(lr, oracle, n, n_in, iters, elc):
    """Noise-free inliers: every all-inlier sample reaches the same count, so the winner is picked by the fixed-point error sum
    (then the hypothesis id) -- a handful of models at the top count (the per-model path of ransac_tie_kernel), hundreds of
    them, and no consensus at all (thousands of models at a count of 1 or 2: its chunked path)."""
    rng = np.random.default_rng(n + iters)
    src = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
    ang = 0.4
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]])
    tgt = (src.astype(np.float64) @ R.T + np.array([3.0, -2.0, 0.5])).astype(np.float32)
    tgt[n_in:] = rng.uniform(-40, 40, (n - n_in, 3)).astype(np.float32)
    perm = rng.permutation(n)
    src, tgt = src[perm], tgt[perm]
    T, info = lr.ransac.ransac_dev(src, tgt, iters, sample_size=3, use_elc=elc, seed=5)
    Te, einfo = oracle.ransac(src, tgt, iters, sample_size=3, use_elc=elc, seed=5)
    assert info == einfo and np.array_equal(T, Te)
    assert info["best_count"] >= (n_in if n_in > 3 else 1)


@pytest.mark.parametrize("n,iters", [(33817, 300), (33817, 64), (12345, 130), (5001, 64), (129, 64), (257, 1000)])
def test_ransac_odd_counts_with_few_hypotheses(lr, oracle, n, iters):
    """Few hypotheses over many correspondences: the scoring kernel splits the correspondences into more chunks than there are, and
    with an odd count the last correspondence must be counted by the one chunk that owns it (found by tools/soak_gc.py: the chunks
    past the end each counted it again)."""
    src, tgt, T_gt = _planted(n=n, seed=n)
    tgt[-1] = (src[-1].astype(np.float64) @ T_gt[:3, :3].T + T_gt[:3, 3]).astype(np.float32)      # the last correspondence is an inlier
    T, info = lr.ransac.ransac_dev(src, tgt, iters, sample_size=3, use_elc=False, seed=7)
    Te, einfo = oracle.ransac(src, tgt, iters, sample_size=3, use_elc=False, seed=7)
    assert info == einfo and np.array_equal(T, Te)


@pytest.mark.parametrize("case", ["far_from_origin", "few_good_models", "nonfinite", "msac_4pt", "noise_free_ties", "just_above_limits",
                                  "odd_tail", "huge_noise", "two_motions"])
def test_ransac_pilot_ordered_scoring_is_exact(lr, oracle, case):
    """The scoring passes skip records a model cannot reach (csrc/lr_ransac.hip, "score": pilot model, residual buckets, reach of
    every model); the oracle scores every model over every correspondence.  Winner id, inlier count, error sum and model must agree
    where the pruning bound is under stress: coordinates of 1e4 m (fp32 rounding against the slack), a pilot that is a bad model,
    non-finite coordinates, MSAC, ties decided by the error sum, sizes at the switch-on limits, an odd sorted list, inlier noise of
    the size of the threshold, and two planted motions of similar support."""
    kw = dict(sample_size=3, use_elc=True, seed=11)
    n, iters, inlier, noise = 6000, 6000, 0.3, 0.05
    if case == "few_good_models": n, iters, inlier = 9000, 30000, 0.06; kw.update(use_elc=False)
    if case == "msac_4pt": kw.update(sample_size=4, scoring=1, use_elc=False); iters = 3000
    if case == "just_above_limits": n, iters = 2048, 2200; kw.update(use_elc=False)
    if case == "odd_tail": n, iters = 2305, 1000; kw.update(use_elc=False)
    if case == "huge_noise": noise = 0.4
    rng = np.random.default_rng(len(case))
    src = np.concatenate([rng.uniform(-80, 80, (n, 2)), rng.uniform(-3, 5, (n, 1))], 1)
    T_gt = synth.random_motion(rng)
    tgt = src @ T_gt[:3, :3].T + T_gt[:3, 3] + (0.0 if case == "noise_free_ties" else rng.normal(0, noise, (n, 3)))
    bad = rng.random(n) > inlier
    tgt[bad] = np.concatenate([rng.uniform(-80, 80, (bad.sum(), 2)), rng.uniform(-3, 5, (bad.sum(), 1))], 1)
    if case == "two_motions":          # a second consistent motion with nearly as many inliers
        T2 = synth.random_motion(rng)
        second = bad & (rng.random(n) < 0.4)
        tgt[second] = src[second] @ T2[:3, :3].T + T2[:3, 3] + rng.normal(0, noise, (second.sum(), 3))
    if case == "far_from_origin":
        src = src + np.array([9000.0, -7000.0, 300.0]); tgt = tgt + np.array([-8000.0, 9500.0, -200.0])
    src, tgt = src.astype(np.float32), tgt.astype(np.float32)
    if case == "nonfinite":
        src[5] = np.nan; tgt[700, 1] = np.inf; src[3000, 2] = -np.inf; tgt[4000] = np.nan
    T, info = lr.ransac.ransac_dev(src, tgt, iters, **kw)
    Te, einfo = oracle.ransac(src, tgt, iters, **kw)
    assert info == einfo, (case, info, einfo)
    assert np.array_equal(T, Te)
    assert info["n_valid"] >= 128 and info["best_count"] > 3          # the pruned path really ran (its switch-on limits) and found something


def test_refit_vs_oracle(lr, oracle):
    p = synth.make_pair(N=6000, rho=0.5, s=0.8, seed=17)
    i0, i1, i2, _ = oracle.find_2nn(p["feats0"], p["feats1"])
    Tn = p["T_gt"].copy(); Tn[:3, 3] += [0.2, -0.1, 0.05]
    T, n = lr.ransac.refit_dev(p["xyz0"], p["xyz1"], i1, Tn)
    Te, ne = oracle.refit(p["xyz0"], p["xyz1"], i1, Tn)
    assert n == ne and n > 500
    np.testing.assert_allclose(T, Te, rtol=0, atol=1e-10)


# ----------------------------------------------------------------------------- FR end to end (a9)
@pytest.mark.parametrize("mode,codebase,N,iters", [("MNN", "open3D", 5000, 1000), ("MMN", "GC", 5000, 1000),
                                                   ("GPF", "GC", 4000, 2000), ("no_filter", "open3D", 3000, 3000)])
def test_FR_matches_oracle_pipeline(lr, oracle, mode, codebase, N, iters):
    p = synth.make_pair(N=N, rho=0.5, s=0.9, seed=51, clustered=(mode == "GPF"))
    a = Args(mode=mode, codebase=codebase, iters=iters, GPF_factor=0.5)
    t = lr.torch.from_numpy
    T, elapsed, pcd0, pcd1, n_init, ir_init, n_filt, ir_filt = lr.FR.FR(t(p["xyz0"]), t(p["xyz1"]), t(p["feats0"]), t(p["feats1"]), a, p["T_gt"])
    kw = gc_oracle_kwargs(a) if codebase == "GC" else dict(sample_size=4, use_elc=True, confidence=a.o3d_conf, refit_on_orig=1, scoring=0)
    e = oracle.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode=mode, iters=iters, seed=51, args=a, **kw)
    assert n_init == N and n_filt == len(e["idx0"])
    # contract: <= 1e-4 rad rotation, <= 1e-3 m translation on identical correspondence inputs
    assert np.radians(oracle.rotation_error_deg(T, e["T"])) <= 1e-4
    assert oracle.translation_error_cm(T, e["T"]) / 100 <= 1e-3
    np.testing.assert_allclose(T, e["T"], rtol=0, atol=1e-9)
    assert T.dtype == np.float64 and T.shape == (4, 4) and elapsed > 0
    assert pcd0.points.shape == (N, 3) and pcd0.points.dtype == np.float64
    assert ir_init == oracle.measure_inlier_ratio(np.arange(N), e["idx1_orig"], p["xyz0"], p["xyz1"], p["T_gt"], 0.3)
    assert ir_filt == oracle.measure_inlier_ratio(e["idx0"], e["idx1"], p["xyz0"], p["xyz1"], p["T_gt"], 0.3)
    assert oracle.rotation_error_deg(T, p["T_gt"]) < 1.0 and oracle.translation_error_cm(T, p["T_gt"]) < 30


def test_FR_full_size_config2(lr, oracle):
    """BASELINE config #2: 30k-pt pair, MNN, 50k iterations -- whole pipeline against the oracle."""
    p = synth.make_pair(N=30000, seed=51)
    a = Args(mode="MNN", codebase="open3D", iters=50000, ransac_n=3, o3d_conf=1.0)      # all 50k hypotheses, as bench.py
    t = lr.torch.from_numpy
    T, elapsed, *_rest = lr.FR.FR(t(p["xyz0"]), t(p["xyz1"]), t(p["feats0"]), t(p["feats1"]), a, p["T_gt"])
    e = oracle.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode="MNN", iters=50000, sample_size=3, seed=51)
    assert _rest[4] == len(e["idx0"])
    assert np.radians(oracle.rotation_error_deg(T, e["T"])) <= 1e-4 and oracle.translation_error_cm(T, e["T"]) / 100 <= 1e-3
    assert oracle.rotation_error_deg(T, p["T_gt"]) < 0.5 and oracle.translation_error_cm(T, p["T_gt"]) < 20


# ----------------------------------------------------------------------------- PROSAC (row f3)
@pytest.mark.parametrize("ns,n,iters,growth,seed", [(3, 4000, 3000, 0, 51), (4, 2500, 2000, 0, 7), (3, 900, 5000, 2000, 9), (3, 5, 64, 0, 1)])
def test_prosac_sampler_same_winner_as_oracle(lr, oracle, ns, n, iters, growth, seed):
    src, tgt, T_gt = _planted(n=n, seed=seed)
    T, info = lr.ransac.ransac_dev(src, tgt, iters, sample_size=ns, seed=seed, sampler=1, prosac_growth=growth)
    Te, einfo = oracle.ransac(src, tgt, iters, sample_size=ns, seed=seed, sampler=1, prosac_growth=growth)
    assert info == einfo and np.array_equal(T, Te)
    # a different run than the uniform sampler's
    _, uinfo = lr.ransac.ransac_dev(src, tgt, iters, sample_size=ns, seed=seed)
    assert n < 10 or uinfo["n_valid"] != info["n_valid"] or uinfo["best_h"] != info["best_h"]


@pytest.mark.parametrize("mode,N,iters", [("MNN", 5000, 1500), ("GPF", 4000, 2000), ("no_filter", 2500, 2500)])
def test_FR_prosac_matches_oracle_pipeline(lr, oracle, mode, N, iters):
    """--codebase GC --prosac True (the reference's default, test.py:308): device-side quality order + PROSAC sampler."""
    p = synth.make_pair(N=N, rho=0.5, s=0.9, seed=52, clustered=(mode == "GPF"))
    a = Args(mode=mode, codebase="GC", iters=iters, GPF_factor=0.5, prosac=True)
    t = lr.torch.from_numpy
    T, _, _, _, n_init, _, n_filt, _ = lr.FR.FR(t(p["xyz0"]), t(p["xyz1"]), t(p["feats0"]), t(p["feats1"]), a, p["T_gt"])
    e = oracle.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode=mode, iters=iters, seed=51, args=a, **gc_oracle_kwargs(a))
    assert n_filt == len(e["idx0"])
    assert np.radians(oracle.rotation_error_deg(T, e["T"])) <= 1e-4 and oracle.translation_error_cm(T, e["T"]) / 100 <= 1e-3
    np.testing.assert_allclose(T, e["T"], rtol=0, atol=1e-9)
    assert oracle.rotation_error_deg(T, p["T_gt"]) < 1.0 and oracle.translation_error_cm(T, p["T_gt"]) < 30
    # the host-level mirror of GC_RANSAC.py sorts by -match_quality itself and must agree with the oracle too
    fd = oracle.calc_distance_ratio_in_feature_space(p["feats0"], p["feats1"], e["idx0"], e["idx1"],
                                                     oracle.find_2nn(p["feats0"], p["feats1"])[2][e["idx0"]]) if mode != "GPF" else None
    if fd is not None:
        A = p["xyz0"][e["idx0"]].astype(np.float32); B = p["xyz1"][e["idx1"]].astype(np.float32)
        Tg, _ = lr.ransac.GC_RANSAC(A, B, 0.6, iters, a, -fd)
        order = oracle.prosac_order(fd)
        # ... and with FR(codebase="GC"): one estimator behind both names (LO + final least squares included)
        Te, _ = oracle.ransac(A[order], B[order], iters, 3, True, 0.6, 51, a.GC_conf, 0, sampler=1, scoring=2, local_opt=1)
        assert np.array_equal(Tg, Te)
        np.testing.assert_allclose(Tg, T, rtol=0, atol=1e-9)


def test_errors_are_loud(lr):
    with pytest.raises(lr.ext.LidarRegError):
        lr.matching.nn_top2_dev(np.zeros((10, 33), np.float32), np.zeros((10, 33), np.float32))      # wider than 32: refused (1..32 are served, tests/test_gpu_dims.py)


def test_workspace_used_on_another_device_is_refused_not_faulted(lr):
    """A workspace is memory of ONE device (include/lidarreg.h conventions): with another device current every entry point that takes it
    returns LR_EINVAL before it launches anything.  The box has one GPU: the library's test hook stands in for hipSetDevice(3)."""
    t = lr.torch
    L = lr.ext.lib()
    dev = t.device("cuda", 0)
    ws = lr.ext.Workspace(1000, 1000, 32, 100, max_pairs=2)
    F = t.randn(500, 32, device=dev); x = t.randn(500, 3, device=dev)
    i1 = t.empty(500, dtype=t.int32, device=dev)
    out = t.zeros((2, 496), dtype=t.uint8, device=dev)
    params = lr.FR.pair_params(Args(iters=100))
    assert L.lr_nn_top2(ws.handle, F.data_ptr(), 500, F.data_ptr(), 500, 32, i1.data_ptr(), None, None, None, None) == 0
    L.lr_debug_fake_current_device(3)
    try:
        assert L.lr_nn_top2(ws.handle, F.data_ptr(), 500, F.data_ptr(), 500, 32, i1.data_ptr(), None, None, None, None) == -1
        assert b"created on device 0 but device 3 is current" in L.lr_last_error()
        assert L.lr_register_pair(ws.handle, x.data_ptr(), x.data_ptr(), F.data_ptr(), F.data_ptr(), 500, 500, 32, params, out.data_ptr(), None) == -1
        with pytest.raises(lr.ext.LidarRegError, match="device 3 is current"):
            lr.FR.register_batch_dev([(x, x, F, F)] * 2, params, out=out, ws=ws)
        assert L.lr_workspace_poison(ws.handle, 1, None) == -1 and L.lr_workspace_timing(ws.handle, 1) == -1
        with pytest.raises(lr.ext.LidarRegError):
            ws.clock()
    finally:
        L.lr_debug_fake_current_device(-1)
    t.cuda.synchronize()
    assert L.lr_nn_top2(ws.handle, F.data_ptr(), 500, F.data_ptr(), 500, 32, i1.data_ptr(), None, None, None, None) == 0
    t.cuda.synchronize()
    assert (i1.cpu().numpy() == np.arange(500)).all()
    # a stream of the right device passes, the null stream too
    s = t.cuda.Stream(device=dev)
    assert L.lr_nn_top2(ws.handle, F.data_ptr(), 500, F.data_ptr(), 500, 32, i1.data_ptr(), None, None, None, s.cuda_stream) == 0
    t.cuda.synchronize()
    ws.close()


# ----------------------------------------------------------------------------- f16 filter robustness (error bound, fallbacks)
@pytest.mark.parametrize("scale0,scale1", [(1.0, 1.0), (37.0, 37.0), (1e-3, 1e-3), (3.0, 0.2), (300.0, 300.0), (1e-6, 1e-6)])
def test_nn_scaled_features_still_bit_exact(lr, oracle, scale0, scale1):
    # non-unit-norm descriptors: the filter's error bound scales with the norms; 300 overflows f16 dot products only
    # mildly, 1e-6 underflows f16 entirely -> every row goes through the exact fallback; results must not change
    F0, F1 = synth.make_features(1500, 1300, 32, 0.5, 1.0, 61)
    F0 = (F0 * np.float32(scale0)).astype(np.float32); F1 = (F1 * np.float32(scale1)).astype(np.float32)
    i1, i2, s1, s2 = lr.matching.nn_top2_dev(F0, F1, want_2nd=True, want_dist=True)
    o1, o2, os1, os2 = oracle.nn_top2(F0, F1)
    assert np.array_equal(i1.cpu().numpy(), o1) and np.array_equal(i2.cpu().numpy(), o2)
    assert np.array_equal(_bits(s1.cpu().numpy()), _bits(os1)) and np.array_equal(_bits(s2.cpu().numpy()), _bits(os2))


def test_nn_f16_overflow_and_mixed_norms(lr, oracle):
    rng = np.random.default_rng(3)
    F0, F1 = synth.make_features(900, 1100, 32, 0.5, 1.0, 62)
    F0 = (F0 * rng.uniform(0.05, 20.0, (900, 1))).astype(np.float32)          # per-row norms over 2.5 decades
    F1 = (F1 * rng.uniform(0.05, 20.0, (1100, 1))).astype(np.float32)
    F1[::97] *= np.float32(1e5)                                                # some rows beyond the f16 range (inf in H)
    i1, i2, _, _ = lr.matching.nn_top2_dev(F0, F1, want_2nd=True)
    o1, o2, _, _ = oracle.nn_top2(F0, F1)
    assert np.array_equal(i1.cpu().numpy(), o1) and np.array_equal(i2.cpu().numpy(), o2)


def test_nn_lattice_many_near_equal_distances(lr, oracle):
    # descriptors on a coarse lattice: huge numbers of equal and nearly equal distances -> candidate lists overflow for
    # many rows, which must route them through the exact row kernel without changing a single index
    rng = np.random.default_rng(9)
    F0 = rng.integers(-1, 2, (800, 32)).astype(np.float32) * 0.25
    F1 = rng.integers(-1, 2, (2000, 32)).astype(np.float32) * 0.25
    F1[:500] = F1[500:1000]                                                    # exact duplicates
    i1, i2, s1, _ = lr.matching.nn_top2_dev(F0, F1, want_2nd=True, want_dist=True)
    o1, o2, os1, _ = oracle.nn_top2(F0, F1)
    assert np.array_equal(i1.cpu().numpy(), o1) and np.array_equal(i2.cpu().numpy(), o2)
    assert np.array_equal(_bits(s1.cpu().numpy()), _bits(os1))


@pytest.mark.parametrize("n0,n1", [(1, 1), (2, 1), (1, 2), (3, 3), (31, 33), (64, 2), (257, 31)])
def test_nn_tiny_clouds(lr, oracle, n0, n1):
    F0, F1 = synth.make_features(n0, n1, 32, 0.5, 1.0, 70 + n0 + n1)
    i1, i2, s1, _ = lr.matching.nn_top2_dev(F0, F1, want_2nd=True, want_dist=True)
    o1, o2, os1, _ = oracle.nn_top2(F0, F1)
    assert np.array_equal(i1.cpu().numpy(), o1)
    if n1 >= 2:
        assert np.array_equal(i2.cpu().numpy(), o2)
    assert np.array_equal(_bits(s1.cpu().numpy()), _bits(os1))


@pytest.mark.parametrize("n0,n1,stride", [(6840, 7499, 0), (3000, 2017, 1), (5000, 4097, 2), (2600, 3231, 0), (900, 33, 0), (700, 95, 1)])
def test_nn_best_neighbour_in_the_partial_last_tile(lr, oracle, n0, n1, stride):
    """Rows whose nearest neighbour is the LAST column of a cloud whose size is not a multiple of the 32-column tile: the staged image
    of that tile repeats the last column in its padding, and phase 1 of the filter pass must not take it for two different columns
    (found by tools/soak_fr.py in round 3: one wrong second neighbour per such row)."""
    from lidarregistration_amd import _ext, matching
    F0, F1 = synth.make_features(n0, n1, 32, 0.5, 1.0, n0 + n1)
    rng = np.random.default_rng(n1)
    rows = rng.choice(n0, 40, replace=False)
    F0[rows] = F1[-1] + rng.normal(0, 0.02, (40, 32)).astype(np.float32)
    F0[rows] /= np.linalg.norm(F0[rows], axis=1, keepdims=True)
    o1, o2, os1, os2 = oracle.nn_top2(F0, F1)
    assert (o1[rows] == n1 - 1).all()
    _ext.DEFAULT_OPTIONS.clear()
    if stride:
        _ext.DEFAULT_OPTIONS["nn_sample_stride"] = stride
    for ws in matching._WS.values():
        ws.close()
    matching._WS.clear()
    try:
        t = lr.torch.from_numpy
        i1, i2, s1, s2 = lr.matching.nn_top2_dev(F0, F1, want_dist=True)
    finally:
        _ext.DEFAULT_OPTIONS.clear()
        for ws in matching._WS.values():
            ws.close()
        matching._WS.clear()
    assert np.array_equal(i1.cpu().numpy(), o1) and np.array_equal(i2.cpu().numpy(), o2)
    assert np.array_equal(_bits(s1.cpu().numpy()), _bits(os1)) and np.array_equal(_bits(s2.cpu().numpy()), _bits(os2))


@pytest.mark.parametrize("mode", ["MNN", "no_filter", "GPF"])
def test_second_neighbour_auto_mode_gives_the_same_pair_result(lr, oracle, mode):
    """The workspace option nn_second_auto drops the second neighbour where no stage reads it; the result block must not change."""
    from lidarregistration_amd import _ext
    p = synth.make_pair(N=4000, rho=0.5, s=0.9, seed=71, clustered=(mode == "GPF"))
    a = Args(mode=mode, codebase="open3D", iters=1500, GPF_factor=0.5)
    params = lr.FR.pair_params(a)
    t = lr.torch.from_numpy
    dev = [t(p[k]).cuda() for k in ("xyz0", "xyz1", "feats0", "feats1")]
    blocks = []
    for auto in (0, 1):
        ws = _ext.Workspace(4000, 4000, 32, 1500)
        ws.set_option("nn_second_auto", auto)
        ws.poison(0xA5)
        out = lr.FR.register_pair_dev(*dev, params, ws=ws)
        r = lr.FR.read_result(out)
        blocks.append((np.array(r.T[:]), r.n_corr, r.ransac.best_h, r.ransac.best_count, r.n_refit))
        ws.close()
    assert np.array_equal(blocks[0][0], blocks[1][0]) and blocks[0][1:] == blocks[1][1:]


def test_FR_gpf_full_size(lr, oracle):
    """30k-point pair through --mode GPF (factor 0.5 so that the filter really selects) against the oracle pipeline."""
    p = synth.make_pair(N=30000, seed=52, clustered=True)
    a = Args(mode="GPF", codebase="GC", iters=20000, GPF_factor=0.5)
    t = lr.torch.from_numpy
    T, elapsed, _, _, n_init, _, n_filt, _ = lr.FR.FR(t(p["xyz0"]), t(p["xyz1"]), t(p["feats0"]), t(p["feats1"]), a, p["T_gt"])
    e = oracle.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode="GPF", iters=20000, seed=51, args=a, **gc_oracle_kwargs(a))
    assert n_filt == len(e["idx0"]) and n_filt < n_init
    assert np.radians(oracle.rotation_error_deg(T, e["T"])) <= 1e-4 and oracle.translation_error_cm(T, e["T"]) / 100 <= 1e-3
    # the reference's own GPF list of this pair (tests/golden/make_golden_full.py): what FR() filtered down to, and the standalone call
    g = golden("g12_full_30k.npz")
    assert n_filt == len(g["c52_gpf05_idx0"]) and np.array_equal(e["idx0"], g["c52_gpf05_idx0"]) and np.array_equal(e["idx1"], g["c52_gpf05_idx1"])
    i0, i1, i2, _ = lr.matching.find_2nn(t(p["feats0"]), t(p["feats1"]))
    assert np.array_equal(i1.numpy(), g["c52_idx1"])
    k = lr.matching.Grid_Prioritized_Filter(t(p["feats0"]), t(p["feats1"]), i0, i1, i2, t(p["xyz0"]), a)
    assert np.array_equal(k[0].numpy(), g["c52_gpf05_idx0"]) and np.array_equal(k[1].numpy(), g["c52_gpf05_idx1"])


@pytest.mark.parametrize("conf,batch", [(0.999, 0), (0.999, 1024), (0.9, 512), (0.999999, 2048)])
def test_ransac_confidence_early_exit_matches_oracle(lr, oracle, conf, batch):
    src, tgt, T_gt = _planted(n=6000, inlier=0.3, seed=31)
    T, info = lr.ransac.ransac_dev(src, tgt, 50000, seed=3, confidence=conf, batch=batch)
    Te, einfo = oracle.ransac(src, tgt, 50000, seed=3, confidence=conf, batch=batch)
    assert info == einfo and np.array_equal(T, Te)
    assert info["n_ids"] < 50000 and info["n_ids"] % (batch or 1024) == 0           # stopped at a batch boundary
    assert oracle.rotation_error_deg(T, T_gt) < 1.0


def test_ransac_iteration_property_more_iters_never_worse(lr):
    src, tgt, _ = _planted(n=5000, inlier=0.25, seed=77)
    best = 0
    for iters in (200, 2000, 20000):
        _, info = lr.ransac.ransac_dev(src, tgt, iters, seed=5)
        assert info["best_count"] >= best          # hypotheses 0..iters-1 are a prefix of the longer run
        best = info["best_count"]


@pytest.mark.parametrize("ns,n,iters,seed", [(3, 3000, 4000, 11), (4, 2000, 3000, 12)])
def test_msac_scoring_same_winner_as_oracle(lr, oracle, ns, n, iters, seed):
    """scoring = 1: the model with the largest sum over inliers of (thr^2 - d^2) wins (GC-RANSAC's MSAC cost)."""
    src, tgt, T_gt = _planted(n=n, inlier=0.25, seed=seed)
    T, info = lr.ransac.ransac_dev(src, tgt, iters, sample_size=ns, seed=seed, scoring=1)
    Te, einfo = oracle.ransac(src, tgt, iters, sample_size=ns, seed=seed, scoring=1)
    assert info == einfo and np.array_equal(T, Te)
    # brute force over the hypothesis ids: nobody has a larger cost (ties towards the lower id)
    Tq = int(np.float32(0.36) * np.float32(1048576.0))
    best = None
    for h in range(iters):
        ok, Th, _ = oracle.hypothesis(src, tgt, h, sample_size=ns, seed=seed)
        if not ok:
            continue
        c, q = oracle.score(src, tgt, Th)
        if c and (best is None or c * Tq - q > best[0]):
            best = (c * Tq - q, h)
    assert best[1] == info["best_h"]
    assert oracle.rotation_error_deg(T, T_gt) < 1.0


def test_ransac_one_million_iterations_config4(lr, oracle):
    """BASELINE config #4 runs --iters 1000000: hypothesis ids, model storage and the batched exit at that length."""
    src, tgt, T_gt = _planted(n=1500, inlier=0.2, seed=31)
    T, info = lr.ransac.ransac_dev(src, tgt, 1_000_000, seed=51)                       # every id evaluated
    Te, einfo = oracle.ransac(src, tgt, 1_000_000, seed=51)
    assert info == einfo and np.array_equal(T, Te) and info["n_ids"] == 1_000_000
    T2, info2 = lr.ransac.ransac_dev(src, tgt, 1_000_000, seed=51, confidence=0.999)   # default batches: 1024, 8192, ... (eightfold)
    Te2, einfo2 = oracle.ransac(src, tgt, 1_000_000, seed=51, confidence=0.999)
    assert info2 == einfo2 and np.array_equal(T2, Te2) and info2["n_ids"] in (1024, 9216)
    assert oracle.rotation_error_deg(T2, T_gt) < 1.0


# ----------------------------------------------------------------------------- ICP (next row f1)
@pytest.mark.parametrize("n,rho,seed,offset", [(3000, 0.6, 5, 0.25), (4000, 0.4, 6, 0.4), (1500, 0.8, 7, 0.1)])
def test_icp_matches_oracle(lr, oracle, n, rho, seed, offset):
    xyz0, xyz1, T_gt = synth.make_clouds(n, n, rho, seed)
    T0 = T_gt.copy(); T0[:3, 3] += [offset, -0.5 * offset, 0.1]
    ang = np.radians(1.0); c, s = np.cos(ang), np.sin(ang)
    T0[:3, :3] = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]) @ T0[:3, :3]
    T, info = lr.ransac.icp_dev(xyz0, xyz1, T0)
    Te, einfo = oracle.icp(xyz0, xyz1, T0)
    assert info["n_corr"] == einfo["n_corr"] and info["iterations"] == einfo["iterations"]
    np.testing.assert_allclose(T, Te, rtol=0, atol=1e-9)
    assert abs(info["fitness"] - einfo["fitness"]) < 1e-12 and abs(info["inlier_rmse"] - einfo["inlier_rmse"]) < 1e-9
    # ICP pulls the perturbed start back onto the planted motion
    assert oracle.translation_error_cm(T, T_gt) < oracle.translation_error_cm(T0, T_gt)
    assert oracle.rotation_error_deg(T, T_gt) < 0.2 and oracle.translation_error_cm(T, T_gt) < 5


def test_icp_edge_cases(lr, oracle):
    xyz0, xyz1, T_gt = synth.make_clouds(800, 700, 0.5, 3)
    # max_iter 0: the start transform is returned, evaluated once
    T, info = lr.ransac.icp_dev(xyz0, xyz1, T_gt, max_iter=0)
    assert np.array_equal(T, T_gt) and info["iterations"] == 0 and info["n_corr"] == oracle.icp(xyz0, xyz1, T_gt, max_iter=0)[1]["n_corr"]
    # no point within reach: nothing to fit, start transform comes back
    far = np.eye(4); far[:3, 3] = [1e4, 1e4, 1e4]
    T, info = lr.ransac.icp_dev(xyz0, xyz1, far)
    assert np.array_equal(T, far) and info["n_corr"] == 0


def test_register_pair_with_icp_block(lr, oracle):
    import ctypes
    p = synth.make_pair(N=5000, rho=0.5, s=0.9, seed=51)
    a = Args(mode="MNN", codebase="GC", iters=2000, icp=True)
    t = lr.torch.from_numpy
    params = lr.FR.pair_params(a)
    assert params.icp == 1
    dev = lr.torch.device("cuda")
    out = lr.FR.register_pair_dev(t(p["xyz0"]).to(dev), t(p["xyz1"]).to(dev), t(p["feats0"]).to(dev), t(p["feats1"]).to(dev), params)
    r = lr.FR.read_result(out)
    T = np.array(r.T[:]).reshape(4, 4); T_icp = np.array(r.T_icp[:]).reshape(4, 4)
    Te, einfo = oracle.icp(p["xyz0"], p["xyz1"], T)
    np.testing.assert_allclose(T_icp, Te, rtol=0, atol=1e-9)
    assert r.icp.n_corr == einfo["n_corr"] and r.icp.iterations == einfo["iterations"]


# ----------------------------------------------------------------------------- sibling callers (next row f4)
def test_fcgf_fast_and_dgr_callers(lr, oracle):
    from lidarregistration_amd import callers
    p = synth.make_pair(N=4000, rho=0.5, s=0.8, seed=23)
    t = lr.torch.from_numpy
    T, elapsed, pcd0, pcd1, ir = callers.FCGF_FAST_RANSAC(t(p["xyz0"]), t(p["xyz1"]), t(p["feats0"]), t(p["feats1"]), p["T_gt"], iters=20000)
    e = oracle.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode="MNN", iters=20000, sample_size=4, seed=51,
                             confidence=0.9999, refit_on_orig=1)
    np.testing.assert_allclose(T, e["T"], rtol=0, atol=1e-9)
    assert 0 < ir < 1 and pcd0.points.shape == (4000, 3)
    r = callers.DGR_register_FCGF(t(p["xyz0"]), t(p["xyz1"]), t(p["feats0"]), t(p["feats1"]), iters=20000, T_gt=p["T_gt"])
    e = oracle.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode="no_filter", iters=20000, sample_size=4, seed=51,
                             confidence=0.9999, refit_on_orig=3)
    np.testing.assert_allclose(r["base"], e["T"], rtol=0, atol=1e-9)
    assert oracle.rotation_error_deg(r["base"], p["T_gt"]) < 0.5
    # 'w_icp' = 'base' refined by ICP at 2 * voxel_size (deep_global_registration.py:556-563), as the oracle's ICP does it; off: = 'base'
    Ti, _ = oracle.icp(p["xyz0"], p["xyz1"], e["T"], max_dist=0.6)
    np.testing.assert_allclose(r["w_icp"], Ti, rtol=0, atol=1e-9)
    assert np.abs(r["w_icp"] - r["base"]).max() > 1e-9
    r0 = callers.DGR_register_FCGF(t(p["xyz0"]), t(p["xyz1"]), t(p["feats0"]), t(p["feats1"]), iters=20000, T_gt=p["T_gt"], use_icp=False)
    assert np.array_equal(r0["base"], r["base"]) and np.array_equal(r0["w_icp"], r0["base"])
    # the weighted refit is really different from the unweighted one
    e1 = oracle.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode="no_filter", iters=20000, sample_size=4, seed=51,
                              confidence=0.9999, refit_on_orig=1)
    assert np.abs(e1["T"] - e["T"]).max() > 1e-7


@pytest.mark.parametrize("kind", ["inf_norm_column", "nan_column", "overflowing_row_and_column"])
def test_nn_one_nonfinite_norm_among_unit_norm_descriptors(lr, oracle, kind):
    """with `inf - min <= 1e-4 inf` one column whose squared norm overflows selected the SIGN form of the walk's test for a
    cloud whose norms are not alike at all (results stayed right, the hit lists flooded).  The form is now chosen from finite norms only;
    here: the results with such a column present."""
    n0, n1 = 1500, 2300
    F0, F1 = synth.make_features(n0, n1, 32, 0.5, 1.0, 91)
    F1 = F1.copy(); F0 = F0.copy()
    if kind == "inf_norm_column":
        F1[777] *= np.float32(3e19)              # squared norm 9e38 > FLT_MAX: inf in fp32 (and in the f16 copy)
    elif kind == "nan_column":
        F1[777, 5] = np.nan
    else:
        F1[12] *= np.float32(3e19); F0[40] *= np.float32(3e19)
    i1, i2, s1, s2 = lr.matching.nn_top2_dev(F0, F1, want_2nd=True, want_dist=True)
    o1, o2, os1, os2 = oracle.nn_top2(F0, F1)
    ok = np.ones(n0, bool)
    if kind == "overflowing_row_and_column":
        ok[40] = False                           # every distance of that row is inf or NaN: its order is not defined by the contract
    assert np.array_equal(i1.cpu().numpy()[ok], o1[ok]) and np.array_equal(i2.cpu().numpy()[ok], o2[ok])
    assert np.array_equal(_bits(s1.cpu().numpy())[ok], _bits(os1)[ok])
    if kind == "nan_column":
        # under the contract (fmaxf(NaN, 1e-30) = 1e-30; torch.min returns the NaN) the NaN column is EVERY row's nearest neighbour: no
        # filter value can say so, the exact kernel re-does every row by the full scan when a column norm is not finite
        assert (o1 == 777).all() and (i1.cpu().numpy() == 777).all()
        t = lr.torch.from_numpy
        m = lr.matching.nn_to_mutual(t(F0), t(F1), lr.torch.arange(n0), t(o1.astype(np.int64)), t(o2.astype(np.int64)))
        em = oracle.nn_to_mutual(F0, F1, np.arange(n0), o1, o2)
        for a, b in zip(m, em):
            assert np.array_equal(a.numpy(), b)


def test_nn_candidate_store_overflow_falls_back_to_the_exact_scan(lr, oracle):
    """Every column identical (and many near-identical): each query row has thousands of candidates, far beyond a wave's list
    and segment -- the rows must come out of the exact full-column scan with torch.min's first-index order intact."""
    rng = np.random.default_rng(12)
    for n0, n1, kind in ((700, 3000, "identical"), (300, 2500, "two_values"), (1000, 1500, "near")):
        F0 = rng.normal(size=(n0, 32)).astype(np.float32); F0 /= np.linalg.norm(F0, axis=1, keepdims=True)
        base = rng.normal(size=(1, 32)).astype(np.float32); base /= np.linalg.norm(base)
        if kind == "identical":
            F1 = np.repeat(base, n1, axis=0)
        elif kind == "two_values":
            other = rng.normal(size=(1, 32)).astype(np.float32); other /= np.linalg.norm(other)
            F1 = np.where((np.arange(n1) % 3 == 0)[:, None], other, base).astype(np.float32)
        else:
            F1 = (base + 1e-6 * rng.normal(size=(n1, 32))).astype(np.float32)
        i1, i2, s1, s2 = lr.matching.nn_top2_dev(F0, F1, want_2nd=True, want_dist=True)
        o1, o2, os1, os2 = oracle.nn_top2(F0, F1)
        assert np.array_equal(i1.cpu().numpy(), o1) and np.array_equal(i2.cpu().numpy(), o2), kind
        assert np.array_equal(_bits(s1.cpu().numpy()), _bits(os1)) and np.array_equal(_bits(s2.cpu().numpy()), _bits(os2))
        # and the mutual filter over the same degenerate clouds
        e0 = np.arange(n0)
        m = oracle.nn_to_mutual(F0, F1, e0, o1, o2)
        g = lr.matching.nn_to_mutual(lr.torch.from_numpy(F0), lr.torch.from_numpy(F1), lr.torch.from_numpy(e0), lr.torch.from_numpy(o1.astype(np.int64)),
                                     lr.torch.from_numpy(o2.astype(np.int64)))
        assert all(np.array_equal(a.numpy(), b) for a, b in zip(g, m)), kind


def test_tuning_options_do_not_change_results(lr):
    """lr_workspace_option: the sampling stride of the filter pass, the strips of the reverse pass, the blocks (column strips) of the
    forward pass and the second-neighbour switch: any value gives the same lists and the same transform.  (The library reads no
    environment variable; _ext.DEFAULT_OPTIONS applies options to the workspaces the host mirror creates.)"""
    import hashlib
    from lidarregistration_amd import _ext, matching
    p = synth.make_pair(N=9000, N1=7000, rho=0.4, s=0.8, seed=77)
    t = lr.torch.from_numpy

    def run(options):
        _ext.DEFAULT_OPTIONS.clear(); _ext.DEFAULT_OPTIONS.update(options)
        for ws in matching._WS.values():
            ws.close()
        matching._WS.clear()
        try:
            i0, i1, i2, _ = matching.find_2nn(t(p["feats0"]), t(p["feats1"]))
            m = matching.nn_to_mutual(t(p["feats0"]), t(p["feats1"]), i0, i1, i2)
            a = Args(mode="MNN", codebase="open3D", iters=3000, ransac_n=3, o3d_conf=1.0)
            T = lr.FR.FR(t(p["xyz0"]), t(p["xyz1"]), t(p["feats0"]), t(p["feats1"]), a, p["T_gt"])[0]
        finally:
            _ext.DEFAULT_OPTIONS.clear()
        h = hashlib.sha256()
        for x in (i1, i2, m[0], m[1]):
            h.update(np.ascontiguousarray(x.numpy()).tobytes())
        h.update(np.ascontiguousarray(T).tobytes())
        return h.hexdigest()

    ref = run({})
    for options in ({"nn_sample_stride": 1}, {"nn_sample_stride": 3}, {"nn_sample_stride": 64}, {"rev_strips": 1}, {"rev_strips": 64},
                    {"nn_blocks": 64}, {"nn_blocks": 4096}, {"nn_second_auto": 1}, {"clock_probe": 1}, {"nn_blocks": 1, "rev_strips": 1}):
        assert run(options) == ref, options
    for ws in matching._WS.values():
        ws.close()
    matching._WS.clear()
    with pytest.raises(lr.ext.LidarRegError):
        lr.ext.Workspace(100, 100, 32, 10).set_option("nn_blocks", -1)


@pytest.mark.parametrize("spread0,spread1", [(0.0, 0.0), (6e-5, 0.0), (0.0, 6e-5), (9e-5, 1.1e-4), (3e-4, 0.0), (0.0, 3e-4), (0.3, 2e-5), (0.5, 0.5)])
def test_nn_both_forms_of_the_walk_test_by_norm_spread(lr, oracle, spread0, spread1):
    """The filter pass tests candidates with a sign test when all column norms lie within 1e-4 of each other and with a comparison
    against the per-column term otherwise (nn16_passb_kernel<SIGN>; both are launched, the norm range decides on the device).  The
    forward pass looks at cloud 1's norms, the reverse pass at cloud 0's: spreads on either side of the switch, in every combination,
    single pair (several strips, pooled thresholds) and a 3-pair batched call -- NN lists, distances and mutual lists bit-exact."""
    rng = np.random.default_rng(int(1e6 * (spread0 + 2 * spread1)) + 5)
    n0, n1 = 6100, 5900
    F0, F1 = synth.make_features(n0, n1, 32, 0.5, 1.0, 77)
    F0 = (F0 * np.sqrt(1.0 + spread0 * rng.random((n0, 1)))).astype(np.float32)      # squared norms in [1, 1 + spread]
    F1 = (F1 * np.sqrt(1.0 + spread1 * rng.random((n1, 1)))).astype(np.float32)
    i1, i2, s1, s2 = lr.matching.nn_top2_dev(F0, F1, want_2nd=True, want_dist=True)
    o1, o2, os1, os2 = oracle.nn_top2(F0, F1)
    assert np.array_equal(i1.cpu().numpy(), o1) and np.array_equal(i2.cpu().numpy(), o2)
    assert np.array_equal(_bits(s1.cpu().numpy()), _bits(os1)) and np.array_equal(_bits(s2.cpu().numpy()), _bits(os2))
    t = lr.torch.from_numpy
    m = oracle.nn_to_mutual(F0, F1, np.arange(n0), o1, o2)
    g = lr.matching.nn_to_mutual(t(F0), t(F1), t(np.arange(n0)), t(o1), t(o2))
    assert all(np.array_equal(a.numpy(), b) for a, b in zip(g, m))
    # the same pair next to two others in one batched call (one strip per row block, joint rounds): its lists again
    from tests.conftest import Args
    import ctypes
    dev = lr.torch.device("cuda", 0)
    params = lr.FR.pair_params(Args(mode="MNN", codebase="open3D", iters=500, ransac_n=3, o3d_conf=1.0))
    others = [synth.make_features(3000 + 500 * k, 4000, 32, 0.5, 1.0, 90 + k) for k in range(2)]
    pairs = []
    for A, B in [(F0, F1)] + others:
        pairs.append((lr.torch.rand(A.shape[0], 3, device=dev), lr.torch.rand(B.shape[0], 3, device=dev), t(A).to(dev), t(B).to(dev)))
    ws = lr.ext.Workspace(7000, 7000, 32, 500, max_pairs=3)
    ws.poison(0xA5)
    out = lr.FR.register_batch_dev(pairs, params, ws=ws).cpu().numpy()
    bufs = [lr.torch.empty(n0, dtype=lr.torch.int32, device=dev) for _ in range(4)]
    lr.ext.check(lr.ext.lib().lr_workspace_lists_at(ws.handle, 0, n0, *[b.data_ptr() for b in bufs], None))
    r = lr.ext.PairResult.from_buffer_copy(out[0].tobytes())
    nn1, nn2, c0, c1 = [b.cpu().numpy() for b in bufs]
    assert np.array_equal(nn1, o1) and np.array_equal(nn2, o2) and r.n_nn_fixed == 0
    assert r.n_corr == len(m[0]) and np.array_equal(c0[:r.n_corr], m[0]) and np.array_equal(c1[:r.n_corr], m[1])
    ws.close()


def test_single_pair_calls_launch_one_form_of_the_filter_pass_and_survive_a_wrong_guess(lr, oracle):
    """A single-pair call launches only the form of the filter pass (sign test / plain test) that the column cloud's norms asked for in
    the PREVIOUS call on the workspace (a flag in pinned host memory).  When the data changes form between two calls the launched
    kernel flags the miss and the exact kernel re-does every row by the full scan: the results stay the oracle's, call after call, in
    every order of unit-norm and scaled clouds -- forward NN and the mutual list (whose reverse pass has cloud 0 as its columns)."""
    n0, n1 = 1800, 2100
    U0, U1 = synth.make_features(n0, n1, 32, 0.5, 1.0, 17)
    rng = np.random.default_rng(3)
    S0 = (U0 * rng.uniform(0.5, 2.0, (n0, 1))).astype(np.float32); S1 = (U1 * rng.uniform(0.5, 2.0, (n1, 1))).astype(np.float32)
    ws_key = lr.matching.workspace(n0, n1, dim=32)          # (every call below lands on this cached workspace)
    t = lr.torch.from_numpy
    seq = [(U0, U1), (U0, U1), (S0, S1), (S0, S1), (U0, S1), (S0, U1), (U0, U1), (S0, S1), (U0, U1)]
    for k, (F0, F1) in enumerate(seq):
        # (after a wrong guess nobody wrote the call's candidate counts: whatever the arena holds must not be read -- garbage counts
        # were a memory fault in tools/soak_nn_big.py before the exact kernel learnt to leave the store alone)
        ws_key.poison(0xff if k % 2 else 0x7f)
        i1, i2, s1, s2 = lr.matching.nn_top2_dev(F0, F1, want_2nd=True, want_dist=True)
        o1, o2, os1, os2 = oracle.nn_top2(F0, F1)
        assert np.array_equal(i1.cpu().numpy(), o1) and np.array_equal(i2.cpu().numpy(), o2), k
        assert np.array_equal(_bits(s1.cpu().numpy()), _bits(os1)), k
        m = lr.matching.nn_to_mutual(t(F0), t(F1), lr.torch.arange(n0), t(o1.astype(np.int64)), t(o2.astype(np.int64)))
        em = oracle.nn_to_mutual(F0, F1, np.arange(n0), o1, o2)
        for a, b in zip(m, em):
            assert np.array_equal(a.numpy(), b), k
    assert lr.matching.workspace(n0, n1, dim=32) is ws_key
