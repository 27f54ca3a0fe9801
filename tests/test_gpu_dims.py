"""Descriptor widths other than FCGF's 32 (Experiments/algorithms/matching.py:22-65 is dimension-agnostic; FCGF_FAST/net/BBR_F.py:148-176
calls it with D = 3).  The library pads narrower descriptors with zeros and runs 32 wide: a zero term changes neither the fma chains of
the arithmetic contract nor the ratio's sum, so everything must equal the oracle's dim-wide results bit for bit."""
import ctypes

import numpy as np
import pytest

from lidarregistration_amd import synth
from tests.conftest import Args, gc_oracle_kwargs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lr():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    from lidarregistration_amd import FR, _ext, matching, ransac
    _ext.lib()
    class NS: pass
    ns = NS(); ns.FR = FR; ns.matching = matching; ns.ransac = ransac; ns.torch = torch; ns.ext = _ext
    return ns


def _bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("dim", [1, 2, 3, 5, 16, 24, 31])
@pytest.mark.parametrize("n0,n1,seed", [(3000, 2500, 2), (251, 499, 3), (33, 65, 5), (5000, 1029, 7)])
def test_nn_top2_bit_exact_at_other_widths(lr, oracle, dim, n0, n1, seed):
    F0, F1 = synth.make_features(n0, n1, dim, 0.5, 1.0, seed)
    if dim > 1:      # (unit vectors in one dimension are +-1: every distance ties)
        F0 = (F0 * np.random.default_rng(seed).uniform(0.5, 2.0, (n0, 1))).astype(np.float32)          # not unit norm: the plain form of the walk's test
    i1, i2, s1, s2 = lr.matching.nn_top2_dev(F0, F1, want_2nd=True, want_dist=True)
    o1, o2, os1, os2 = oracle.nn_top2(F0, F1)
    assert np.array_equal(i1.cpu().numpy(), o1) and np.array_equal(i2.cpu().numpy(), o2)
    assert np.array_equal(_bits(s1.cpu().numpy()), _bits(os1)) and np.array_equal(_bits(s2.cpu().numpy()), _bits(os2))


@pytest.mark.parametrize("dim", [3, 16, 31])
def test_filters_at_other_widths(lr, oracle, dim):
    n0, n1, seed = 3000, 2600, 77
    F0, F1 = synth.make_features(n0, n1, dim, 0.5, 1.0, seed)
    xyz0, _, _ = synth.make_clouds(n0, n1, 0.5, seed, clustered=True)
    t = lr.torch.from_numpy
    i0, i1, i2, _ = lr.matching.find_2nn(t(F0), t(F1))
    e0, e1, e2, _ = oracle.find_2nn(F0, F1)
    assert np.array_equal(i1.numpy(), e1) and np.array_equal(i2.numpy(), e2)
    m = lr.matching.nn_to_mutual(t(F0), t(F1), i0, i1, i2)
    em = oracle.nn_to_mutual(F0, F1, e0, e1, e2)
    for a, b in zip(m, em):
        assert np.array_equal(a.numpy(), b)
    r = lr.matching.calc_distance_ratio_in_feature_space(t(F0), t(F1), m[0], m[1], m[2]).cpu().numpy()
    assert np.array_equal(_bits(r), _bits(oracle.calc_distance_ratio_in_feature_space(F0, F1, em[0], em[1], em[2])))
    for factor in (2.0, 0.4):
        a = Args(GPF_factor=factor)
        g = lr.matching.Grid_Prioritized_Filter(t(F0), t(F1), i0, i1, i2, t(xyz0), a)
        eg = oracle.Grid_Prioritized_Filter(F0, F1, e0, e1, e2, xyz0, a)
        assert np.array_equal(g[0].numpy(), eg[0]) and np.array_equal(g[1].numpy(), eg[1]) and np.array_equal(g[2].numpy(), eg[2])


@pytest.mark.parametrize("dim,mode,codebase", [(16, "MNN", "open3D"), (16, "GPF", "GC"), (3, "MNN", "GC"), (31, "no_filter", "open3D")])
def test_FR_at_other_widths_matches_oracle_pipeline(lr, oracle, dim, mode, codebase):
    N, iters = 4000, 2000
    p = synth.make_pair(N=N, D=dim, rho=0.5, s=0.3 if dim == 3 else 0.9, seed=51, clustered=(mode == "GPF"))
    a = Args(mode=mode, codebase=codebase, iters=iters, GPF_factor=0.5)
    t = lr.torch.from_numpy
    T, elapsed, _, _, n_init, _, n_filt, _ = lr.FR.FR(t(p["xyz0"]), t(p["xyz1"]), t(p["feats0"]), t(p["feats1"]), a, p["T_gt"])
    kw = gc_oracle_kwargs(a) if codebase == "GC" else dict(sample_size=4, use_elc=True, confidence=a.o3d_conf, refit_on_orig=1, scoring=0)
    e = oracle.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode=mode, iters=iters, seed=51, args=a, **kw)
    assert n_init == N and n_filt == len(e["idx0"])
    np.testing.assert_allclose(T, e["T"], rtol=0, atol=1e-9)


def test_batch_at_width_16_is_bit_identical_to_single_pairs(lr):
    dev = lr.torch.device("cuda", 0)
    sizes = [(3000, 3000), (4100, 2500), (257, 999), (64, 64)]
    a = Args(mode="GPF", codebase="GC", iters=3000, GPF_factor=0.5, prosac=True)
    params = lr.FR.pair_params(a)
    devp = []
    for k, (n0, n1) in enumerate(sizes):
        p = synth.make_pair(N=n0, N1=n1, D=16, rho=0.5, s=0.9, seed=300 + k, clustered=True)
        devp.append(tuple(lr.torch.from_numpy(p[key]).to(dev) for key in ("xyz0", "xyz1", "feats0", "feats1")))
    wsb = lr.ext.Workspace(5000, 5000, 16, a.iters, max_pairs=len(sizes))
    wsb.poison(0xA5)
    outb = lr.FR.register_batch_dev(devp, params, ws=wsb)
    lr.torch.cuda.synchronize()
    ws1 = lr.ext.Workspace(5000, 5000, 16, a.iters)
    for k in range(len(sizes)):
        ws1.poison(0x3C + k)
        out1 = lr.FR.register_pair_dev(*devp[k], params, ws=ws1)
        lr.torch.cuda.synchronize()
        bb = outb[k].cpu().numpy().copy(); ss = out1.cpu().numpy().copy()
        bb[312:316] = 0; ss[312:316] = 0          # (reserved[0]: a scheduling-dependent diagnostic, see tests/test_gpu_batch.py)
        assert bb.tobytes() == ss.tobytes(), k


def test_width_mismatch_and_too_wide_are_refused(lr):
    ws = lr.ext.Workspace(1000, 1000, 16, 100)
    F = lr.torch.zeros((1000, 32), device="cuda")
    i1 = lr.torch.zeros(1000, dtype=lr.torch.int32, device="cuda")
    rc = lr.ext.lib().lr_nn_top2(ws.handle, F.data_ptr(), 1000, F.data_ptr(), 1000, 32, i1.data_ptr(), None, None, None, None)
    assert rc == -1 and b"dim 32 != workspace dim 16" in lr.ext.lib().lr_last_error()
    h = ctypes.c_void_p()
    assert lr.ext.lib().lr_workspace_create(ctypes.byref(h), 100, 100, 33, 10) == -1
