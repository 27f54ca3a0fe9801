"""Randomised parity sweep through the C ABI with a large, *poisoned* scratch arena: no entry point may depend on what an
earlier call (or hipMalloc) left in the workspace.  Every case is run with the arena filled with 0x00, 0xFF and 0x7F and
compared with the oracle.  Needs an MI355X."""
import numpy as np
import pytest

from lidarregistration_amd import synth
from tests.conftest import Args

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lr():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    from lidarregistration_amd import FR, _ext, matching, ransac
    _ext.lib()
    class NS: pass
    ns = NS(); ns.FR = FR; ns.matching = matching; ns.ransac = ransac; ns.torch = torch; ns.ext = _ext
    # one big workspace for the whole module, so every small case runs inside a much larger arena
    ns.ws = matching.workspace(12000, 12000, 20000)
    return ns


def _poison(lr, byte):
    ws = lr.matching.workspace(1, 1, 1)
    assert ws is lr.ws, "the fuzz cases must stay inside the module's workspace"
    ws.poison(byte)


def _features(rng, n0, n1, kind):
    if kind == "correlated":
        return synth.make_features(n0, n1, 32, 0.5, 0.8, int(rng.integers(1 << 30)))
    if kind == "duplicates":
        base = rng.standard_normal((max(2, min(n0, n1) // 3), 32)).astype(np.float32)
        return base[rng.integers(len(base), size=n0)].copy(), base[rng.integers(len(base), size=n1)].copy()
    if kind == "scales":
        s0 = 10.0 ** rng.uniform(-3, 2, (n0, 1)); s1 = 10.0 ** rng.uniform(-3, 2, (n1, 1))
        return (rng.standard_normal((n0, 32)) * s0).astype(np.float32), (rng.standard_normal((n1, 32)) * s1).astype(np.float32)
    F0 = rng.standard_normal((n0, 32)).astype(np.float32); F1 = rng.standard_normal((n1, 32)).astype(np.float32)
    return F0, F1


@pytest.mark.parametrize("seed", range(6))
def test_nn_and_mutual_fuzz_with_poisoned_scratch(lr, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    t = lr.torch.from_numpy
    for kind in ("correlated", "duplicates", "scales", "random"):
        n0, n1 = int(rng.integers(1, 4000)), int(rng.integers(1, 4000))
        F0, F1 = _features(rng, n0, n1, kind)
        e0, e1, e2, _ = oracle.find_2nn(F0, F1)
        m = oracle.nn_to_mutual(F0, F1, e0, e1, e2)
        for byte in (0x00, 0xFF, 0x7F):
            _poison(lr, byte)
            i0, i1, i2, _ = lr.matching.find_2nn(t(F0), t(F1))
            assert np.array_equal(i1.numpy(), e1), (kind, n0, n1, byte)
            if n1 > 1:
                assert np.array_equal(i2.numpy(), e2), (kind, n0, n1, byte)
            _poison(lr, byte)
            g = lr.matching.nn_to_mutual(t(F0), t(F1), t(e0), t(e1), t(e2))
            assert all(np.array_equal(a.numpy(), b) for a, b in zip(g, m)), (kind, n0, n1, byte)


@pytest.mark.parametrize("seed,mode,codebase,prosac", [(1, "MNN", "open3D", False), (2, "GPF", "GC", True), (3, "MNN", "GC", True),
                                                      (4, "no_filter", "GC", False)])
def test_FR_with_poisoned_scratch(lr, oracle, seed, mode, codebase, prosac):
    N = 2000 + 500 * seed
    p = synth.make_pair(N=N, rho=0.5, s=0.9, seed=60 + seed, clustered=(mode == "GPF"))
    a = Args(mode=mode, codebase=codebase, iters=1500, GPF_factor=0.5, prosac=prosac, icp=True)
    from tests.conftest import gc_oracle_kwargs
    kw = gc_oracle_kwargs(a) if codebase == "GC" else dict(sample_size=4, use_elc=True, confidence=a.o3d_conf, refit_on_orig=1, scoring=0)
    e = oracle.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode=mode, iters=1500, seed=51, args=a, **kw)
    t = lr.torch.from_numpy
    outs = []
    for byte in (0x00, 0xFF, 0x7F):
        _poison(lr, byte)
        T = lr.FR.FR(t(p["xyz0"]), t(p["xyz1"]), t(p["feats0"]), t(p["feats1"]), a, p["T_gt"])[0]
        np.testing.assert_allclose(T, e["T"], rtol=0, atol=1e-9)
        outs.append(T)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
