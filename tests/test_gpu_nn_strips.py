"""One column strip per row block -- what every 32-pair batched call of the headline workload runs with -- against several strips and the
oracle: NN lists, distances and mutual lists bit for bit on inputs that stress the candidate store and the exact verification (segment
overflow from duplicates, query rows that are not finite or out of the f16 range, tiny and ragged clouds), a 32-pair batched call at
test sizes with the strip count forced both ways, and the clock probe of the filter pass (lr_workspace_clock).  (Written in round 6 for the
fused verification -- a filter-pass wave verifying its own rows -- which measured 1 % slower and was removed again: docs/HISTORY.md,
profiles/r06_fused_verify_ab.txt.  The cases stay: they pin the strip-count independence of every list.)  Needs an MI355X."""
import ctypes

import numpy as np
import pytest

from lidarregistration_amd import synth
from tests.conftest import Args

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lr():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    from lidarregistration_amd import FR, _ext, matching
    _ext.lib()
    class NS: pass
    ns = NS(); ns.FR = FR; ns.torch = torch; ns.ext = _ext; ns.matching = matching
    return ns


def _bits(a):
    return np.asarray(a, np.float32).view(np.uint32)


def _nn(lr, F0, F1, options, need2=True):
    """lr_nn_top2 + lr_nn_to_mutual on a fresh, poisoned workspace with `options`."""
    t = lr.torch
    dev = t.device("cuda", 0)
    n0, n1 = len(F0), len(F1)
    ws = lr.ext.Workspace(n0, n1, 32, 10)
    for k, v in options.items():
        ws.set_option(k, v)
    ws.poison(0x5B)
    f0, f1 = t.from_numpy(F0).to(dev), t.from_numpy(F1).to(dev)
    i1 = t.empty(n0, dtype=t.int32, device=dev); i2 = t.empty(n0, dtype=t.int32, device=dev)
    s1 = t.empty(n0, dtype=t.float32, device=dev); s2 = t.empty(n0, dtype=t.float32, device=dev)
    L = lr.ext.lib()
    lr.ext.check(L.lr_nn_top2(ws.handle, f0.data_ptr(), n0, f1.data_ptr(), n1, 32, i1.data_ptr(), i2.data_ptr() if need2 else None,
                              s1.data_ptr(), s2.data_ptr() if need2 else None, None))
    bb = t.empty(n0, dtype=t.uint8, device=dev); o0 = t.empty(n0, dtype=t.int32, device=dev); o1 = t.empty(n0, dtype=t.int32, device=dev)
    cnt = t.zeros(1, dtype=t.int32, device=dev)
    lr.ext.check(L.lr_nn_to_mutual(ws.handle, f0.data_ptr(), n0, f1.data_ptr(), n1, 32, i1.data_ptr(), None, bb.data_ptr(), o0.data_ptr(), o1.data_ptr(), None,
                                   cnt.data_ptr(), None))
    t.cuda.synchronize()
    m = int(cnt.item())
    out = dict(i1=i1.cpu().numpy(), s1=s1.cpu().numpy(), bb=bb.cpu().numpy(), o0=o0.cpu().numpy()[:m], o1=o1.cpu().numpy()[:m])
    if need2:
        out.update(i2=i2.cpu().numpy(), s2=s2.cpu().numpy())
    ws.close()
    return out


ONE_STRIP = {"nn_blocks": 1, "rev_strips": 1}       # every row block of both directions has one strip


def _cases():
    rng = np.random.default_rng(606)
    unit = lambda n: (lambda a: (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32))(rng.normal(size=(n, 32)))
    yield "unit", unit(5000), unit(6100)
    F0, F1 = synth.make_features(7000, 5200, 32, 0.5, 0.9, 3)
    yield "overlap", F0, F1
    yield "scaled", (unit(3000) * rng.uniform(0.5, 3.0, size=(3000, 1))).astype(np.float32), (unit(3300) * rng.uniform(0.5, 3.0, size=(3300, 1))).astype(np.float32)
    # duplicates: segments overflow -> those waves give up, their rows go through the exact kernel's full scan
    base = unit(1)
    dup = np.where((np.arange(4500) % 3 == 0)[:, None], unit(1), base).astype(np.float32)
    yield "duplicates", unit(1500), dup
    # a mix: a few hundred identical columns among ordinary ones (some waves overflow, most do not)
    mix = unit(6000); mix[1000:1400] = mix[1000]
    q = unit(4000); q[::7] = mix[1000] + 1e-3 * unit(len(q[::7]))
    yield "mixed_overflow", q.astype(np.float32), mix
    # query rows that are not finite / out of the f16 range: their waves give up (all or nothing per wave), the others verify
    bad = unit(4000); bad[5] = np.nan; bad[700, 3] = np.inf; bad[2049] *= 1e6; bad[3999, 31] = -7e4
    yield "bad_rows", bad, unit(4200)
    # tiny and ragged
    yield "tiny", unit(3), unit(2)
    yield "ragged", unit(257), unit(65)
    yield "one_column", unit(300), unit(1)


@pytest.mark.parametrize("name,F0,F1", list(_cases()), ids=[c[0] for c in _cases()])
def test_one_strip_equals_default_equals_oracle(lr, oracle, name, F0, F1):
    one = _nn(lr, F0, F1, ONE_STRIP)
    many = _nn(lr, F0, F1, {"nn_blocks": 4096, "rev_strips": 16})
    default = _nn(lr, F0, F1, {})
    for k in one:
        assert np.array_equal(_bits(one[k]) if one[k].dtype == np.float32 else one[k], _bits(many[k]) if many[k].dtype == np.float32 else many[k]), (name, k)
        assert np.array_equal(_bits(one[k]) if one[k].dtype == np.float32 else one[k], _bits(default[k]) if default[k].dtype == np.float32 else default[k]), (name, k)
    o1, o2, os1, os2 = oracle.nn_top2(F0, F1)
    finite = np.isfinite(F0).all(axis=1)
    assert np.array_equal(one["i1"][finite], o1[finite]) and np.array_equal(_bits(one["s1"])[finite], _bits(os1)[finite]), name
    if len(F1) > 1:
        assert np.array_equal(one["i2"][finite], o2[finite]) and np.array_equal(_bits(one["s2"])[finite], _bits(os2)[finite]), name
    if finite.all():
        m = oracle.nn_to_mutual(F0, F1, np.arange(len(F0)), o1, o2)
        assert np.array_equal(one["o0"], m[0]) and np.array_equal(one["o1"], m[1]), name


def test_one_strip_top1_only(lr, oracle):
    """need = 1 (no second neighbour asked for): shorter lists, the column key of the reverse pass is the NN distance."""
    F0, F1 = synth.make_features(5200, 4800, 32, 0.4, 0.9, 11)
    a = _nn(lr, F0, F1, ONE_STRIP, need2=False)
    b = _nn(lr, F0, F1, {}, need2=False)
    o1, _, os1, _ = oracle.nn_top2(F0, F1)
    assert np.array_equal(a["i1"], o1) and np.array_equal(_bits(a["s1"]), _bits(os1))
    for k in a:
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("kw", [dict(mode="MNN", codebase="open3D", ransac_n=3, o3d_conf=1.0),
                                dict(mode="GPF", codebase="GC", GPF_factor=0.5, prosac=True)])
def test_batched_call_one_strip_vs_many(lr, kw):
    """A 32-pair batched call at the strip count the headline workload runs with (one strip per row block, forced here at test sizes by
    nn_blocks_batch = 1) and with as many strips as the clouds allow: result blocks and lists bit for bit."""
    a = Args(iters=2000, **kw)
    params = lr.FR.pair_params(a)
    dev = lr.torch.device("cuda", 0)
    sizes = [(2600 + 97 * (k % 7), 2400 + 131 * (k % 5)) for k in range(32)]
    sizes[3] = (64, 70); sizes[17] = (4099, 5000)
    devp = []
    for k, (n0, n1) in enumerate(sizes):
        p = synth.make_pair(N=n0, N1=n1, rho=0.5, s=0.9, seed=4000 + k, clustered=(a.mode == "GPF"))
        devp.append(tuple(lr.torch.from_numpy(p[key]).to(dev) for key in ("xyz0", "xyz1", "feats0", "feats1")))
    outs, lists = [], []
    for one in (1, 0):
        ws = lr.ext.Workspace(4099, 5000, 32, a.iters, max_pairs=32)
        ws.set_option("nn_blocks_batch", 1 if one else 1 << 20)
        ws.poison(0x77 + one)
        out = lr.FR.register_batch_dev(devp, params, ws=ws)
        lr.torch.cuda.synchronize()
        o = out.cpu().numpy().copy(); o[:, 312:316] = 0          # (reserved[0]: a scheduling-dependent diagnostic)
        outs.append(o)
        w = 5000
        bufs = [lr.torch.zeros((32, w), dtype=lr.torch.int32, device=dev) for _ in range(4)]
        res = [lr.ext.PairResult.from_buffer_copy(out[k].cpu().numpy().tobytes()) for k in range(32)]
        lr.ext.check(lr.ext.lib().lr_workspace_lists_batch(ws.handle, 32, 4099, *[b.data_ptr() for b in bufs], None))
        lr.torch.cuda.synchronize()
        per = []
        for k in range(32):
            n0 = sizes[k][0]
            flat = [b.view(-1)[k * 4099:k * 4099 + 4099].cpu().numpy() for b in bufs]
            per.append((flat[0][:n0], flat[1][:n0], flat[2][:res[k].n_corr], flat[3][:res[k].n_corr]))
        lists.append(per)
        assert all(r.n_nn_fixed == 0 for r in res)
        ws.close()
    assert np.array_equal(outs[0], outs[1])
    for k in range(32):
        for x, y in zip(lists[0][k], lists[1][k]):
            assert np.array_equal(x, y), k


def test_headline_batch_and_clock_probe(lr):
    """config #2 itself: 32 pairs of 30k points in one call.  Results equal with the clock probe on and off, and the probe reports a
    plausible shader clock for the filter-pass blocks (lr_workspace_clock)."""
    a = Args(mode="MNN", codebase="open3D", iters=50000, ransac_n=3, o3d_conf=1.0)
    params = lr.FR.pair_params(a)
    dev = lr.torch.device("cuda", 0)
    devp = []
    for k in range(32):
        p = synth.make_pair_dev(N=30000, seed=51 + k, device=dev)
        devp.append((p["xyz0"], p["xyz1"], p["feats0"], p["feats1"]))
    outs = []
    for probe in (1, 0):
        ws = lr.ext.Workspace(30000, 30000, 32, a.iters, max_pairs=32)
        ws.set_option("clock_probe", probe)
        assert ws.clock(reset=True)[1] == 0
        out = lr.FR.register_batch_dev(devp, params, ws=ws)
        lr.torch.cuda.synchronize()
        mhz, cyc, tk = ws.clock(reset=True)
        if probe:
            assert 500.0 < mhz < 3000.0 and cyc > 0 and tk > 0, (mhz, cyc, tk)
        else:
            assert (mhz, cyc, tk) == (0.0, 0, 0)
        assert ws.clock()[1] == 0
        o = out.cpu().numpy().copy(); o[:, 312:316] = 0
        outs.append(o)
        ws.close()
    assert np.array_equal(outs[0], outs[1])
    r = lr.ext.PairResult.from_buffer_copy(outs[0][0].tobytes())
    assert r.status == 0 and r.n_nn_fixed == 0 and r.n_corr > 10000
