"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, and exports every symbol the header declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

from lidarregistration_amd import _ext

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    return _ext.build()


def test_header_symbols_are_exported(libpath):
    hdr = open(os.path.join(ROOT, "include", "lidarreg.h")).read()
    declared = sorted(set(re.findall(r"LR_API\s+[\w\s\*]+?\b(lr_\w+)\s*\(", hdr)))
    assert declared, "no LR_API declarations found"
    assert sorted(_ext.SYMBOLS) == declared
    L = ctypes.CDLL(libpath)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/lidarreg.h but not exported"


def test_version_and_error_string(libpath):
    L = _ext.lib()
    assert L.lr_version() >= 100
    assert isinstance(L.lr_last_error(), bytes)


def test_struct_layouts_match_header():
    # sizes the C side static-asserts implicitly through its field order (include/lidarreg.h); both params structs start with struct_size
    R, P = _ext.RansacParams, _ext.PairParams
    assert ctypes.sizeof(R) == 72 and R.struct_size.offset == 0 and R.sample_size.offset == 4 and R.seed.offset == 24
    assert R.sampler.offset == 40 and R.scoring.offset == 48 and R.local_opt.offset == 52 and R.lo_rounds.offset == 56 and R.min_iters.offset == 68
    assert ctypes.sizeof(_ext.RansacResult) == 40
    assert ctypes.sizeof(_ext.PairResult) == 496
    assert ctypes.sizeof(P) == 112 and P.struct_size.offset == 0 and P.mode.offset == 4 and P.ransac.offset == 16 and P.gpf_factor.offset == 96
    # the Python mirror fills the sizes in (positional arguments start at sample_size, as before)
    r = R(3, 1, 0.36, 50000, 51)
    assert (r.struct_size, r.sample_size, r.use_elc, r.iters, r.seed) == (72, 3, 1, 50000, 51)
    p = P()
    assert p.struct_size == 112 and p.ransac.struct_size == 72
    p.ransac = r
    assert p.ransac.struct_size == 72 and p.ransac.iters == 50000


def test_params_from_another_header_version_are_rejected():
    """A caller built against lr_version 101 (no struct_size: the struct started with sample_size = 3 or 4; 64 / 96 bytes) is turned
    away before anything is read past the end of its shorter struct.  Checked before any HIP call: safe without a device."""
    L = _ext.lib()
    assert L.lr_version() == 103
    r = _ext.RansacParams(3, 1, 0.36, 1000, 51)
    r.struct_size = 3                                   # what an old caller's first field would hold
    one = ctypes.c_void_p(1)                            # non-null dummies: the size check comes before any dereference of them
    assert L.lr_ransac(one, one, one, 10, None, ctypes.byref(r), one, one, None) == -1
    assert b"struct_size" in L.lr_last_error() and b"lr_version 103" in L.lr_last_error()
    p = _ext.PairParams()
    p.mode = 1
    p.struct_size = 96
    pp = (ctypes.c_void_p * 1)(1)
    ip = (ctypes.c_int32 * 1)(10)
    assert L.lr_register_batch(one, 1, pp, pp, pp, pp, ip, ip, 32, ctypes.byref(p), one, None) == -1 and b"lr_pair_params.struct_size is 96" in L.lr_last_error()
    assert L.lr_register_pair(one, one, one, one, one, 10, 10, 32, ctypes.byref(p), one, None) == -1 and b"struct_size" in L.lr_last_error()


def test_bad_arguments_are_reported_not_crashed(libpath):
    L = _ext.lib()
    h = ctypes.c_void_p()
    # argument validation happens before any HIP call, so this is safe without a device
    assert L.lr_workspace_create(ctypes.byref(h), 0, 10, 32, 10) == -1
    assert b"positive" in L.lr_last_error()
    assert L.lr_workspace_create(ctypes.byref(h), 10, 10, 33, 10) == -1 and b"1 to 32" in L.lr_last_error()      # wider than FCGF's 32: refused
    assert L.lr_workspace_create(ctypes.byref(h), 10, 10, 0, 10) == -1
    assert L.lr_feat_ratio(None, None, 32, 4, None, None, None, None, None) == -1


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "lidarregistration_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), f"{f} imports the oracle"
                assert "liboracle" not in text, f"{f} references the oracle library"


def test_mode_aliases_and_defaults():
    from lidarregistration_amd import FR as fr
    from tests.conftest import Args
    p = fr.pair_params(Args(mode="MMN", codebase="GC", iters=None))
    assert p.mode == _ext.LR_MODE_MNN and p.ransac.iters == 500000 and p.ransac.sample_size == 3 and p.ransac.use_elc == 1
    assert abs(p.ransac.confidence - 0.999) < 1e-6 and p.refit == 0 and p.ransac.local_opt == 1     # GC: LO + final LSQ inside the RANSAC call
    p = fr.pair_params(Args(mode="GPF", codebase="open3D", iters=1000, GPF_factor=0.5, GPF_grid_wid=4))
    assert p.mode == _ext.LR_MODE_GPF and p.ransac.sample_size == 4 and p.gpf_factor == 0.5 and p.gpf_grid_wid == 4 and p.refit == 1
    assert abs(p.refit_thr2 - 0.36) < 1e-15 and abs(p.ransac.thr2 - 0.36) < 1e-7
    with pytest.raises(AssertionError):
        fr.pair_params(Args(mode="bogus"))


def test_pair_params_follow_the_reference_flags():
    """args namespace (Experiments/test.py:294-313) -> lr_pair_params, no device needed."""
    from lidarregistration_amd import FR
    from tests.conftest import Args
    p = FR.pair_params(Args(mode="GPF", codebase="GC", iters=None, prosac=True, GPF_factor=1.5, GPF_grid_wid=12))
    assert (p.mode, p.refit, p.gpf_grid_wid, p.gpf_factor) == (_ext.LR_MODE_GPF, 0, 12, 1.5)
    r = p.ransac
    assert (r.sample_size, r.use_elc, r.iters, r.sampler, r.scoring, r.local_opt) == (3, 1, 500000, 1, 2, 1)       # FR.py:65-67, GC_RANSAC.py:19-37
    assert (r.lo_rounds, r.lo_trials, r.lo_max_calls, r.min_iters) == (0, 0, 0, 0) and abs(r.effective_thr2() - 0.81) < 1e-6   # library defaults; (3/2 * 0.6)^2
    assert abs(r.confidence - 0.999) < 1e-7 and abs(r.thr2 - 0.36) < 1e-7
    p = FR.pair_params(Args(mode="MMN", codebase="GC", iters=1000, prosac=False, fast_rejection="NONE", GC_conf=0.9))
    assert (p.mode, p.ransac.sampler, p.ransac.use_elc, p.ransac.iters) == (_ext.LR_MODE_MNN, 2, 0, 1000)     # uniform, unique indices
    assert abs(p.ransac.confidence - 0.9) < 1e-7
    # --GC_LO False: only the final least squares (GC_RANSAC.py:36-37); flags the HIP path does not implement are refused
    assert FR.pair_params(Args(codebase="GC", GC_LO=False)).ransac.local_opt == 2
    # ... but gcransac_python.cpp:518-521,553-556 honour that switch only in the branches with a pre-verification (:571-591 do not)
    assert FR.pair_params(Args(codebase="GC", GC_LO=False, fast_rejection="NONE")).ransac.local_opt == 1
    assert FR.pair_params(Args(codebase="GC", lo_trials=5, lo_max_calls=3, min_iters=100)).ransac.lo_trials == 5
    import pytest
    assert FR.pair_params(Args(codebase="GC", fast_rejection="SPRT")).ransac.use_elc == 2
    with pytest.raises(NotImplementedError):
        FR.pair_params(Args(codebase="GC", spatial_coherence_weight=0.1))
    FR.pair_params(Args(codebase="open3D", fast_rejection="SPRT", spatial_coherence_weight=0.1))      # GC-only flags: ignored, as in FR.py:70-97
    with pytest.raises(AssertionError):
        FR.pair_params(Args(codebase="GC", fast_rejection="bogus"))
    p = FR.pair_params(Args(mode="no_filter", codebase="open3D", iters=2000))
    r = p.ransac
    assert (p.mode, p.refit, r.sample_size, r.use_elc, r.sampler, r.scoring) == (_ext.LR_MODE_NO_FILTER, 1, 4, 1, 0, 0)   # FR.py:128-137
    assert abs(r.confidence - 0.9995) < 1e-7 and r.local_opt == 0
    with pytest.raises(AssertionError):
        FR.pair_params(Args(mode="bogus"))
    with pytest.raises(AssertionError):
        FR.pair_params(Args(codebase="other"))


def test_no_ablation_switches_in_the_shipped_kernels():
    """The library's .hip files carry no development variants: the ablation switches of rounds 3-5 (LR_PB_EXP bits, geometric
    schedule, per-wave rounds, ...) are rebuilt from history by tools/pb_variant.sh; lr_nn16.hip keeps ONE switch, the statistics probe."""
    csrc = os.path.join(ROOT, "lidarregistration_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            text = open(os.path.join(csrc, f)).read()
            for word in ("LR_PB_EXP", "LR_PB_GEO", "LR_PB_JOINT", "LR_PB_P1FOLD", "LR_PB_PRIO"):
                assert word not in text, f"{f} still carries the development switch {word}"
    nn = open(os.path.join(csrc, "lr_nn16.hip")).read()
    conds = re.findall(r"^\s*#\s*(?:if|ifdef|ifndef|elif)\b.*$", nn, flags=re.M)
    assert len(conds) <= 6, conds
    assert any("LR_PB_PROBE" in c for c in conds)
