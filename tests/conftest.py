import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def rot_diff_rad(A, B):
    """Rotation angle between two nearly equal transforms from the Frobenius distance of their rotation blocks
    (|Ra - Rb|_F = 2 sqrt(2) sin(theta/2)).  Unlike arccos((tr - 1)/2) it has no sqrt(eps) noise floor, which matters when one
    side is a float32-quantised golden (arccos of 1 - 1e-7 is 4e-4 rad)."""
    d = np.linalg.norm(np.asarray(A, np.float64)[:3, :3] - np.asarray(B, np.float64)[:3, :3])
    return float(2.0 * np.arcsin(min(1.0, d / (2.0 * np.sqrt(2.0)))))


def gc_oracle_kwargs(a):
    """Keyword arguments of oracle.register_pair that restate what --codebase GC runs for the flags in `a`
    (FR.pair_params): MSAC at the truncated threshold, unique-index sampling (PROSAC when a.prosac), local optimisation unless
    a.GC_LO is False (and a pre-verification is selected: gcransac_python.cpp:518-521), final iterated least squares inside the
    RANSAC call, no refit stage."""
    return dict(sample_size=3, use_elc={"NONE": 0, "ELC": 1, "SPRT": 2}[a.fast_rejection], confidence=a.GC_conf, refit_on_orig=0, scoring=2,
                local_opt=1 if (a.GC_LO or a.fast_rejection == "NONE") else 2, prosac=bool(a.prosac), unique=not a.prosac)


class Args:
    """argparse-like bag with the reference's hot-path defaults (Experiments/test.py:294-313)."""
    def __init__(self, **kw):
        self.mode = "MNN"; self.codebase = "open3D"; self.iters = 50000; self.prosac = False
        self.GPF_grid_wid = 10; self.GPF_factor = 2.0; self.GPF_max_matches = 10 ** 9
        self.spatial_coherence_weight = 0.0; self.GC_conf = 0.999; self.o3d_conf = 0.9995; self.fast_rejection = "ELC"; self.GC_LO = True
        self.__dict__.update(kw)
