"""The CALL the reference makes into its native RANSAC, recorded -- not restated.

Runs only where /root/reference exists.  Experiments/algorithms/GC_RANSAC.py is executed as it is (its mixed tab / space indentation at
:29-30, a TabError under Python 3, normalised in memory with str.expandtabs) with a stand-in `pygcransac` module that records what
`findRigidTransform` receives: the keyword arguments for every combination of the CLI flags, and whether the point arrays arrive sorted by
descending match quality.  The table is written to g14_gc_call.json; tests/test_gpu_gc.py::test_pygcransac_call_shape replays it against
lidarregistration_amd.pygcransac.

    python tests/golden/make_golden_gc_call.py
"""
import itertools
import json
import os
import sys
import types

import numpy as np

REF = os.environ.get("LIDARREG_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    src = open(os.path.join(REF, "Experiments", "algorithms", "GC_RANSAC.py")).read().expandtabs(4)
    seen = {}
    stub = types.ModuleType("pygcransac")

    def findRigidTransform(x1, x2, **kw):
        seen["x1"], seen["x2"], seen["kw"] = np.array(x1), np.array(x2), dict(kw)
        pose = np.arange(16, dtype=np.float64).reshape(4, 4)
        return pose, np.ones(len(x1), bool)
    stub.findRigidTransform = findRigidTransform
    sys.modules["pygcransac"] = stub
    ns = {"__name__": "ref_gc_ransac"}
    exec(compile(src, "GC_RANSAC.py", "exec"), ns)
    rng = np.random.default_rng(0)
    A = rng.normal(size=(50, 3)); B = rng.normal(size=(50, 3)); q = rng.random(50).astype(np.float32)
    rows = []
    for fast_rejection, GC_LO, prosac in itertools.product(["ELC", "SPRT", "NONE"], [True, False], [True, False]):
        args = types.SimpleNamespace(spatial_coherence_weight=0.0, prosac=prosac, GC_conf=0.999, fast_rejection=fast_rejection, GC_LO=GC_LO)
        out = ns["GC_RANSAC"](A, B, distance_threshold=0.6, num_iterations=20000, args=args, match_quality=q)
        T = out[0] if isinstance(out, tuple) else out
        order = np.argsort(-q)
        presorted = bool(np.array_equal(seen["x1"], A[order]) and np.array_equal(seen["x2"], B[order]))
        assert presorted or (np.array_equal(seen["x1"], A) and np.array_equal(seen["x2"], B))
        kw = {k: (bool(v) if isinstance(v, (bool, np.bool_)) else (int(v) if isinstance(v, (int, np.integer)) else float(v))) for k, v in seen["kw"].items()}
        rows.append(dict(flags=dict(fast_rejection=fast_rejection, GC_LO=GC_LO, prosac=prosac, GC_conf=0.999, spatial_coherence_weight=0.0),
                         threshold_arg=0.6, iterations_arg=20000, kwargs=kw, presorted_by_descending_quality=presorted,
                         pose_is_transposed=bool(np.array_equal(np.asarray(T), np.arange(16, dtype=np.float64).reshape(4, 4).T))))
    with open(os.path.join(HERE, "g14_gc_call.json"), "w") as f:
        json.dump(rows, f, indent=1)
    print(len(rows), "rows ->", os.path.join(HERE, "g14_gc_call.json"))
    print(rows[0])


if __name__ == "__main__":
    main()
