"""Single-pair latency of FR() (what the reference's harness logs as elapsed_time)."""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarregistration_amd import FR, synth
from tests.conftest import Args
p = synth.make_pair(N=30000, seed=51)
t = torch.from_numpy
A, B, FA, FB = (t(p[k]).cuda() for k in ("xyz0", "xyz1", "feats0", "feats1"))
for mode, cb in (("MNN", "open3D"), ("GPF", "GC")):
    a = Args(mode=mode, codebase=cb, iters=50000, ransac_n=3, o3d_conf=1.0, GC_conf=1.0, prosac=(cb == "GC"))
    ts, wh = [], []
    for _ in range(30):
        ts.append(FR.FR(A, B, FA, FB, a, p["T_gt"])[1]); wh.append(FR.last_timing["whole_path_s"])
    print(mode, cb, "FR elapsed us: median %.0f min %.0f   (whole device path of the call incl. the forward NN and the result read-back: median %.0f min %.0f)" % (
        1e6 * np.median(ts[5:]), 1e6 * min(ts), 1e6 * np.median(wh[5:]), 1e6 * min(wh)))
# the reference CLI's defaults: --codebase GC --prosac True --GC_conf 0.999, iters = 500k (FR.py:65-67)
for mode in ("MNN", "GPF"):
    a = Args(mode=mode, codebase="GC", iters=None, prosac=True)
    ts = []
    for _ in range(20):
        ts.append(FR.FR(A, B, FA, FB, a, p["T_gt"])[1])
    print(mode, "GC defaults (500k iters, conf 0.999): FR elapsed us: median %.0f min %.0f" % (1e6 * np.median(ts[5:]), 1e6 * min(ts)))
# with the harness' ICP refinement stage inside the same call (Experiments/test.py:183-189)
from lidarregistration_amd import _ext
import ctypes, time
for mode in ("MNN", "GPF"):
    a = Args(mode=mode, codebase="GC", iters=None, prosac=True, icp=True)
    params = FR.pair_params(a)
    ws = FR.workspace(30000, 30000, params.ransac.iters)
    ts = []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.time()
        out = FR.register_pair_dev(A, B, FA, FB, params, ws=ws)
        r = FR.read_result(out)
        ts.append(time.time() - t0)
    print(mode, "GC defaults + ICP: us median %.0f  (icp iterations %d, fitness %.3f)" % (1e6 * np.median(ts[5:]), r.icp.iterations, r.icp.fitness))
