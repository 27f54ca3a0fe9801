import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarregistration_amd import FR, synth
from tests.conftest import Args
p = synth.make_pair(N=30000, seed=51)
t = torch.from_numpy
A, B, FA, FB = (t(p[k]).cuda() for k in ("xyz0", "xyz1", "feats0", "feats1"))
a = Args(mode="MNN", codebase="GC", iters=None, prosac=True, icp=True)
params = FR.pair_params(a)
ws = FR.workspace(30000, 30000, params.ransac.iters)
for _ in range(20):
    out = FR.register_pair_dev(A, B, FA, FB, params, ws=ws)
    FR.read_result(out)
