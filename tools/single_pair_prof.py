"""One pair at a time (lr_register_pair, the reference harness' call pattern): run under rocprofv3 --kernel-trace --stats."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarregistration_amd import FR, synth, _ext
for kv in os.environ.get("LIDARREG_OPTS", "").split():      # e.g. LIDARREG_OPTS="rev_strips=16 nn_blocks=768"
    k, v = kv.split("="); _ext.DEFAULT_OPTIONS[k] = int(v)
from tests.conftest import Args
cb = sys.argv[1] if len(sys.argv) > 1 else "open3D"
mode = sys.argv[2] if len(sys.argv) > 2 else "MNN"
p = synth.make_pair(N=30000, seed=51)
t = torch.from_numpy
A, B, FA, FB = (t(p[k]).cuda() for k in ("xyz0", "xyz1", "feats0", "feats1"))
a = Args(mode=mode, codebase=cb, iters=50000, ransac_n=3, o3d_conf=1.0, prosac=True) if cb == "open3D" else Args(mode=mode, codebase="GC", iters=None, prosac=True)
params = FR.pair_params(a)
ws = FR.workspace(30000, 30000, params.ransac.iters)
for _ in range(40):
    out = FR.register_pair_dev(A, B, FA, FB, params, ws=ws)
    torch.cuda.synchronize()
print(FR.read_result(out).n_corr)
