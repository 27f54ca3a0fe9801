import sys, os, ctypes, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from lidarregistration_amd import FR, _ext, synth
class A:
    mode = "MNN"; codebase = "open3D"; iters = 50000; ransac_n = 3; o3d_conf = 1.0
params = FR.pair_params(A)
dev = torch.device("cuda", 0)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pairs = []
for k in range(P):
    p = synth.make_pair_dev(N=30000, seed=51 + k, device=dev)
    pairs.append((p["xyz0"], p["xyz1"], p["feats0"], p["feats1"]))
ws = _ext.Workspace(30000, 30000, 32, 50000, max_pairs=P)
out = FR.register_batch_dev(pairs, params, ws=ws); torch.cuda.synchronize()
t = time.time(); out = FR.register_batch_dev(pairs, params, ws=ws); torch.cuda.synchronize(); print("batch ms", (time.time() - t) * 1e3)
for k in range(P):
    r = _ext.PairResult.from_buffer_copy(out[k].cpu().numpy().tobytes())
    print(k, "n_corr", r.n_corr, "n_nn_fixed", r.n_nn_fixed, "best", r.ransac.best_count)
ws1 = _ext.Workspace(30000, 30000, 32, 50000)
o1 = FR.register_pair_dev(*pairs[0], params, ws=ws1); torch.cuda.synchronize()
t = time.time(); o1 = FR.register_pair_dev(*pairs[0], params, ws=ws1); torch.cuda.synchronize(); print("single ms", (time.time() - t) * 1e3)
r = _ext.PairResult.from_buffer_copy(o1.cpu().numpy().tobytes()); print("single n_nn_fixed", r.n_nn_fixed)
