"""lr_register_pair captured in a HIP graph (torch.cuda.CUDAGraph): the entry point launches on the given stream only, allocates
nothing and never synchronises, so a caller can capture it once and replay it on new data copied into the same buffers."""
import sys, os, time, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lidarregistration_amd import FR, synth, _ext
from tests.conftest import Args

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
for cb in ("open3D", "GC"):
    a = Args(mode="MNN", codebase=cb, iters=50000 if cb == "open3D" else None, ransac_n=3, o3d_conf=1.0, prosac=True)
    params = FR.pair_params(a); params.icp = 0
    ws = _ext.Workspace(n, n, 32, params.ransac.iters)
    bufs = [torch.empty((n, 3), device=dev), torch.empty((n, 3), device=dev), torch.empty((n, 32), device=dev), torch.empty((n, 32), device=dev)]
    out = torch.empty(ctypes.sizeof(_ext.PairResult), dtype=torch.uint8, device=dev)
    pairs = [synth.make_pair_dev(N=n, seed=51 + k, device=dev) for k in range(8)]
    def load(k):
        p = pairs[k % len(pairs)]
        for b, key in zip(bufs, ("xyz0", "xyz1", "feats0", "feats1")):
            b.copy_(p[key])
    load(0)
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        FR.register_pair_dev(*bufs, params, out=out, ws=ws, stream=s.cuda_stream)      # warm-up outside the capture
    s.synchronize()
    ref = [None] * len(pairs)
    t_eager = []
    for k in range(24):
        load(k); torch.cuda.synchronize()
        t0 = time.perf_counter()
        FR.register_pair_dev(*bufs, params, out=out, ws=ws, stream=s.cuda_stream)
        s.synchronize()
        t_eager.append(time.perf_counter() - t0)
        ref[k % len(pairs)] = out.cpu().numpy().copy()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        FR.register_pair_dev(*bufs, params, out=out, ws=ws, stream=s.cuda_stream)
    t_graph = []
    for k in range(24):
        load(k); torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        t_graph.append(time.perf_counter() - t0)
        got = out.cpu().numpy()
        assert np.array_equal(got[:256], ref[k % len(pairs)][:256]), "graph replay differs from the eager call"      # T and T_ransac, bit for bit
    print(f"{cb:7s} n={n}: eager call + sync {1e6 * np.median(t_eager[4:]):7.1f} us, graph replay + sync {1e6 * np.median(t_graph[4:]):7.1f} us (results identical)")
    ws.close()
