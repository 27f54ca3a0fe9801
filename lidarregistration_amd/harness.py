"""Per-rank evaluation loop around FR(): the role of eval_KITTI_per_pair in the reference (Experiments/test.py:91-234).

Pairs come from a PairSource (synthetic, list-driven synthetic surrogate, or a feature cache).  Several pairs are kept in
flight on separate HIP streams, each with its own workspace; results stay on the device until the shard is done.
Stats row layout = Experiments/test.py:98-100 (22 columns); the 4x4 transforms are returned next to it.
"""
import ctypes
import time

import numpy as np
import torch

from . import FR as fr
from . import _ext, io_lists, metrics, synth


class SyntheticSource:
    """`num_pairs` synthetic pairs; when `pair_list` (io_lists.read_pair_list) is given, pair k takes its ground-truth
    motion and overlap from row k of the list (SURVEY.md 8d "list-driven synthetic surrogate")."""

    def __init__(self, num_pairs, n=30000, rho=0.5, s=1.2, seed=51, pair_list=None, rho_scale=1.0, noise=0.05):
        self.num_pairs, self.n, self.rho, self.s, self.seed, self.pair_list = num_pairs, n, rho, s, seed, pair_list
        self.rho_scale, self.noise = rho_scale, noise        # (get_dev only) overlap multiplier and coordinate noise: the "hard" surrogate

    def __len__(self):
        return self.num_pairs

    def ids(self, k):
        if self.pair_list is not None:
            return int(self.pair_list["session"][k]), int(self.pair_list["src"][k]), int(self.pair_list["tgt"][k])
        return 0, k, k

    def get(self, k):
        rho = self.rho
        if self.pair_list is not None and self.pair_list["overlap"] is not None:
            rho = float(np.clip(self.pair_list["overlap"][k], 0.05, 0.95))
        p = synth.make_pair(N=self.n, rho=rho, s=self.s, seed=self.seed + k)
        if self.pair_list is not None:
            # re-plant the list's ground-truth motion: move cloud 1 from the synthetic motion to the listed one
            T_new, T_old = self.pair_list["T_gt"][k], p["T_gt"]
            M = T_new @ np.linalg.inv(T_old)
            p["xyz1"] = (p["xyz1"].astype(np.float64) @ M[:3, :3].T + M[:3, 3]).astype(np.float32)
            p["T_gt"] = T_new
        return p

    def get_dev(self, k, device):
        """The same recipe drawn on the device (synth.make_pair_dev: milliseconds instead of a quarter second per 30k-point pair;
        same distributions, different numbers than get()): dict of float32 device tensors + T_gt."""
        rho = self.rho
        T_gt = None
        if self.pair_list is not None:
            if self.pair_list["overlap"] is not None:
                rho = float(np.clip(self.pair_list["overlap"][k], 0.05, 0.95))
            T_gt = self.pair_list["T_gt"][k]
        return synth.make_pair_dev(N=self.n, rho=rho * self.rho_scale, s=self.s, seed=self.seed + int(k), device=device, T_gt=T_gt, noise=self.noise)


def _upload(p, device):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device, non_blocking=True)
    return dict(xyz0=t(p["xyz0"]), xyz1=t(p["xyz1"]), feats0=t(p["feats0"]), feats1=t(p["feats1"]), T_gt=p["T_gt"])


class CacheSource:
    """Real clouds + FCGF features from a feature cache (io_lists.save_cloud) for the pairs of a balanced list."""

    def __init__(self, pair_list, cache_dir):
        self.pair_list, self.cache_dir = pair_list, cache_dir

    def __len__(self):
        return len(self.pair_list["session"])

    def ids(self, k):
        return int(self.pair_list["session"][k]), int(self.pair_list["src"][k]), int(self.pair_list["tgt"][k])

    def get(self, k):
        s, i, j = self.ids(k)
        xyz0, f0 = io_lists.load_cloud(self.cache_dir, s, i)
        xyz1, f1 = io_lists.load_cloud(self.cache_dir, s, j)
        return dict(xyz0=xyz0, xyz1=xyz1, feats0=f0, feats1=f1, T_gt=self.pair_list["T_gt"][k])

    def get_dev(self, k, device):
        return _upload(self.get(k), device)

    # host half / device half of get_dev: eval_pairs reads the files of the NEXT window on worker threads while this one registers
    load_host = get

    def finish(self, host, device):
        return _upload(host, device)


class RefCloudSource:
    """Pairs of a balanced list from the REFERENCE's cloud cache (io_lists.load_ref_cloud: raw scans, [N,3] float64): every
    scan is voxel-deduplicated on the GPU exactly as the reference's loader does before the network sees it
    (voxel.voxel_downsample == ME.utils.sparse_quantize(xyz / 0.3, return_index=True), generic_balanced_loader.py:62-66).
    FCGF itself is out of scope, so the 32-d descriptors of the kept points come from a feature cache written for the same
    voxelisation (io_lists.save_cloud(feat_dir, session, idx, xyz[sel], feats)); a missing or mismatching entry is an error."""

    def __init__(self, pair_list, cloud_dir, feat_dir, voxel_size=0.3):
        self.pair_list, self.cloud_dir, self.feat_dir, self.voxel_size = pair_list, cloud_dir, feat_dir, voxel_size

    def __len__(self):
        return len(self.pair_list["session"])

    def ids(self, k):
        return int(self.pair_list["session"][k]), int(self.pair_list["src"][k]), int(self.pair_list["tgt"][k])

    def _files(self, s, i):
        return io_lists.load_ref_cloud(self.cloud_dir, s, i), io_lists.load_cloud(self.feat_dir, s, i)

    def _cloud(self, s, i, files=None):
        from . import voxel
        raw, (cx, feats) = files if files is not None else self._files(s, i)
        xyz, sel = voxel.voxel_downsample(raw, self.voxel_size)
        xyz = xyz.cpu().numpy()
        if feats.shape[0] != xyz.shape[0] or not np.allclose(cx, xyz, atol=1e-4):
            raise ValueError(f"feature cache entry {s}_{i} was not computed for this voxelisation ({feats.shape[0]} vs {xyz.shape[0]} points)")
        return xyz, feats

    def get(self, k, files=None):
        s, i, j = self.ids(k)
        xyz0, f0 = self._cloud(s, i, files and files[0])
        xyz1, f1 = self._cloud(s, j, files and files[1])
        return dict(xyz0=xyz0, xyz1=xyz1, feats0=f0, feats1=f1, T_gt=self.pair_list["T_gt"][k])

    def get_dev(self, k, device):
        return _upload(self.get(k), device)

    def load_host(self, k):          # the file reads (worker threads); the voxel de-duplication stays on the caller's thread and device
        s, i, j = self.ids(k)
        return k, (self._files(s, i), self._files(s, j))

    def finish(self, host, device):
        return _upload(self.get(host[0], host[1]), device)


def load_list_fixture(dataset):
    """The reference's balanced test list of `dataset` ("A" | "B") from tests/golden/lists (made by tests/golden/make_lists.py from
    balanced_sets/<set>/test.txt) in the dict form of io_lists.read_pair_list."""
    import os
    name = io_lists.DATASET_NAMES[dataset]
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lists", f"{name}_test.npz")
    z = np.load(path)
    return dict(session=z["session"].astype(np.int64), src=z["src"].astype(np.int64), tgt=z["tgt"].astype(np.int64),
                T_gt=z["T_gt"].reshape(-1, 4, 4).astype(np.float64), overlap=z["overlap"].astype(np.float64))


def second_nn_share(n, device, reps=5):
    """What FR.py:117 adds to the billed time: find_nn with the second neighbour minus find_nn without (matching.py:12-18), as a
    SHARE of the former -- measured on one synthetic pair of this size (FR.second_nn_share), applied by the callers to the
    forward-NN time of the calls they really made."""
    p = synth.make_pair_dev(N=n, seed=7, device=device)
    ws = _ext.Workspace(n, n, 32, 1)
    with torch.cuda.device(device):
        share = fr.second_nn_share(p["feats0"], p["feats1"], ws, reps=reps)
    ws.close()
    return share


# The "hard" list-driven surrogate: a fraction of the listed overlap, noisier descriptors and coordinates -- chosen (tools/hard_explore.py)
# so that the pipeline's recall@(5 deg, 0.6 m) sits near 90 % on both lists: a setting on which a loss of accuracy can show
# (the plain surrogate is recovered on every row).  tests/test_gpu_lists.py holds the oracle pipeline to the same rows, flag by flag.
HARD = {"A": dict(rho_scale=0.45, s=1.7, noise=0.15), "B": dict(rho_scale=0.3, s=1.8, noise=0.2)}


def eval_list_batched(pair_list, indices, args, n=30000, s=1.2, batch=32, nstreams=2, resident=256, device=None, seed=51, verbose=False,
                      rho_scale=1.0, noise=0.05):
    """The list-driven synthetic surrogate for the rows `indices` of a balanced list, registered the way bench.py
    registers its pairs: `resident` pairs are synthesised on the device (row k: its ground-truth motion, overlap -> rho), then
    registered by batched calls (`batch` pairs per lr_register_batch, round-robin over `nstreams` streams / workspaces) inside
    a timed region that contains nothing else; repeat.  Returns dict(T [P,4,4], re_deg, te_m, n_corr, n_ids, seconds = the timed
    regions' sum, stage_ms = per-pair means of [whole call, forward NN, forward filter, reverse filter, RANSAC gen+score] from the
    library's own events, second_nn_share)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    params = registration_params(args)
    P = len(indices)
    batch = max(1, min(batch, 64))
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    wss = [_ext.Workspace(n, n, 32, params.ransac.iters, max_pairs=batch) for _ in range(nstreams)]
    size = ctypes.sizeof(_ext.PairResult)
    Ts = np.tile(np.eye(4), (P, 1, 1)); re = np.zeros(P); te = np.zeros(P); n_corr = np.zeros(P, np.int64); n_ids = np.zeros(P, np.int64)
    status = np.zeros(P, np.int64)
    seconds = 0.0
    for lo in range(0, P, resident):
        rows = indices[lo:lo + resident]
        pairs, gts = [], []
        for k in rows:
            rho = float(np.clip(pair_list["overlap"][k], 0.05, 0.95)) * rho_scale
            p = synth.make_pair_dev(N=n, rho=rho, s=s, seed=seed + int(k), device=dev, T_gt=pair_list["T_gt"][k], noise=noise)
            pairs.append((p["xyz0"], p["xyz1"], p["feats0"], p["feats1"])); gts.append(p["T_gt"])
        outs = torch.zeros((len(rows), size), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))
        t0 = time.perf_counter()
        for c, b0 in enumerate(range(0, len(rows), batch)):
            sidx = c % nstreams
            chunk = pairs[b0:b0 + batch]
            fr.register_batch_dev(chunk, params, out=outs[b0:b0 + len(chunk)], ws=wss[sidx], stream=streams[sidx].cuda_stream)
        for st in streams:
            torch.cuda.current_stream(dev).wait_stream(st)
        host = outs.cpu()                              # the step's product on the host (the D2H copy is inside the timed region)
        seconds += time.perf_counter() - t0
        hb = host.numpy()
        for j, k in enumerate(rows):
            r = _ext.PairResult.from_buffer_copy(hb[j].tobytes())
            T = np.array(r.T[:], np.float64).reshape(4, 4) if r.status == 0 else np.eye(4)
            Ts[lo + j] = T
            re[lo + j] = metrics.rotation_error_deg(T, gts[j]); te[lo + j] = metrics.translation_error_cm(T, gts[j]) / 100.0
            n_corr[lo + j] = r.n_corr; n_ids[lo + j] = r.ransac.n_ids; status[lo + j] = r.status
        if verbose:
            print(f"{time.strftime('%m/%d %H:%M:%S')} {lo + len(rows)}/{P} pairs, {(lo + len(rows)) / max(seconds, 1e-9):.0f} pairs/s", flush=True)
        del pairs, outs
    # stage times: a separate, serialised pass over a sample of the rows (one timed call at a time per workspace, events read after
    # each call) -- the throughput loop above keeps several calls in flight and cannot attribute time to stages
    sample = indices[:min(P, 4 * batch)]
    stage = np.zeros(6); timed_pairs = 0
    pairs = []
    for k in sample:
        rho = float(np.clip(pair_list["overlap"][k], 0.05, 0.95)) * rho_scale
        p = synth.make_pair_dev(N=n, rho=rho, s=s, seed=seed + int(k), device=dev, T_gt=pair_list["T_gt"][k], noise=noise)
        pairs.append((p["xyz0"], p["xyz1"], p["feats0"], p["feats1"]))
    outs = torch.zeros((len(sample), size), dtype=torch.uint8, device=dev)
    for rep in range(2):                               # first pass warms up, second is read
        wss[0].timing(True)
        for b0 in range(0, len(sample), batch):
            chunk = pairs[b0:b0 + batch]
            fr.register_batch_dev(chunk, params, out=outs[b0:b0 + len(chunk)], ws=wss[0], stream=streams[0].cuda_stream)
            streams[0].synchronize()
            if rep == 1:
                ms, _ = wss[0].stage_times()
                stage = np.array(ms); timed_pairs = b0 + len(chunk)
    share = second_nn_share(n, dev)
    for w in wss:
        w.close()
    return dict(T=Ts, re_deg=re, te_m=te, n_corr=n_corr, n_ids=n_ids, status=status, seconds=seconds,
                stage_ms_per_pair=(stage / max(timed_pairs, 1)).tolist(), stage_sample_pairs=int(timed_pairs), second_nn_share=share)


def registration_params(args):
    """lr_pair_params of the timed registration call.  The harness times ICP on its own (test.py:183-193, stats column 11),
    so the fused ICP stage of lr_register_pair stays off here: column 9 must not contain an ICP."""
    params = fr.pair_params(args)
    params.icp = 0
    return params


def inlier_ratios_dev(xyz0, xyz1, nn1, c0, c1, n0, n_corr, T_gt):
    """measure_inlier_ratio (matching.py:241-249) of a batch of pairs on the device, in float64 with a fixed elementwise
    operation order (so a pair's value does not depend on the batch it was in): returns (ratio over the n0[k] NN pairs
    (i, nn1[k, i]), ratio over the first n_corr[k] filtered pairs (c0[k, c], c1[k, c])) as two [B] float64 device tensors.
    xyz0 / xyz1: lists of [n,3] float32 device tensors; nn1 / c0 / c1: [B, W] int32; n0 / n_corr: [B] device int tensors;
    T_gt: [B,4,4] float64 on the host."""
    B, W = nn1.shape
    dev = nn1.device
    thr2 = (2 * fr.VOXEL_SIZE) ** 2

    def padded(xs):
        w = max(int(x.shape[0]) for x in xs)
        if all(int(x.shape[0]) == w for x in xs):
            return torch.stack(xs).to(torch.float64)
        out = torch.zeros((len(xs), w, 3), dtype=torch.float64, device=dev)
        for k, x in enumerate(xs):
            out[k, :x.shape[0]] = x
        return out

    X0, X1 = padded(xyz0), padded(xyz1)
    T = torch.from_numpy(np.ascontiguousarray(T_gt, np.float64)).to(dev)
    # R p + t, component by component (no BLAS: one fixed order of operations)
    P = [X0[..., 0] * T[:, r, 0, None] + X0[..., 1] * T[:, r, 1, None] + X0[..., 2] * T[:, r, 2, None] + T[:, r, 3, None] for r in range(3)]
    ar = torch.arange(W, device=dev)[None, :]

    def ratio(i0, i1, count):
        i0 = i0.long().clamp(0, X0.shape[1] - 1); i1 = i1.long().clamp(0, X1.shape[1] - 1)
        d2 = torch.zeros(i0.shape, dtype=torch.float64, device=dev)
        for r in range(3):
            d = torch.gather(P[r], 1, i0) - torch.gather(X1[..., r], 1, i1)
            d2 += d * d
        live = ar < count[:, None]
        hits = ((d2 < thr2) & live).sum(1).to(torch.float64)
        return torch.where(count > 0, hits / count.clamp(min=1).to(torch.float64), torch.zeros_like(hits))

    init = ratio(ar.expand(B, W), nn1, n0)
    filt = ratio(c0, c1, n_corr)
    return init, filt


_NCORR_OFF = _ext.PairResult.n_corr.offset


def eval_pairs(source, indices, args, device=None, batch=32, in_flight=6, nstreams=3, verbose=False, workers=8):
    """Register source[k] for k in indices through the BATCHED engine -- what Experiments/test.py:108-234 does pair by pair.
    The list is taken in windows of `in_flight` batches of `batch` rows; per window:
      A  data: source.get_dev for every row (device synthesis / upload), synchronised                       -> column 10
         (a source with load_host / finish -- the file-backed ones -- has the files of the NEXT window read by `workers` threads while
         this window registers; column 10 is then the wait for them + the device half)
      B  registration: one lr_register_batch per batch (every kernel of the path launched once for all its pairs), batch i on
         workspace i and stream i % nstreams, + the strided copies of the lists; nothing else is in the region; synchronised;
         its wall time is the window's registration time                                                        -> column 9
      C  ICP (args.icp): one lr_icp_batch per batch on the same workspaces, timed the same way (test.py:183-193) -> column 11
      D  statistics: ground-truth inlier ratios on the device, RE / TE on the host                               -> the rest
    Returns (stats [n,22] float64, T [n,4,4] float64) in the reference's 22-column layout (test.py:98-100).  Time columns are
    ATTRIBUTED shares of a window (attribute_window_time): the window's registration wall time goes to its calls by their device
    time and to a call's pairs by n0 x n1 (NN stages) and ids examined x correspondences (the rest); column 9 takes off the first
    neighbour's part of the forward NN, which the reference treats as given (matching.py:7-11; the second neighbour's share is
    calibrated once per cloud-size class, FR.second_nn_share); the whole path is kept in LAST_WHOLE_PATH.  LAST_RUN holds the run's totals (seconds in A, B, C, D; pairs).
    Results are bit-identical to the one-pair-at-a-time path (eval_pairs_serial; tests/test_gpu_cli.py)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    params = registration_params(args)
    n = len(indices)
    stats = np.full((n, 22), np.nan)
    Ts = np.tile(np.eye(4), (n, 1, 1))
    whole_path = np.full(n, np.nan)
    batch = max(1, min(int(batch), 64))
    in_flight = max(1, int(in_flight))
    nstreams = max(1, min(int(nstreams), in_flight))
    streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
    wss = [None] * in_flight
    seen = [None] * in_flight
    size = ctypes.sizeof(_ext.PairResult)
    want_icp = bool(getattr(args, "icp", False))
    lib = _ext.lib()
    totals = dict(data_s=0.0, registration_s=0.0, icp_s=0.0, stats_s=0.0, pairs=n, batch=batch, in_flight=in_flight, nstreams=nstreams)
    window = batch * in_flight
    cur = torch.cuda.current_stream(dev)
    pool = None
    if workers > 1 and hasattr(source, "load_host"):
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=int(workers))
    ahead = [pool.submit(source.load_host, indices[row]) for row in range(0, min(window, n))] if pool else None
    # Whatever ends the loop -- a file that cannot be read, a failing library call, out of memory, Ctrl-C -- the readers of the next
    # window are stopped and every workspace is released on the way out (they would otherwise keep reading files and holding device
    # memory until the interpreter exits)
    try:
        for w0 in range(0, n, window):
            rows_w = list(range(w0, min(w0 + window, n)))
            # ---- A: data
            t0 = time.time()
            if pool:
                mine, ahead = ahead, [pool.submit(source.load_host, indices[row]) for row in range(w0 + window, min(w0 + 2 * window, n))]
                ps = [source.finish(f.result(), dev) for f in mine]      # (a missing / mismatching file raises here: the finally below stops the readers)
            else:
                ps = [source.get_dev(indices[row], dev) for row in rows_w]
            torch.cuda.synchronize(dev)
            t_data = time.time() - t0
            totals["data_s"] += t_data
            groups = []
            for i, g0 in enumerate(range(0, len(rows_w), batch)):
                sl = slice(g0, min(g0 + batch, len(rows_w)))
                gp = ps[sl]
                n0s = [int(p["feats0"].shape[0]) for p in gp]; n1s = [int(p["feats1"].shape[0]) for p in gp]
                d = int(gp[0]["feats0"].shape[1])
                if wss[i] is None or not wss[i].fits(max(n0s), max(n1s), params.ransac.iters):
                    if wss[i] is not None:
                        wss[i].close()
                    ragged = len(set(n0s + n1s)) > 1          # real data: leave headroom for the next windows
                    wss[i] = _ext.Workspace(int(max(n0s) * (1.25 if ragged else 1)), int(max(n1s) * (1.25 if ragged else 1)), d, params.ransac.iters, max_pairs=batch)
                    seen[i] = None
                k_big = int(np.argmax(n0s))
                fresh = fr.share_key(n0s[k_big], n1s[k_big], dev) not in fr._SHARE
                share = fr.second_nn_share(gp[k_big]["feats0"], gp[k_big]["feats1"], wss[i])      # (the GPU is idle here: calibrated once per size class)
                if fresh or seen[i] is None:
                    torch.cuda.synchronize(dev)
                    wss[i].timing(True); seen[i] = [0.0] * 6
                groups.append(dict(rows=rows_w[sl], ps=gp, n0=n0s, n1=n1s, share=share, slot=i,
                                   out=torch.empty((len(gp), size), dtype=torch.uint8, device=dev)))
            torch.cuda.synchronize(dev)
            for st in streams:
                st.wait_stream(cur)
            # ---- B: registration
            t0 = time.time()
            for g in groups:
                st = streams[g["slot"] % nstreams]
                chunk = [(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"]) for p in g["ps"]]
                fr.register_batch_dev(chunk, params, out=g["out"], ws=wss[g["slot"]], stream=st.cuda_stream)
                W = max(g["n0"]); B = len(chunk)
                with torch.cuda.stream(st):
                    g["nn1"] = torch.empty((B, W), dtype=torch.int32, device=dev); g["c0"] = torch.empty_like(g["nn1"]); g["c1"] = torch.empty_like(g["nn1"])
                _ext.check(lib.lr_workspace_lists_batch(wss[g["slot"]].handle, B, W, g["nn1"].data_ptr(), None, g["c0"].data_ptr(), g["c1"].data_ptr(), st.cuda_stream))
            for st in streams:
                st.synchronize()
            t_reg = time.time() - t0
            totals["registration_s"] += t_reg
            # ---- C: ICP
            t_icp = 0.0
            if want_icp:
                t0 = time.time()
                for g in groups:
                    _ext.check(lib.lr_icp_batch(wss[g["slot"]].handle, 2 * fr.VOXEL_SIZE, 30, 1e-6, 1e-6, g["out"].data_ptr(), streams[g["slot"] % nstreams].cuda_stream))
                for st in streams:
                    st.synchronize()
                t_icp = time.time() - t0
                totals["icp_s"] += t_icp
            # ---- D: statistics
            t0 = time.time()
            # The window's registration wall time is attributed to its pairs by what each of them cost (it used to be a
            # flat share): a call gets the part of the window its own device time (the library's events) is of all calls' device times;
            # inside a call the NN stages (forward + reverse: events) go to the pairs by n0 x n1, the rest -- filter, RANSAC, refit --
            # by hypothesis ids examined x correspondences (what the scoring passes do).  Column 9 then takes off the first neighbour's
            # part of the pair's forward NN, as FR.py:117 does.  Still an attribution, not a measurement: --serial True measures.
            d_call = 0.0
            for g in groups:
                ms, _ = wss[g["slot"]].stage_times()
                g["call_ms"], g["fwd_ms"], g["rev_ms"] = ms[0] - seen[g["slot"]][0], ms[1] - seen[g["slot"]][1], ms[5] - seen[g["slot"]][5]
                d_call += g["call_ms"]
                seen[g["slot"]] = ms
            for g in groups:
                gp = g["ps"]
                with torch.cuda.stream(streams[g["slot"] % nstreams]):
                    n_corr = g["out"][:, _NCORR_OFF:_NCORR_OFF + 4].contiguous().view(torch.int32).view(-1)
                    T_gt = np.stack([np.asarray(p["T_gt"], np.float64) for p in gp])
                    ri, rf = inlier_ratios_dev([p["xyz0"] for p in gp], [p["xyz1"] for p in gp], g["nn1"], g["c0"], g["c1"],
                                               torch.tensor(g["n0"], dtype=torch.int32, device=dev), n_corr, T_gt)
                    hb = g["out"].cpu().numpy(); ri = ri.cpu().numpy(); rf = rf.cpu().numpy()
                res_g = [_ext.PairResult.from_buffer_copy(hb[j].tobytes()) for j in range(len(g["rows"]))]
                whole_g, billed_g = attribute_window_time(t_reg, g["call_ms"], d_call, g["fwd_ms"], g["rev_ms"], g["share"], g["n0"], g["n1"],
                                                          [r.ransac.n_ids for r in res_g], [r.n_corr for r in res_g])
                for j, row in enumerate(g["rows"]):
                    r = res_g[j]
                    T = np.array(r.T[:], np.float64).reshape(4, 4) if r.status == 0 else np.eye(4)
                    gt = T_gt[j]
                    re, te = metrics.rotation_error_deg(T, gt), metrics.translation_error_cm(T, gt)
                    sess, si, ti = source.ids(indices[row])
                    stats[row, 0] = float(re < metrics.RE_THRE_DEG and te < metrics.TE_THRE_CM)
                    stats[row, 1], stats[row, 2] = re, te
                    stats[row, 9] = billed_g[j]
                    whole_path[row] = whole_g[j]
                    stats[row, 10], stats[row, 11] = t_data / len(rows_w), 0.0
                    if want_icp:
                        T_icp = np.array(r.T_icp[:], np.float64).reshape(4, 4) if r.status == 0 else np.eye(4)
                        re_i, te_i = metrics.rotation_error_deg(T_icp, gt), metrics.translation_error_cm(T_icp, gt)
                        stats[row, 11] = t_icp / len(rows_w)
                        stats[row, 12] = float(re_i < metrics.RE_THRE_DEG and te_i < metrics.TE_THRE_CM)
                        stats[row, 13], stats[row, 14] = re_i, te_i
                    stats[row, 15] = g["n0"][j]
                    stats[row, 16] = ri[j]
                    stats[row, 17] = int(r.n_corr)
                    stats[row, 18] = rf[j]
                    stats[row, 19], stats[row, 20], stats[row, 21] = sess, si, ti
                    Ts[row] = T
            totals["stats_s"] += time.time() - t0
            if verbose:
                print(f"{time.strftime('%m/%d %H:%M:%S')} Finished pair:{rows_w[-1]}/{n}  ({len(rows_w) / max(t_reg, 1e-9):.0f} pairs/s in the registration region)", flush=True)
            del ps, groups
        torch.cuda.synchronize(dev)
    finally:
        if pool:
            for f in (ahead or []):
                f.cancel()
            pool.shutdown(wait=True, cancel_futures=True)
        for w in wss:
            if w is not None:
                w.close()
    global LAST_WHOLE_PATH, LAST_RUN
    LAST_WHOLE_PATH = whole_path
    LAST_RUN = totals
    return stats, Ts


def attribute_window_time(t_window, call_ms, all_calls_ms, fwd_ms, rev_ms, share, n0, n1, n_ids, n_corr):
    """Per-pair (whole path seconds, seconds billed the reference's way) of ONE batched call inside a window of `t_window` seconds:
    the call's part of the window = call_ms / all_calls_ms; its NN stages (fwd_ms + rev_ms of call_ms) are spread over its pairs by
    n0 x n1, the rest by n_ids x n_corr (+1: a pair with no correspondences still costs its launches); billed = whole - the first
    neighbour's part of the pair's forward NN (1 - share of it, FR.py:117 / matching.py:7-11).  The pairs' whole-path times add up to the
    call's part of the window."""
    n0 = np.asarray(n0, np.float64); n1 = np.asarray(n1, np.float64)
    t_call = t_window * (call_ms / all_calls_ms) if all_calls_ms > 0 else t_window / max(len(n0), 1)
    f_fwd = min(max(fwd_ms / call_ms, 0.0), 1.0) if call_ms > 0 else 0.0
    f_nn = min(max((fwd_ms + rev_ms) / call_ms, f_fwd), 1.0) if call_ms > 0 else 0.0
    w_nn = n0 * n1; w_nn = w_nn / w_nn.sum()
    w_rest = np.asarray(n_ids, np.float64) * np.asarray(n_corr, np.float64) + 1.0; w_rest = w_rest / w_rest.sum()
    whole = t_call * (f_nn * w_nn + (1.0 - f_nn) * w_rest)
    billed = whole - t_call * f_fwd * w_nn * (1.0 - share)
    return whole, np.maximum(billed, 0.0)


def stats_columns(serial):
    """What the 22 columns of raw_stats.npy hold (layout of the reference's Experiments/test.py:96-100), written next to the file: the
    time columns of the batched engine are per-pair SHARES of a window, not per-pair measurements."""
    t = ("measured per pair (device events of the call, billed as FR.py:117)" if serial else
         "ATTRIBUTED, not measured: the registration wall time of a window of batched calls, split over its calls by their device time (library events) and "
         "over a call's pairs by n0 x n1 (NN stages) and hypothesis ids examined x correspondences (the rest), minus the first neighbour's part of the forward NN "
         "(FR.py:117); the rows of a window add up to its wall time; depends on --batch / --in_flight; run with --serial True for per-pair measurements")
    names = ["success", "RE (deg)", "TE (cm)", "(unused here) input inlier number", "(unused) input inlier ratio", "(unused) output inlier number",
             "(unused) output inlier precision", "(unused) output inlier recall", "(unused) output inlier F1", "model_time / reg time (s): " + t,
             "data_time (s)" + ("" if serial else ": window share"), "icp_time (s)" + ("" if serial else ": window share"), "recall_icp", "RE_icp (deg)", "TE_icp (cm)",
             "num_pairs_init", "inlier_ratio_init", "num_pairs_filtered", "inlier_ratio_filtered", "drive / session", "t0 / source index", "t1 / target index"]
    return "".join(f"{k}: {v}\n" for k, v in enumerate(names))


def eval_pairs_serial(source, indices, args, device=None, in_flight=4, verbose=False):
    """The one-pair-at-a-time call pattern of the reference harness (one lr_register_pair per list row, `in_flight` pairs on
    separate streams): kept as the cross-check of eval_pairs (same data source, same statistics code) and for latency
    measurements.  Returns (stats [n,22] float64, T [n,4,4] float64)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    params = registration_params(args)
    n = len(indices)
    stats = np.full((n, 22), np.nan)
    Ts = np.tile(np.eye(4), (n, 1, 1))
    streams = [torch.cuda.Stream(device=dev) for _ in range(in_flight)]
    wss = [None] * in_flight
    slots = [None] * in_flight          # (row, result buffer, events, host data) of the pair in flight on each stream
    whole_path = np.full(n, np.nan)     # device time of the whole call per pair (incl. the forward NN, which column 9 does not bill)
    seen = [None] * in_flight           # stage-time sums of each workspace at its last read

    def retire(s):
        row, out, ev0, ev1, p, t_data, n0, icp_buf, ev2 = slots[s]
        p["_end"].synchronize()
        r = fr.read_result(out)
        T = np.array(r.T[:], np.float64).reshape(4, 4) if r.status == 0 else np.eye(4)
        ri, rf = (float(v.cpu()[0]) for v in p["_ratios"])
        re, te = metrics.rotation_error_deg(T, p["T_gt"]), metrics.translation_error_cm(T, p["T_gt"])
        sess, si, ti = source.ids(indices[row])
        stats[row, 0] = float(re < metrics.RE_THRE_DEG and te < metrics.TE_THRE_CM)
        stats[row, 1], stats[row, 2] = re, te
        # column 9 = what FR.py:117 bills: filter + RANSAC (+ refit) + the second neighbour's surcharge, from the library's own stage
        # events of this call; the whole device path (forward NN included) is kept next to it (LAST_WHOLE_PATH)
        ms, _ = wss[s].stage_times()
        d_call, d_fwd = ms[0] - seen[s][0], ms[1] - seen[s][1]
        seen[s] = ms
        whole_path[row] = ev0.elapsed_time(ev1) * 1e-3
        stats[row, 9] = max(whole_path[row] - d_fwd * 1e-3 * (1.0 - p["_share"]), 0.0)
        stats[row, 10], stats[row, 11] = t_data, 0.0
        if icp_buf is not None:                                # stats columns 11-14 = the harness' ICP stage (test.py:183-193)
            T_icp = icp_buf[0].cpu().numpy().reshape(4, 4) if r.status == 0 else np.eye(4)
            re_i, te_i = metrics.rotation_error_deg(T_icp, p["T_gt"]), metrics.translation_error_cm(T_icp, p["T_gt"])
            stats[row, 11] = ev1.elapsed_time(ev2) * 1e-3
            stats[row, 12] = float(re_i < metrics.RE_THRE_DEG and te_i < metrics.TE_THRE_CM)
            stats[row, 13], stats[row, 14] = re_i, te_i
        stats[row, 15] = n0
        stats[row, 16] = ri
        stats[row, 17] = int(r.n_corr)
        stats[row, 18] = rf
        stats[row, 19], stats[row, 20], stats[row, 21] = sess, si, ti
        Ts[row] = T
        slots[s] = None
        if verbose:
            print(f"{time.strftime('%m/%d %H:%M:%S')} Finished pair:{row}/{n}", flush=True)

    for row, k in enumerate(indices):
        s = row % in_flight
        if slots[s] is not None:
            retire(s)
        t0 = time.time()
        with torch.cuda.stream(streams[s]):
            p = source.get_dev(k, dev)
        t_data = time.time() - t0
        n0, n1, d = p["feats0"].shape[0], p["feats1"].shape[0], p["feats0"].shape[1]
        if wss[s] is None or not wss[s].fits(n0, n1, params.ransac.iters):
            if wss[s] is not None:
                torch.cuda.synchronize(dev)
                wss[s].close()
            wss[s] = _ext.Workspace(int(n0 * 1.25), int(n1 * 1.25), d, params.ransac.iters)
            seen[s] = None
        with torch.cuda.stream(streams[s]):
            x0, x1, f0, f1 = p["xyz0"], p["xyz1"], p["feats0"], p["feats1"]
            # the second neighbour's share of the forward NN for clouds of this size (measured once per size class with a few
            # timed NN calls on this workspace, which is idle here); the stage events are (re)armed afterwards
            fresh = fr.share_key(n0, n1, dev) not in fr._SHARE
            if fresh:
                torch.cuda.synchronize(dev)               # calibrate on an idle GPU
            share = fr.second_nn_share(f0, f1, wss[s])
            if fresh or seen[s] is None:
                wss[s].timing(True); seen[s] = [0.0] * 6
            out = torch.empty(ctypes.sizeof(_ext.PairResult), dtype=torch.uint8, device=dev)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(streams[s])
            fr.register_pair_dev(x0, x1, f0, f1, params, out=out, ws=wss[s], stream=streams[s].cuda_stream)
            ev1.record(streams[s])
            icp_buf, ev2 = None, None
            if getattr(args, "icp", False):
                # ICP refinement of the registration result, timed on its own like the reference's icp_timer
                T_icp = torch.empty(16, dtype=torch.float64, device=dev)
                res_icp = torch.empty(ctypes.sizeof(_ext.IcpResult), dtype=torch.uint8, device=dev)
                _ext.check(_ext.lib().lr_icp(wss[s].handle, x0.data_ptr(), n0, x1.data_ptr(), n1, out.data_ptr(), 2 * fr.VOXEL_SIZE, 30,
                                              1e-6, 1e-6, T_icp.data_ptr(), res_icp.data_ptr(), streams[s].cuda_stream))
                ev2 = torch.cuda.Event(enable_timing=True)
                ev2.record(streams[s])
                icp_buf = (T_icp, res_icp)
            nn1 = torch.empty((1, n0), dtype=torch.int32, device=dev); c0 = torch.empty_like(nn1); c1 = torch.empty_like(nn1)
            _ext.check(_ext.lib().lr_workspace_lists_at(wss[s].handle, 0, n0, nn1.data_ptr(), None, c0.data_ptr(), c1.data_ptr(), streams[s].cuda_stream))
            n_corr = out[_NCORR_OFF:_NCORR_OFF + 4].view(torch.int32)
            ratios = inlier_ratios_dev([x0], [x1], nn1, c0, c1, torch.tensor([n0], dtype=torch.int32, device=dev), n_corr,
                                       np.asarray(p["T_gt"], np.float64)[None])
            ev3 = torch.cuda.Event(); ev3.record(streams[s])
        slots[s] = (row, out, ev0, ev1, dict(p, _keep=(x0, x1, f0, f1), _share=share, _ratios=ratios, _end=ev3), t_data, n0, icp_buf, ev2)
    for s in range(in_flight):
        if slots[s] is not None:
            retire(s)
    for w in wss:
        if w is not None:
            w.close()
    global LAST_WHOLE_PATH
    LAST_WHOLE_PATH = whole_path
    return stats, Ts


LAST_WHOLE_PATH = None      # eval_pairs: whole-call device seconds per row of its last run
LAST_RUN = None             # eval_pairs: totals of its last run (seconds in the data / registration / ICP / statistics phases, pairs)
