"""Registration-pairs/sec benchmark (BASELINE.json config #2: 30k-pt x 32-d synthetic FCGF pairs,
mutual-NN + 50k-iteration RANSAC + LS refit) on N GPUs of one node.

    python bench.py [--gpus N --steps K --warmup W]            # N > 1 without a launcher: spawns one process per GPU itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path over `--pairs` DISTINCT resident pairs per GPU (inputs in HBM before the timed region),
registered as pair-batched launches (lr_register_batch, `--batch` pairs per call, round-robin over `--streams` streams), the
4x4 transforms copied back to pinned host memory inside the timed region.  Pairs shard over ranks with no data-path
collective (weak scaling); one RCCL all_gather per step returns the result rows.  Prints ONE JSON line on rank 0 with
`roofline` for the dominant kernel (nn16_passb_kernel, the f16 matrix-core filter pass), `pair_roofline` (whole pair against
the blended matrix / vector floor) and `cpu_baseline` (the reference's torch-einsum NN + numpy filter + OpenMP RANSAC on the
host cores, rank 0, N=1).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3        # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
MFMA_F16_PEAK_TFLOPS = 2500.0       # same table, "Peak BF16/FP16 MFMA ~2.5 PF dense"
VALU_F32_PEAK_TFLOPS = 157.3        # same table, "Peak FP32 (vector)"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=192, help="distinct resident pairs per step per GPU")
    ap.add_argument("--n", "--points", dest="n", type=int, default=30000, help="points per cloud")
    ap.add_argument("--iters", type=int, default=50000)
    ap.add_argument("--mode", default="MNN")
    ap.add_argument("--codebase", default="open3D", choices=["open3D", "GC"],
                    help="open3D (default): 3-point samples, edge-length check, EVERY one of --iters hypotheses scored, refit over the NN pairs "
                         "(the headline workload); GC: the reference CLI's defaults (test.py:301-313: PROSAC, ELC, MSAC, confidence 0.999 "
                         "early exit, local optimisation + final least squares) -- an additional, lighter workload, never the headline")
    ap.add_argument("--batch", type=int, default=0, help="pairs per batched call (0: 32, or 8 for clouds above 60k points)")
    ap.add_argument("--streams", type=int, default=0, help="batched calls in flight per GPU (0: 2 for the headline workload, 3 with --mode GPF, 4 with --codebase GC whose "
                                                            "one-block-per-pair local optimisation leaves most CUs to the other calls; 6 with --list)")
    ap.add_argument("--sustain-s", type=float, default=None, help="after the K timed steps, run the same step loop for at least this many seconds and report it as "
                                                                  "`sustained` (outside `value`; 0: skip) -- the timed region of the contract is a fraction of a second.  "
                                                                  "Default 10, or 0 under rocprofv3 (ROCPROF* in the environment): ~600 untimed steps would otherwise dominate "
                                                                  "every counter pass and kernel summary")
    ap.add_argument("--include-h2d", action="store_true", help="copy each pair from pinned host memory inside the timed region (PCIe-inclusive rate; not the headline value)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=8)
    ap.add_argument("--cpu-procs", type=int, default=0, help="worker processes of the process-parallel CPU baseline (0: min(8, host threads // best thread count))")
    ap.add_argument("--cpu-budget-s", type=float, default=45.0, help="stop the CPU baseline sample after this many seconds (at least 2 pairs)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="collective backend of the N > 1 run: nccl (= RCCL over xGMI, the product path) or gloo (host tensors; lets two ranks share "
                         "one GPU so that the launcher, rendezvous, per-rank seeds and max-over-ranks timing can be exercised on a 1-GPU box)")
    ap.add_argument("--devices", type=str, default=None, help="comma-separated device index per rank (default: rank r -> device r); e.g. 0,0 with --dist-backend gloo")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE",
                    help="tuning option for every workspace (lr_workspace_option: nn_blocks, nn_blocks_batch, nn_sample_stride, rev_strips, nn_second_auto); none changes a result")
    ap.add_argument("--list", choices=["A", "B"], default=None,
                    help="instead of the headline workload: the reference's README commands over a FULL balanced test list with the list-driven "
                         "synthetic surrogate (SURVEY 8d; every row's ground-truth motion and overlap, --n points): A = Apollo-Southbay, 7008 rows, "
                         "--mode GPF --iters 50000; B = NuScenes-Boston, 2592 rows, --mode MMN --iters 1000000 --GC_conf 0.9995 (README.md:54-55; "
                         "codebase GC defaults).  Prints its own JSON line (recall, pairs/s, whole-path and reference-style time per pair)")
    ap.add_argument("--list-stride", type=int, default=1, help="with --list: every k-th row only")
    ap.add_argument("--extra-list", choices=["A", "B", "none"], default="A",
                    help="the default run (N = 1, headline workload) also registers every --extra-list-stride-th row of this balanced test list with the reference's "
                         "README command (configs[2] / [3], list-driven surrogate) after everything else and reports it under `extra.list_<X>`: a driver-run "
                         "record of a list workload next to the headline (outside `value`)")
    ap.add_argument("--extra-list-stride", type=int, default=8)
    ap.add_argument("--hard", type=int, default=1, help="with --list: also run the rows under the HARD surrogate settings (harness.HARD: a fraction of the listed overlap, "
                                                        "noisier descriptors / coordinates; recall near 90 %%) and report that recall next to the plain one (0: skip)")
    ap.add_argument("--traffic-key", action="store_true", help="print the key this command's PMC traffic is filed under in profiles/pmc_traffic.json and exit (tools/pmc_traffic.sh)")
    ap.add_argument("--dry-run", action="store_true", help="test hook: no GPU work, gloo collectives, fake result rows (exercises the launcher, the gather and the JSON line on CPU)")
    args = ap.parse_args(argv)
    if args.sustain_s is None:
        args.sustain_s = 0.0 if any(k.startswith("ROCPROF") for k in os.environ) else 10.0
    if any(k.startswith("ROCPROF") for k in os.environ):
        args.extra_list = "none"          # (a kernel summary / counter pass of the headline workload must not contain a list run's kernels)
    return args


# ----------------------------------------------------------------------------- self-launch (no torch.distributed.run)
def spawn_ranks(args):
    """`python bench.py --gpus N` with no rendezvous environment: start one child per GPU (the launch shape of the
    reference's test_parallel.sh:18-24) and wait.  The parent never touches HIP (no torch.cuda call, no exec after init);
    lidarregistration_amd.launch.run_ranks polls ALL children: a rank that dies during init would otherwise leave the parent waiting
    on rank 0 until the collective's timeout; the first non-zero exit ends the run and the remaining ranks are killed."""
    from lidarregistration_amd import launch
    port = launch.free_port()
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    envs = [dict(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)) for r in range(args.gpus)]
    return launch.run_ranks([cmd] * args.gpus, envs)


# ----------------------------------------------------------------------------- helpers
def traffic_key(n, mode, codebase, iters, pairs_per_launch):
    """The workload a set of PMC traffic figures belongs to (profiles/pmc_traffic.json is keyed by it: the counters of the default
    workload say nothing about 100k-point clouds, another filter mode, another estimator or another batch size)."""
    mode = "MNN" if mode == "MMN" else mode
    return f"n={int(n)},mode={mode},codebase={codebase},iters={int(iters)},pairs_per_launch={int(pairs_per_launch)}"


DEFAULT_TRAFFIC_KEY = traffic_key(30000, "MNN", "open3D", 50000, 32)


def lookup_traffic(tdoc, key):
    """(filter-pass HBM bytes per launch, all kernels' HBM bytes per pair, source note) for workload `key`, or (None, None, None) when
    no PMC pass was taken for it.  Files written before round 5 hold one unkeyed workload: the default one."""
    if not tdoc:
        return None, None, None
    w = tdoc.get("workloads", {}).get(key)
    if w is None and "workloads" not in tdoc and key == DEFAULT_TRAFFIC_KEY:
        w = tdoc
    if w is None:
        return None, None, None
    # (the filter pass has two instantiations, both launched; the one the data asks for does the work)
    live = [v for k, v in w.items() if k.startswith("nn16_passb_kernel<true") and isinstance(v, dict)]
    traffic = int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in live) / max(1, sum(v["launches"] for v in live))) if live else \
        w.get("nn16_passb_kernel", {}).get("hbm_bytes_per_launch")
    src = ("NOT measured in this run: profiles/pmc_traffic.json[" + key + "], the builder's separate rocprofv3 --pmc passes of this command ("
           + str(w.get("_meta", {}).get("commit", "commit unrecorded")) + "), per launch")
    return traffic, w.get("_pair", {}).get("hbm_bytes_per_pair"), (None if traffic is None else src)


def transform_errors(T, G):
    re = np.degrees(np.arccos(np.clip((np.trace(T[:3, :3].T @ G[:3, :3]) - 1) / 2, -1, 1)))
    return re, np.linalg.norm(T[:3, 3] - G[:3, 3])


def cpu_baseline(args, seed0):
    """The reference's own way of computing a pair on host cores (oracle/torch_cpu.py: chunked torch.einsum NN with
    nn_max_n = 250, numpy mutual filter, OpenMP RANSAC + refit), bounded sample; plus the scalar-fma oracle port on 2 pairs."""
    import torch
    from lidarregistration_amd import synth
    from oracle import oracle as orc, torch_cpu
    orc.build()
    w = synth.make_pair(N=2000, seed=1)
    torch_cpu.register_pair(w["xyz0"], w["xyz1"], w["feats0"], w["feats1"], mode=args.mode, iters=1000)     # page-in / thread pools: untimed
    # How many threads does the reference's chunked NN want?  Its GEMMs are 250 rows tall (nn_max_n): on a 128-thread host torch's default
    # (all of them) thrashes -- BENCH_r04 timed 5.0 s per NN pass on 128 threads where 8 vCPUs take 1.2 s.  One NN pass of one pair per
    # candidate count, the fastest is used for the sample (and reported as `cores`).
    max_threads = int(torch.get_num_threads())
    sweep = {}
    p0 = synth.make_pair(N=args.n, seed=seed0)
    for nt in sorted({t for t in (8, 16, 32, 64, 128) if t <= max_threads} | {max_threads}):
        torch.set_num_threads(nt)
        t1 = time.perf_counter()
        torch_cpu.find_nn(p0["feats0"], p0["feats1"], return_2nd=True)
        sweep[nt] = round(time.perf_counter() - t1, 3)
        if sum(sweep.values()) > 0.6 * args.cpu_budget_s:
            break
    best_threads = min(sweep, key=sweep.get)
    torch.set_num_threads(best_threads)
    t_reg = 0.0; done = 0; parts = np.zeros(3)
    for k in range(args.cpu_pairs):
        p = synth.make_pair(N=args.n, seed=seed0 + k)
        t1 = time.perf_counter()
        r = torch_cpu.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode=args.mode, iters=args.iters, sample_size=3, seed=51)
        t_reg += time.perf_counter() - t1; done += 1
        parts += [r["t_nn"], r["t_filter"], r["t_ransac"]]
        if t_reg > args.cpu_budget_s and done >= 2:
            break
    t_port = 0.0
    for k in range(2):
        p = synth.make_pair(N=args.n, seed=seed0 + k)
        t1 = time.perf_counter()
        orc.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode=args.mode, iters=args.iters, sample_size=3, seed=51)
        t_port += time.perf_counter() - t1
    model = ""
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    torch.set_num_threads(max_threads)
    # ---- the reference's own sharding on host cores (test_parallel.sh:18-20: one OS process per shard of the pair list): P processes of the
    #      same path, each with `best_threads` threads, started together (oracle/cpu_worker.py); the honest "reference path on this box's
    #      host cores" -- one process at its best thread count leaves most of a 128-thread host idle
    avail = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or max_threads))
    cores = len(avail)          # the hardware threads this process may run on (os.cpu_count() also counts those outside the cgroup's cpuset)
    nproc = args.cpu_procs if args.cpu_procs > 0 else max(1, min(8, cores // max(1, best_threads)))
    pp = None
    if nproc > 1:
      try:
        per = 2          # pairs per worker (measured on the pool's 2 x 64-core hosts: eight 16-thread workers side by side take ~13 s per pair each -- the chunked einsum NN is memory-bound)
        env = dict(os.environ, OMP_NUM_THREADS=str(best_threads), MKL_NUM_THREADS=str(best_threads), HIP_VISIBLE_DEVICES="", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        if nproc * best_threads > cores:      # (oversubscribed on request: spinning thread pools then cost an order of magnitude; measured 1.2 against 21 pairs/s)
            env.update(OMP_WAIT_POLICY="passive", GOMP_SPINCOUNT="0", KMP_BLOCKTIME="0")
        # every worker on its own cores (what numactl / a job scheduler would do): unpinned, the workers' spinning thread pools land on the same
        # cores and the leg runs 3.5x SLOWER than one process (measured on the round's first box: 0.45 against 1.55 pairs/s)
        pin = nproc * best_threads <= cores
        procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", str(best_threads), str(args.n), args.mode, str(args.iters), str(per), str(seed0 + 1000 * (w + 1)),
                                   ",".join(str(c) for c in avail[w * best_threads:(w + 1) * best_threads]) if pin else "-"],
                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT) for w in range(nproc)]
        try:
            import select
            def line_of(p, deadline):
                while time.perf_counter() < deadline:
                    if select.select([p.stdout], [], [], 0.5)[0]:
                        return p.stdout.readline()
                return ""
            t_ready = time.perf_counter() + 240.0
            ready = all(line_of(p, t_ready).strip() == "READY" for p in procs)
            if ready:
                t1 = time.perf_counter()
                for p in procs:
                    p.stdin.write("GO\n"); p.stdin.flush()
                t_done = t1 + 90.0          # (bounded: a leg that takes longer than this says nothing useful and would hold the bench line back)
                outs = [line_of(p, t_done).split() for p in procs]
                wall = time.perf_counter() - t1
                if all(len(o) == 3 and o[0] == "DONE" for o in outs):
                    pp = {"value": round(nproc * per / wall, 4), "processes": nproc, "threads_per_process": int(best_threads), "pairs_per_process": per, "pinned": bool(pin),
                          "wall_s": round(wall, 2), "slowest_worker_s": round(max(float(o[2]) for o in outs), 2)}
                else:
                    pp = {"value": None, "processes": nproc, "threads_per_process": int(best_threads), "note": "not finished within 90 s"}
        finally:
            for p in procs:
                try:
                    p.stdin.close()
                except Exception:
                    pass
                try:
                    p.wait(timeout=5)
                except Exception:
                    p.kill()
      except Exception as e:          # (an additional figure: a host that cannot start or pin the workers still gets its single-process baseline)
        pp = {"value": None, "processes": nproc, "threads_per_process": int(best_threads), "note": f"failed: {type(e).__name__}: {e}"}
    return {"value": round(done / t_reg, 4), "unit": "pairs/s", "cores": int(best_threads), "kind": "port",
            "value_process_parallel": None if pp is None else pp.get("value"), "process_parallel": pp, "host_hardware_threads": int(cores),
            "torch_threads_sweep_s_per_nn_pass": {str(k): v for k, v in sweep.items()},
            "impl": "restatement of the reference's Python path: torch-CPU chunked einsum NN (nn_max_n=250, matching.py:22-65) x2 directions + "
                    "numpy mutual filter + OpenMP RANSAC/refit (oracle.c) -- oracle/torch_cpu.py",
            "sample": f"{done} pairs of the same workload (N={args.n}, {args.mode}, {args.iters} iters); seconds per pair: NN {parts[0] / done:.2f}, "
                      f"reverse NN + filter {parts[1] / done:.2f}, RANSAC + refit {parts[2] / done:.2f}",
            "cpu_model": model, "omp_threads": int(orc.lib().orc_num_threads()),
            "oracle_port_pairs_per_s": round(2 / t_port, 4)}


def list_run(args, standalone=True):
    """bench.py --list A|B: BASELINE.json configs[2] / configs[3] as far as this environment allows (list-driven surrogate).
    standalone=False: called from the default run on an initialised single process (no process group of its own); returns the line."""
    import torch
    import torch.distributed as dist
    from lidarregistration_amd import harness, shard, metrics
    world = int(os.environ.get("WORLD_SIZE", "1")) if standalone else 1
    rank = int(os.environ.get("RANK", "0")) if standalone else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if standalone else 0
    use_dist = standalone and "WORLD_SIZE" in os.environ
    dev_index = (local_rank if not args.devices else int(args.devices.split(",")[local_rank])) if standalone else torch.cuda.current_device()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    host_coll = use_dist and args.dist_backend == "gloo"
    if use_dist:
        if host_coll:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    line = None

    class A:      # Experiments/test.py:294-313 defaults + the README's flags
        codebase = "GC"; prosac = True; fast_rejection = "ELC"; GC_LO = True; GPF_factor = 2.0; GPF_grid_wid = 10
    if args.list == "A":
        A.mode, A.iters, A.GC_conf = "GPF", 50000, 0.999
    else:
        A.mode, A.iters, A.GC_conf = "MMN", 1000000, 0.9995
    L = harness.load_list_fixture(args.list)
    rows_all = list(range(0, len(L["session"]), max(1, args.list_stride)))
    mine = [rows_all[i] for i in shard.shard_indices(len(rows_all), world, rank)]      # rank r: rows r, r+W, ... (DistributedSampler order)
    B = args.batch if args.batch > 0 else 32
    nstreams = args.streams if args.streams > 0 else 6      # (4 -> 6 calls in flight: +8 % on list A, +6 % on list B; 8 and 12 fall back, round 4)
    t0 = time.perf_counter()
    res = harness.eval_list_batched(L, mine, A, n=args.n, batch=B, nstreams=nstreams, device=dev)
    wall = time.perf_counter() - t0
    local = np.zeros((len(mine), shard.ROW))
    local[:, 1] = res["re_deg"]; local[:, 2] = res["te_m"] * 100; local[:, 17] = res["n_corr"]; local[:, 22:38] = res["T"].reshape(-1, 16)
    hard = None
    if args.hard:
        hs = harness.HARD[args.list]
        rh = harness.eval_list_batched(L, mine, A, n=args.n, batch=B, nstreams=nstreams, device=dev, **hs)
        local[:, 3] = rh["re_deg"]; local[:, 4] = rh["te_m"] * 100; local[:, 5] = rh["seconds"]
    if use_dist:
        dist.barrier()
        sec = torch.tensor([res["seconds"]], dtype=torch.float64, device="cpu" if host_coll else dev)
        dist.all_reduce(sec, op=dist.ReduceOp.MAX)
        seconds = float(sec.item())
        table = shard.gather_rows(local, len(rows_all), world, rank, device=None if host_coll else dev)
    else:
        seconds, table = res["seconds"], local[:len(rows_all)]
    if rank == 0:
        re, te = table[:, 1], table[:, 2] / 100
        ok5 = (re < metrics.RE_THRE_DEG) & (te < 0.6); ok2 = (re < 2.0) & (te < 0.6)
        call, fwd = res["stage_ms_per_pair"][0], res["stage_ms_per_pair"][1]
        ref_style = call - fwd + res["second_nn_share"] * fwd
        name = {"A": "ApolloSouthbay", "B": "NuScenes_boston"}[args.list]
        published = {"A": {"recall_5deg_0.6m": 0.9706, "recall_2deg_0.6m": 0.9702}, "B": {"recall_5deg_0.6m": 0.8279, "recall_2deg_0.6m": 0.8156}}[args.list]
        if args.hard:
            reh, teh = table[:, 3], table[:, 4] / 100
            okh5 = (reh < metrics.RE_THRE_DEG) & (teh < 0.6); okh2 = (reh < 2.0) & (teh < 0.6)
            hard = {"settings": harness.HARD[args.list], "recall_5deg_0.6m": round(float(okh5.mean()), 4), "recall_2deg_0.6m": round(float(okh2.mean()), 4),
                    "failed": int((~okh5).sum()), "pairs_per_s": round(len(rows_all) / max(float(table[:, 5].max()), 1e-9), 2),
                    "note": "same rows, same commands, harder synthetic data (harness.HARD): the recall that can move; the oracle pipeline is held to "
                            "the same success flags row by row in tests/test_gpu_lists.py"}
        line = {
            "metric": f"registration pairs/sec over the {name} balanced test list (list-driven synthetic surrogate, {args.n // 1000}k-pt pairs)",
            "value": round(len(rows_all) / seconds, 2), "unit": "pairs/s", "n_gpus": world, "higher_is_better": True, "scaling": "strong",
            "dtype": "f32", "data": "synthetic surrogate: each list row's ground-truth motion and overlap (-> rho), synthetic clouds and descriptors "
                                    "(SURVEY 8d); real scans / FCGF weights are not in this environment",
            "config": {"workload": f"configs[{2 if args.list == 'A' else 3}]: balanced_sets/{name}/test.txt, {len(rows_all)} of {len(L['session'])} rows, "
                                   f"--algo RANSAC --mode {A.mode} --iters {A.iters} --GC_conf {A.GC_conf} (codebase GC defaults: PROSAC, ELC, MSAC at the "
                                   f"truncated threshold, LO + final LS)", "pairs_per_batched_call": B, "batched_calls_in_flight_per_gpu": nstreams,
                       "parallelism": f"pair-sharded x{world}"},
            "pairs": len(rows_all), "seconds_registration": round(seconds, 3), "seconds_wall_incl_synthesis": round(wall, 2),
            "recall_5deg_0.6m": round(float(ok5.mean()), 4), "recall_2deg_0.6m": round(float(ok2.mean()), 4), "hard": hard,
            "RE_deg_mean_over_successes": round(float(re[ok5].mean()), 4) if ok5.any() else None,
            "TE_cm_mean_over_successes": round(float(te[ok5].mean() * 100), 3) if ok5.any() else None,
            "reference_recall_on_real_data": dict(published, source="BASELINE.md section 2 (reference's own test.coarse_motions.txt vs test.txt); NOT comparable: "
                                                                    "different data (surrogate) and no FCGF network here"),
            "time_per_pair_us": {"throughput": round(seconds / len(rows_all) * 1e6, 2),
                                 "whole_call_serialised": round(call * 1e3, 2), "forward_nn": round(fwd * 1e3, 2),
                                 "reference_style_FR.py:117": round(ref_style * 1e3, 2), "second_nn_share_of_forward_nn": round(res["second_nn_share"], 4),
                                 "note": "library events (lr_workspace_stage_times) over one batched call at a time, divided by its pairs; "
                                         "reference style = whole call - forward NN + the second neighbour's surcharge (filter incl. reverse NN + RANSAC + "
                                         f"final LS), sample of {res['stage_sample_pairs']} pairs"},
            "ransac_ids_examined_mean": round(float(res["n_ids"].mean()), 1), "filtered_pairs_mean": round(float(res["n_corr"].mean()), 1),
        }
        if standalone:
            print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier(); dist.destroy_process_group()
    return line


def main():
    args = parse()
    if args.traffic_key:
        B = args.batch if args.batch > 0 else (32 if args.n <= 60000 else 8)
        print(traffic_key(args.n, args.mode, args.codebase, args.iters, max(1, min(B, args.pairs, 64))))
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    if args.opt:
        from lidarregistration_amd import _ext as _e
        for kv in args.opt:
            k, v = kv.split("=")
            _e.DEFAULT_OPTIONS[k] = int(v)
    if args.list:
        return list_run(args)
    import torch
    import torch.distributed as dist
    from lidarregistration_amd import shard, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # under a launcher (torch.distributed.run sets WORLD_SIZE) the process group and the per-step collectives run even with one
    # rank, so that the RCCL branch is the same code at N = 1 as at N = 8
    use_dist = "WORLD_SIZE" in os.environ
    dry = args.dry_run
    if dry:
        dev = torch.device("cpu")
        if use_dist:
            dist.init_process_group("gloo")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        dev_index = local_rank if not args.devices else int(args.devices.split(",")[local_rank])
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
        if use_dist:
            if args.dist_backend == "nccl":
                dist.init_process_group("nccl", device_id=dev)
            else:
                dist.init_process_group("gloo")
    host_coll = use_dist and (dry or args.dist_backend == "gloo")      # collectives on host tensors

    B = args.batch if args.batch > 0 else (32 if args.n <= 60000 else 8)
    B = max(1, min(B, args.pairs, 64))
    # calls in flight (round 6, tools/r6_sweep2.sh, same box, alternating): the headline workload 12 200 pairs/s with three, 12 510 with TWO (step spread 14.7-16.5 ms
    # instead of 13-18; the filter-pass blocks at 2 080 instead of 1 990 MHz), 11 310 with one; --mode GPF flat; --codebase GC best with four
    nstreams = args.streams if args.streams > 0 else (4 if args.codebase == "GC" else (2 if args.mode in ("MNN", "MMN") else 3))
    res_bytes = 496
    pairs, T_gt = [], []
    if not dry:
        from lidarregistration_amd import FR, _ext

        class A:
            mode = args.mode; codebase = args.codebase; iters = args.iters; ransac_n = 3; GPF_factor = 2.0; GPF_grid_wid = 10
            o3d_conf = 1.0          # every one of the --iters hypotheses is evaluated (no confidence-based early exit)
            # (--codebase GC: FR.pair_params takes the reference CLI's defaults for everything else)
        params = FR.pair_params(A)
        res_bytes = ctypes.sizeof(_ext.PairResult)
        # resident inputs: `pairs` DISTINCT synthetic pairs per GPU (~8.4 MB each at 30k points), generated on the device
        for k in range(args.pairs):
            p = synth.make_pair_dev(N=args.n, seed=51 + rank * 100003 + k, device=dev)
            pairs.append((p["xyz0"], p["xyz1"], p["feats0"], p["feats1"])); T_gt.append(p["T_gt"])
        streams = [torch.cuda.Stream(device=dev) for _ in range(nstreams)]
        wss = [_ext.Workspace(args.n, args.n, 32, args.iters, max_pairs=B) for _ in range(nstreams)]
        have_clock = hasattr(_ext.lib(), "lr_workspace_clock")      # (absent only from older builds loaded through LIDARREG_LIB for an A/B)
        for w in wss:      # the filter-pass blocks sum their shader cycles / 100 MHz ticks: the clock the timed steps really ran at (lr_workspace_clock)
            if have_clock:
                w.set_option("clock_probe", 1)
    outs = torch.zeros((args.pairs, res_bytes), dtype=torch.uint8, device=dev)
    rows = torch.zeros((args.pairs, shard.ROW), dtype=torch.float64, device=dev)
    gathered = torch.zeros((world * args.pairs, shard.ROW), dtype=torch.float64, device="cpu" if host_coll else dev) if use_dist else None
    host_T = torch.zeros((args.pairs, 16), dtype=torch.float64)
    if not dry:
        host_T = host_T.pin_memory()

    host = None
    if args.include_h2d and not dry:
        host = [tuple(t.cpu().pin_memory() for t in pr) for pr in pairs]
        staged = [[tuple(torch.empty_like(t) for t in pairs[0]) for _ in range(B)] for _ in range(nstreams)]

    enq = [0.0]
    step_events = []          # one event per step on the current stream, behind the step's last operation: consecutive differences = step durations

    def step():
        t_enq = time.perf_counter()
        if dry:
            outs[:, :128] = torch.full((args.pairs, 128), rank + 1, dtype=torch.uint8)
        else:
            for c, lo in enumerate(range(0, args.pairs, B)):
                s = c % nstreams
                chunk = pairs[lo:lo + B]
                if host is not None:
                    with torch.cuda.stream(streams[s]):
                        for j in range(len(chunk)):
                            for dst, src in zip(staged[s][j], host[lo + j]):
                                dst.copy_(src, non_blocking=True)
                    chunk = staged[s][:len(chunk)]
                FR.register_batch_dev(chunk, params, out=outs[lo:lo + len(chunk)], ws=wss[s], stream=streams[s].cuda_stream)
            for s in streams:
                torch.cuda.current_stream().wait_stream(s)
        enq[0] += time.perf_counter() - t_enq
        # the step's product: the 4x4 transforms, on the host (128 bytes per pair, SURVEY 8d metric text)
        Tdev = outs[:, :128].view(torch.float64).view(args.pairs, 16)
        host_T.copy_(Tdev, non_blocking=not dry)
        if use_dist:
            # result rows = the 16 doubles of T (+ stats columns, zero here); one collective per step
            rows[:, 22:38] = Tdev
            dist.all_gather_into_tensor(gathered, rows.cpu() if host_coll and not dry else rows)
        if step_events is not None and not dry:
            ev = torch.cuda.Event(enable_timing=True); ev.record(); step_events.append(ev)

    def sync_all():
        if not dry:
            torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
            if not dry:
                torch.cuda.synchronize(dev)

    if not dry:
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
    for _ in range(args.warmup):
        step()
    sync_all()
    enq[0] = 0.0
    if not dry:
        for w in wss:
            if have_clock:
                w.clock(reset=True)
        del step_events[:]
        ev0 = torch.cuda.Event(enable_timing=True); ev0.record(); step_events.append(ev0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    # ---- what makes the number comparable between boxes: the shader clock of the timed steps' filter-pass blocks, the spread of the steps
    clock_mhz = step_stats = None
    if not dry:
        cyc = tk = 0
        for w in wss:
            if not have_clock:
                break
            _, c, t = w.clock(reset=True)
            cyc += c; tk += t
            w.set_option("clock_probe", 0)
        clock_mhz = 100.0 * cyc / tk if tk else None
        ms = sorted(step_events[i].elapsed_time(step_events[i + 1]) for i in range(len(step_events) - 1))
        if ms:
            step_stats = [round(ms[0], 3), round(ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2]), 3), round(ms[-1], 3)]
    step_events = None          # (the sustained leg and the roofline calls below record nothing)
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if host_coll else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- sustained rate: the same step loop for >= --sustain-s seconds (the K timed steps above last a fraction of a second; a
    #      power-limited kernel may settle at a lower clock over seconds).  Reported next to `value`, never instead of it.
    enq_timed = enq[0]                    # host time spent enqueueing the K timed steps (the sustained leg below keeps adding to enq[0])
    sustained = None
    if args.sustain_s > 0 and not dry:
        k2 = max(args.steps, int(np.ceil(args.sustain_s / (dt / args.steps))))
        if use_dist:      # every rank runs the same number of steps
            kt = torch.tensor([k2], dtype=torch.int64, device="cpu" if host_coll else dev)
            dist.all_reduce(kt, op=dist.ReduceOp.MAX)
            k2 = int(kt.item())
        sync_all()
        t0 = time.perf_counter()
        for _ in range(k2):
            step()
        sync_all()
        dt2 = time.perf_counter() - t0
        if use_dist:
            tmax = torch.tensor([dt2], dtype=torch.float64, device="cpu" if host_coll else dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt2 = float(tmax.item())
        sustained = {"seconds": round(dt2, 3), "steps": k2, "pairs_per_s": round(world * args.pairs * k2 / dt2, 2),
                     "ratio_to_value": round((world * args.pairs * k2 / dt2) / (world * args.pairs * args.steps / dt), 4),
                     "note": "same step loop, same barrier + synchronize bracket, run right after the K timed steps"}

    # ---- sanity of what was timed: every distinct pair of the last step registered correctly (from the host copy of T)
    recall = recall5 = None
    nn_fallback_rows = None
    score_frac = None
    n_corr_mean = n_valid_mean = 0.0
    if not dry:
        ok = ok5 = 0
        Th = host_T.numpy().reshape(-1, 4, 4)
        for i in range(args.pairs):
            re, te = transform_errors(Th[i], T_gt[i])
            ok += int(re < 2 and te < 0.6)           # BASELINE.json metric: recall@(2 deg, 0.6 m)
            ok5 += int(re < 5 and te < 0.6)
        recall, recall5 = ok / args.pairs, ok5 / args.pairs
        res = [_ext.PairResult.from_buffer_copy(outs[i].cpu().numpy().tobytes()) for i in range(args.pairs)]
        n_corr_mean = float(np.mean([r.n_corr for r in res])); n_valid_mean = float(np.mean([r.ransac.n_valid for r in res]))
        nn_fallback_rows = float(np.mean([r.n_nn_fixed for r in res]))
        score_frac = float(np.mean([r.reserved[0] for r in res])) / 1e6      # (model, correspondence) evaluations done / V*M (pilot-ordered scoring)
        if use_dist:      # the gathered table holds this rank's rows where the shard order says
            mine = gathered.view(world, args.pairs, shard.ROW)[rank, :, 22:38].cpu().numpy()
            assert np.array_equal(mine, host_T.numpy()), "gather order"
    elif use_dist:
        g = gathered.view(world, args.pairs, shard.ROW)[:, 0, 22]
        assert [float(v) for v in g] == [float(np.frombuffer(bytes([r + 1] * 8), np.float64)[0]) for r in range(world)], "gather order"

    # ---- roofline of the dominant kernel (pass B of the f16 filter): HIP events recorded by the library on the launch
    #      stream around the batched forward and reverse launches, averaged over `reps` batched calls
    roof = pair_roof = None
    total_pairs = world * args.pairs * args.steps
    value = total_pairs / dt
    if rank == 0 and not dry:
        ws = wss[0]
        reps = 10
        chunk = pairs[:B]
        ws.timing(True)
        for _ in range(reps):      # one timed call at a time: the library's events of a call are folded in before the next one
            FR.register_batch_dev(chunk, params, out=outs[:len(chunk)], ws=ws, stream=streams[0].cuda_stream)
            streams[0].synchronize()
            stage, ns = ws.stage_times()
        ws.timing(False)
        call_ms, fwd_nn_ms, fwd_filter_ms, rev_filter_ms, rs_ms, rev_nn_ms = [v / max(ns, 1) for v in stage]
        flop_pass = 2.0 * 32 * args.n * args.n                  # W_NN = 2 D N0 N1 per pair (ONE pass is algorithmic)
        launches_per_call = 1 if args.mode == "no_filter" else 2  # forward + reverse NN are separate launches of the same kernel
        # the library timed both filter-pass launches of each batched call: average duration per launch = AverageNs of this kernel
        # in the rocprofv3 summary of the same command
        t_launch = (fwd_filter_ms + rev_filter_ms) / launches_per_call * 1e-3
        flop_launch = flop_pass * len(chunk) / launches_per_call   # algorithmic flops one launch accounts for (len(chunk) pairs)
        achieved = flop_launch / t_launch / 1e12
        tj = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        tdoc = json.load(open(tj)) if os.path.exists(tj) else {}
        # traffic figures only for the workload they were collected on (else null: say nothing rather than the wrong launch's bytes)
        tkey = traffic_key(args.n, args.mode, args.codebase, args.iters, len(chunk))
        traffic, hbm_pair_doc, traffic_source = lookup_traffic(tdoc, tkey)
        nn_stage_s = (fwd_nn_ms + rev_nn_ms) * 1e-3
        roof = {"bound": "mfma", "kernel": "nn16_passb_kernel", "achieved": round(achieved, 3), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_F16_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source, "traffic_key": tkey, "launch_ms": round(t_launch * 1e3, 4),
                "forward_launch_ms": round(fwd_filter_ms, 4), "reverse_launch_ms": round(rev_filter_ms, 4),
                "pairs_per_launch": len(chunk), "launches_per_call": launches_per_call,
                # the ladder from the kernel to the step, all from this run's events (one batched call at a time):
                "forward_launch_frac": round(flop_pass * len(chunk) / (fwd_filter_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                "nn_stage_frac": round(flop_pass * len(chunk) / nn_stage_s / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                "whole_call_frac": round(flop_pass * len(chunk) / (call_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                "filter_pass_share_of_call": round((fwd_filter_ms + rev_filter_ms) / call_ms, 4),
                "call_ms": round(call_ms, 4), "forward_nn_ms": round(fwd_nn_ms, 4), "reverse_nn_ms": round(rev_nn_ms, 4),
                "note": "f16 MFMA filter (sample phase + walk in one kernel) + exact fp32 verification; one launch covers all pairs of a batched call; "
                        "launch_ms = average over the forward (all tiles) and the reverse (ordered, pruned) launch; achieved = (W/2 per pair x pairs per "
                        "launch) / launch_ms; forward_launch_frac credits the forward launch with all of W, nn_stage_frac = W / (forward NN + reverse NN: "
                        "prep, filter passes, exact verification, reverse ordering), whole_call_frac = W / the whole batched call",
                "ransac_gen_score_ms_per_call": round(rs_ms, 4)}
        # whole pair against the blended floor: one NN pass on the matrix pipe + V*M*27 flop of scoring on the vector pipe
        t_min = flop_pass / (MFMA_F16_PEAK_TFLOPS * 1e12) + n_valid_mean * n_corr_mean * 27.0 / (VALU_F32_PEAK_TFLOPS * 1e12)
        t_pair = dt / (args.pairs * args.steps)
        hbm_pair = hbm_pair_doc      # HBM bytes of ALL the library's kernels per pair (PMC passes of THIS workload, profiles/pmc_traffic.json; else null)
        pair_roof = {"t_min_us": round(t_min * 1e6, 2), "t_pair_us": round(t_pair * 1e6, 2), "frac": round(t_min / t_pair, 4),
                     "hbm_bytes_per_pair": hbm_pair, "hbm_GBps": None if hbm_pair is None else round(hbm_pair / t_pair / 1e9, 1),
                     "hbm_frac_of_8TBps": None if hbm_pair is None else round(hbm_pair / t_pair / 8.0e12, 4),
                     "hbm_source": None if hbm_pair is None else traffic_source,
                     "note": "t_min = W/peak_f16 + V*M*27/peak_fp32 (V = hypotheses past the pre-check, M = filtered pairs, means over the step)",
                     "V": round(n_valid_mean, 1), "M": round(n_corr_mean, 1)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not dry and args.codebase == "open3D":
        cpu = cpu_baseline(args, seed0=51)

    extra = None
    if rank == 0 and world == 1 and not dry and not use_dist and args.extra_list != "none" and args.codebase == "open3D" and args.n == 30000 and args.mode in ("MNN", "MMN"):
        # configs[2] (or [3]) in front of the driver: the reference's README command over every 8th row of the balanced list, surrogate data
        for w in wss:
            w.close()
        del pairs[:]
        torch.cuda.empty_cache()
        la = argparse.Namespace(**vars(args)); la.list = args.extra_list; la.list_stride = args.extra_list_stride; la.batch = 0; la.streams = 0; la.hard = 1
        try:          # (an additional record: whatever goes wrong in it must not cost the headline line)
            ll = list_run(la, standalone=False)
            extra = {"list_" + args.extra_list: {k: ll[k] for k in ("metric", "value", "unit", "pairs", "seconds_registration", "recall_5deg_0.6m", "recall_2deg_0.6m", "config",
                                                                     "time_per_pair_us", "ransac_ids_examined_mean", "filtered_pairs_mean")}}
            extra["list_" + args.extra_list]["hard"] = None if not ll.get("hard") else {k: ll["hard"][k] for k in ("recall_5deg_0.6m", "recall_2deg_0.6m", "failed", "pairs_per_s")}
            extra["list_" + args.extra_list]["data"] = ll["data"]
        except Exception as e:
            extra = {"list_" + args.extra_list: {"error": f"{type(e).__name__}: {e}"}}
    if rank == 0:
        line = {
            "metric": f"registration pairs/sec ({args.n // 1000}k-pt FCGF pairs, {'mutual-NN' if args.mode in ('MNN', 'MMN') else args.mode} + {args.iters // 1000}k RANSAC iters + refit)",
            "value": round(value, 2), "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            # rank 0's view of the K timed steps (device events behind each step) and the clock they ran at: `value` of two boxes -- or two rounds --
            # is compared on value_at_2.0GHz (the pool's boxes differ by +-4 %, mostly in the clock the power-limited filter pass settles at)
            "ms_per_step_minmedmax": step_stats, "clock_MHz": None if clock_mhz is None else round(clock_mhz, 1),
            "value_at_2.0GHz": None if not clock_mhz else round(value * 2000.0 / clock_mhz, 2),
            "dtype": "f32", "data": "synthetic" + (" (inputs copied from pinned host memory inside the timed region)" if args.include_h2d else ""),
            "config": {"workload": f"{'configs[1]' if args.n == 30000 else 'configs[4]-like dense'}: {args.n}-pt x32-d synthetic FCGF pair, --mode {args.mode} --iters {args.iters}, "
                                   f"{'3-pt sampling + ELC + LS refit' if args.codebase == 'open3D' else 'codebase GC defaults (PROSAC, ELC, MSAC, conf 0.999, LO + final LS)'}; {args.pairs} distinct resident pairs per GPU, T copied to the host inside the timed region",
                       "pairs_per_step_per_gpu": args.pairs, "pairs_per_batched_call": B, "batched_calls_in_flight_per_gpu": nstreams,
                       "parallelism": f"pair-sharded x{world}"},
            "recall_2deg_0.6m": None if recall is None else round(recall, 4), "recall_5deg_0.6m": None if recall5 is None else round(recall5, 4),
            "host_enqueue_ms_per_step": round(enq_timed / args.steps * 1e3, 3),
            "nn_rows_redone_by_full_scan_per_pair": nn_fallback_rows,
            "ransac_score_evaluations_frac_of_VxM": score_frac,
            "sustained": sustained,
            "roofline": roof, "pair_roofline": pair_roof, "cpu_baseline": cpu,
            "extra": extra,
        }
        if dry:
            line["data"] = "dry-run (no GPU work)"
        if cpu:
            # against BOTH host implementations, the headline ratio against the FASTER one (the OpenMP port of the oracle, not the
            # reference-style torch path whose 250-row einsum chunks leave most of the cores idle)
            line["speedup_vs_reference_style_path"] = round(value / cpu["value"], 1)      # (at the thread count the reference-style NN runs fastest with: cpu_baseline.cores)
            line["speedup_vs_openmp_port"] = round(value / cpu["oracle_port_pairs_per_s"], 1)
            if cpu.get("value_process_parallel"):
                line["speedup_vs_process_parallel_reference_path"] = round(value / cpu["value_process_parallel"], 1)      # (P processes x threads: cpu_baseline.process_parallel)
            # (the largest host figure: one process at its best thread count, the OpenMP port, or -- the reference's own sharding -- P processes side by side)
            line["speedup_vs_cpu_baseline"] = round(value / max(cpu["value"], cpu["oracle_port_pairs_per_s"], cpu.get("value_process_parallel") or 0.0), 1)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
