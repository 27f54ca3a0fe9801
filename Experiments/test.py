"""CLI with the reference's surface (Experiments/test.py:273-353): run from this directory as

    python -m test --dataset A --algo RANSAC --mode GPF --iters 50000
    ./test_parallel.sh --dataset B --algo RANSAC --mode MNN --iters 1000000 --GC_conf 0.9995      (= python -m test launch ...)

Same flags and defaults for the RANSAC path, same `test_parallel <start_time> <tmp_base> <world> <rank|analysis>` protocol,
same outputs (`outputs/<dataset>.Test.<time>/{raw_stats.npy,log.txt}`) plus `coarse_motions.txt` (format of
FCGF_FAST/test.py:86-106).  Datasets and FCGF weights are not part of this repo; pairs come from, in order:
  0. the reference's cloud cache (env LIDARREG_CLOUD_CACHE=<dir with <session>_<idx>.npy>, voxel-deduplicated on the GPU like the
     reference's loader) + a feature cache for the same voxelisation + LIDARREG_BALANCED_SETS,
  1. a feature cache  (env LIDARREG_FEATURE_CACHE=<dir> + LIDARREG_BALANCED_SETS=<dir with <set>/<phase>.txt>),
  2. the list-driven synthetic surrogate (GT motion + overlap from the list rows: LIDARREG_BALANCED_SETS, or for --dataset A / B
     the full test lists committed under tests/golden/lists),
  3. plain synthetic pairs (--dataset synthetic --num_pairs P).
"""
import argparse
import datetime
import logging
import os
import sys
import tempfile
import time
from glob import glob

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s", datefmt="%m/%d %H:%M:%S", stream=sys.stdout)


def str2bool(v):
    return str(v).lower() in ("true", "1")       # Experiments/config.py:18-19


def get_args(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] == "test_parallel":        # Experiments/test.py:275-285
        start_time, tmp_file_base, world_size = argv[1], argv[2], int(argv[3])
        do_analysis = argv[4] == "analysis"
        rank = None if do_analysis else int(argv[4])
        argv = argv[5:]
    else:
        start_time = None
        tmp_file_base = tempfile.gettempdir() + "/test_%016d" % int(np.random.rand() * 10 ** 16)
        world_size, rank, do_analysis = 1, 0, True
    p = argparse.ArgumentParser()
    p.add_argument("--dataset", type=str, default="synthetic", help="A, B, S, K, L (balanced sets) or synthetic")
    p.add_argument("--algo", type=str, default="RANSAC", choices=["RANSAC"])
    p.add_argument("--codebase", type=str, default="GC", choices=["open3D", "GC"])
    p.add_argument("--mode", type=str, default="MNN", help="MNN (alias MMN), GPF or no_filter")
    p.add_argument("--max_samples", type=int, default=None)
    p.add_argument("--iters", type=int, default=None)
    p.add_argument("--phase", type=str, default="test", choices=["train", "validation", "test"])
    p.add_argument("--spatial_coherence_weight", type=float, default=0.0)
    p.add_argument("--fast_rejection", type=str, default="ELC", choices=["SPRT", "ELC", "NONE"])
    p.add_argument("--prosac", type=str2bool, default=True)
    p.add_argument("--GPF_factor", type=float, default=2.0)
    p.add_argument("--GPF_grid_wid", type=int, default=10)
    p.add_argument("--GPF_max_matches", type=int, default=10 ** 9)
    p.add_argument("--GC_conf", type=float, default=0.999)
    p.add_argument("--GC_LO", type=str2bool, default=True)
    # additions of this implementation
    p.add_argument("--num_pairs", type=int, default=32, help="synthetic: number of pairs")
    p.add_argument("--synthetic_n", type=int, default=30000, help="synthetic: points per cloud")
    p.add_argument("--seed", type=int, default=51)
    p.add_argument("--batch", type=int, default=32, help="list rows per batched call (lr_register_batch: every kernel launched once for all of them)")
    p.add_argument("--num_workers", type=int, default=8, help="threads that read the NEXT window's files (feature cache / cloud cache) while this window registers; 0: read on the main thread, as the reference's DataLoader(num_workers=0) does")
    p.add_argument("--streams", type=int, default=0, help="HIP streams the window's batched calls are spread over (0: 3; 6 measured the same over both lists -- the calls in flight count, not the streams)")
    p.add_argument("--in_flight", type=int, default=6, help="batched calls per window (each on its own workspace, round-robin over 3 streams); with --serial: pairs in flight")
    p.add_argument("--serial", type=str2bool, default=False, help="one lr_register_pair per list row, like the reference harness (cross-check / latency)")
    p.add_argument("--icp", type=str2bool, default=True, help="refine by point-to-point ICP and fill stats columns 11-14 (test.py:183-193)")
    p.add_argument("--o3d_conf", type=float, default=0.9995, help="confidence of the open3D codebase (FR.py:136)")
    args = p.parse_args(argv)
    args.start_time, args.tmp_file_base, args.world_size, args.rank, args.do_analysis = start_time, tmp_file_base, world_size, rank, do_analysis
    from lidarregistration_amd import io_lists
    args.dataset_name = io_lists.DATASET_NAMES.get(args.dataset, args.dataset)
    t = args.start_time or datetime.datetime.now().strftime("%Y%m%d_%H_%M_%S")
    args.outdir = f"outputs/{args.dataset_name}.Test.{t}/"
    os.makedirs(args.outdir, exist_ok=True)
    return args


def make_source(args):
    from lidarregistration_amd import harness, io_lists
    sets, cache = os.environ.get("LIDARREG_BALANCED_SETS"), os.environ.get("LIDARREG_FEATURE_CACHE")
    if args.dataset in io_lists.DATASET_NAMES and sets:
        lst = io_lists.read_pair_list(os.path.join(sets, args.dataset_name, args.phase + ".txt"))
        clouds = os.environ.get("LIDARREG_CLOUD_CACHE")
        if clouds and cache:
            return harness.RefCloudSource(lst, os.path.join(clouds, args.dataset_name), os.path.join(cache, args.dataset_name, args.phase))
        if cache:
            return harness.CacheSource(lst, os.path.join(cache, args.dataset_name, args.phase))
        return harness.SyntheticSource(len(lst["session"]), n=args.synthetic_n, seed=args.seed, pair_list=lst)
    if args.dataset in ("A", "B") and args.phase == "test":
        # no list directory given: the reference's FULL balanced test lists as committed fixtures (tests/golden/lists, made from
        # balanced_sets/<set>/test.txt) drive the synthetic surrogate -- 7008 / 2592 rows
        lst = harness.load_list_fixture(args.dataset)
        return harness.SyntheticSource(len(lst["session"]), n=args.synthetic_n, seed=args.seed, pair_list=lst)
    return harness.SyntheticSource(args.num_pairs, n=args.synthetic_n, seed=args.seed)


def test_subset(args):
    import torch
    from lidarregistration_amd import harness, shard
    source = make_source(args)
    P = len(source) if args.max_samples is None else min(len(source), args.max_samples)
    idx = shard.shard_indices(P, args.world_size, args.rank)
    print("process %d, GPU: cuda:%d, %d pairs" % (args.rank, torch.cuda.current_device(), len(idx)))
    t0 = time.time()
    if args.serial:
        stats, T = harness.eval_pairs_serial(source, idx, args, in_flight=min(args.in_flight, 4), verbose=args.rank == 0)
    else:
        stats, T = harness.eval_pairs(source, idx, args, batch=args.batch, in_flight=args.in_flight, nstreams=args.streams or 3, verbose=False, workers=args.num_workers)
    wall = time.time() - t0
    msg = "process %d: %d pairs in %.2f s end to end (data source + registration + ICP + statistics): %.1f pairs/s" % (args.rank, len(idx), wall, len(idx) / max(wall, 1e-9))
    if not args.serial:
        r = harness.LAST_RUN
        msg += "; registration region %.3f s = %.1f pairs/s (data %.2f s, ICP %.2f s, statistics %.2f s)" % (
            r["registration_s"], len(idx) / max(r["registration_s"], 1e-9), r["data_s"], r["icp_s"], r["stats_s"])
    print(msg, flush=True)
    with open(f"{args.tmp_file_base}_throughput_{args.world_size}_{args.rank}.txt", "w") as fid:
        fid.write(msg + "\n")
    np.save(f"{args.tmp_file_base}_res_{args.world_size}_{args.rank}.npy",
            np.concatenate([stats, T.reshape(-1, 16), harness.LAST_WHOLE_PATH[:, None], np.asarray(idx, np.float64)[:, None]], 1))


def analyze_stats(args):
    from lidarregistration_amd import io_lists, metrics
    parts = [np.load(f) for f in sorted(glob(args.tmp_file_base + "_res_*"))]
    allrows = np.vstack(parts)
    _, first = np.unique(allrows[:, -1], return_index=True)          # drop wrap-around padding, restore list order
    allrows = allrows[first]
    stats, T, whole = allrows[:, :22], allrows[:, 22:38].reshape(-1, 4, 4), allrows[:, 38]
    np.save(args.outdir + "raw_stats.npy", stats)
    from lidarregistration_amd import harness
    with open(args.outdir + "raw_stats.columns.txt", "w") as fid:      # (the batched engine's time columns are window shares: say so next to the file)
        fid.write(harness.stats_columns(args.serial))
    s = metrics.summarize(stats, args.algo)
    # the reference bills filter + RANSAC + the second neighbour's surcharge (FR.py:117; column 9 above); the whole device path of a
    # call additionally contains the first nearest-neighbour search
    s += "\nwhole device path per pair incl. the forward NN (not billed by the reference): mean %.2f ms, 99%% %.2f ms" % (
        np.nanmean(whole) * 1e3, np.nanpercentile(whole, 99) * 1e3)
    logging.info(s)
    with open(args.outdir + "log.txt", "w") as fid:
        for k, v in args.__dict__.items():
            fid.write(f"{k} = {v}\n")
        fid.write("\n" + s)
        for f in sorted(glob(args.tmp_file_base + "_throughput_*")):
            fid.write("\n" + open(f).read().strip())
    io_lists.write_coarse_motions(args.outdir + "coarse_motions.txt", stats[:, 19], stats[:, 20], stats[:, 21], T)
    return stats


def launch(argv):
    """`python -m test launch <flags>` (= ./test_parallel.sh <flags>): one rank per GPU over the positional protocol
    `test_parallel <start_time> <tmp_base> <world> <rank>` (Experiments/test.py:275-285 of the reference), then the analysis pass in
    this process.  All ranks are watched (lidarregistration_amd.launch.run_ranks): when one dies the others are stopped, the partial
    files are removed and NO analysis is run -- the exit code is the failed rank's."""
    from lidarregistration_amd import launch as L
    gpus = L.gpu_list()
    if not gpus:
        sys.exit("no GPU visible (set LIDARREG_GPUS to choose devices)")
    world = len(gpus)
    start_time = datetime.datetime.now().strftime("%Y%m%d_%H_%M_%S")
    fd, base = tempfile.mkstemp(prefix="lidarreg_ranks_")
    os.close(fd)
    here = os.path.dirname(os.path.abspath(__file__))
    cmds = [[sys.executable, "-m", "test", "test_parallel", start_time, base, str(world), str(r)] + list(argv) for r in range(world)]
    pp = here + (os.pathsep + os.environ["PYTHONPATH"] if os.environ.get("PYTHONPATH") else "")
    rc = L.run_ranks(cmds, [{var: value, "PYTHONPATH": pp} for var, value in gpus])
    try:
        if rc != 0:
            logging.error("a rank failed (exit code %d): the other ranks were stopped, no analysis", rc)
            sys.exit(rc)
        return main(["test_parallel", start_time, base, str(world), "analysis"] + list(argv))
    finally:
        for f in glob(base + "*"):
            os.remove(f)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] == "launch":
        return launch(argv[1:])
    args = get_args(argv)
    logging.info("Starting")
    if args.rank is not None:
        test_subset(args)
    if args.do_analysis:
        stats = analyze_stats(args)
        for f in glob(args.tmp_file_base + "_res_*") + glob(args.tmp_file_base + "_throughput_*"):
            os.remove(f)
        return stats


if __name__ == "__main__":
    main()
