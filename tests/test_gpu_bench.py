"""bench.py on the GPU with a small workload: the JSON line carries the contract keys, the roofline / pair_roofline / cpu_baseline
objects, and every timed pair registers correctly.  Needs an MI355X."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--n", "4000", "--iters", "3000", "--pairs", "12", "--batch", "4",
                        "--steps", "2", "--warmup", "1", "--sustain-s", "0.5"] + extra, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_line_small_workload():
    line = _bench(["--cpu-pairs", "2", "--cpu-budget-s", "5"])
    assert line["n_gpus"] == 1 and line["unit"] == "pairs/s" and line["value"] > 0 and line["scaling"] == "weak" and line["dtype"] == "f32"
    assert line["recall_2deg_0.6m"] == 1.0 and line["nn_rows_redone_by_full_scan_per_pair"] == 0.0
    assert line["config"]["pairs_per_batched_call"] == 4 and "workload" in line["config"]
    sus = line["sustained"]          # the same step loop run on for --sustain-s seconds, reported next to `value`
    assert sus["seconds"] > 0.05 and sus["steps"] >= 2 and sus["pairs_per_s"] > 0 and 0.3 < sus["ratio_to_value"] < 3.0
    roof = line["roofline"]
    assert roof["bound"] == "mfma" and roof["kernel"] == "nn16_passb_kernel" and roof["unit"] == "TFLOP/s" and roof["peak"] == 2500.0
    assert 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and roof["launch_ms"] > 0
    pr = line["pair_roofline"]
    assert 0 < pr["frac"] < 1 and pr["t_min_us"] < pr["t_pair_us"]
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "pairs/s" and cpu["value"] > 0 and cpu["cores"] >= 1 and "einsum" in cpu["impl"]
    # the headline ratio is taken against the FASTEST host figure: one process at its best thread count, the OpenMP port, or P pinned processes side by side
    pp = cpu["process_parallel"]
    assert "value_process_parallel" in cpu and "host_hardware_threads" in cpu and (pp is None or (pp["processes"] >= 2 and pp["threads_per_process"] == cpu["cores"]))
    ratios = [line["speedup_vs_reference_style_path"], line["speedup_vs_openmp_port"]] + ([line["speedup_vs_process_parallel_reference_path"]] if cpu["value_process_parallel"] else [])
    assert line["speedup_vs_cpu_baseline"] > 1 and line["speedup_vs_cpu_baseline"] == min(ratios)
    # what makes two records comparable: the clock of the timed steps' filter-pass blocks, the spread of the steps
    lo, med, hi = line["ms_per_step_minmedmax"]
    assert 0 < lo <= med <= hi and 500 < line["clock_MHz"] < 3000 and abs(line["value_at_2.0GHz"] - line["value"] * 2000 / line["clock_MHz"]) < 0.02 * line["value"]
    assert line["extra"] is None          # (the list-A leg belongs to the headline workload only)
    assert 0 < roof["whole_call_frac"] <= roof["nn_stage_frac"] <= roof["forward_launch_frac"] < 1 and roof["forward_launch_ms"] > roof["reverse_launch_ms"] > 0


def test_bench_gc_codebase_workload():
    """--codebase GC: the reference CLI's defaults (PROSAC, ELC, MSAC, confidence exit, local optimisation + final least squares)."""
    line = _bench(["--codebase", "GC", "--no-cpu-baseline"])
    assert line["recall_2deg_0.6m"] == 1.0 and "codebase GC defaults" in line["config"]["workload"] and line["cpu_baseline"] is None
    assert line["config"]["batched_calls_in_flight_per_gpu"] == 4


def test_bench_gpf_mode_and_h2d_variant():
    line = _bench(["--mode", "GPF", "--no-cpu-baseline", "--include-h2d"])
    assert line["recall_5deg_0.6m"] == 1.0 and "GPF" in line["metric"] and "pinned host memory" in line["data"] and line["cpu_baseline"] is None


def test_bench_under_the_launcher_runs_the_rccl_branch_with_one_rank():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the nccl process group, the per-step all_gather
    of the result rows and the max-over-ranks all_reduce execute on this GPU exactly as they do at N = 8."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--points", "4000", "--iters", "3000", "--pairs", "12",
                        "--batch", "4", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["recall_2deg_0.6m"] == 1.0


def test_two_ranks_share_this_gpu_through_the_self_launcher():
    """`python bench.py --gpus 2 --dist-backend gloo --devices 0,0`: the N > 1 path on a 1-GPU box -- spawn_ranks starts both ranks from a
    parent that never touches HIP, they rendezvous on 127.0.0.1, register their own (differently seeded) pairs on device 0, gather the
    result rows with one collective per step (host tensors: gloo) and take the max-over-ranks time.  The launch shape of the reference's
    test_parallel.sh:18-24; the same code runs with --dist-backend nccl at 8 GPUs."""
    line = _bench(["--gpus", "2", "--dist-backend", "gloo", "--devices", "0,0", "--no-cpu-baseline"])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["recall_2deg_0.6m"] == 1.0 and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "pair-sharded x2"
    roof = line["roofline"]
    assert roof["traffic_source"] is None or "NOT measured in this run" in roof["traffic_source"]
    assert 0 < roof["whole_call_frac"] <= roof["nn_stage_frac"] <= roof["forward_launch_frac"] < 1 and 0 < roof["filter_pass_share_of_call"] < 1


def test_a_dying_rank_ends_the_self_launched_run_quickly():
    """spawn_ranks polls all children: a rank that exits non-zero ends the run at once instead of after the collective's timeout."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--devices", "0,7", "--n", "4000",
                        "--iters", "3000", "--pairs", "4", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and time.time() - t0 < 120          # rank 1 has no device 7 on this box


def test_list_mode_with_two_ranks_on_this_gpu():
    """`bench.py --list B` through the self-launcher with two ranks on one GPU (gloo, host tensors): the list rows shard round-robin
    over the ranks (DistributedSampler order), the merged table covers every row once, the sustained leg belongs to the headline
    workload only."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--devices", "0,0", "--list", "B",
                        "--list-stride", "216", "--batch", "4", "--streams", "2", "--n", "8000", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["pairs"] == 12 and d["recall_5deg_0.6m"] == 1.0 and d["config"]["parallelism"] == "pair-sharded x2"
    assert 0.0 <= d["hard"]["recall_5deg_0.6m"] <= 1.0


# ------------------------------------------------------------------ world size 8 -- on ONE GPU: no scaling number comes out of these
# (DESIGN.md section 7); they make sure that eight ranks rendezvous, shard, gather and time together before an 8-GPU node is the first to try
def test_eight_ranks_share_this_gpu_through_the_self_launcher():
    line = _bench(["--gpus", "8", "--dist-backend", "gloo", "--devices", "0,0,0,0,0,0,0,0", "--no-cpu-baseline", "--pairs", "16", "--steps", "3", "--warmup", "1"])
    assert line["n_gpus"] == 8 and line["steps"] == 3 and line["value"] > 0 and line["recall_2deg_0.6m"] == 1.0 and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "pair-sharded x8"


def test_list_mode_with_eight_ranks_on_this_gpu():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dist-backend", "gloo", "--devices", "0,0,0,0,0,0,0,0", "--list", "B",
                        "--list-stride", "64", "--batch", "4", "--streams", "2", "--n", "8000", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["pairs"] == 41 and d["recall_5deg_0.6m"] == 1.0 and d["config"]["parallelism"] == "pair-sharded x8"      # ceil(2592 / 64) rows


def test_a_dying_rank_ends_the_eight_rank_run_quickly():
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dist-backend", "gloo", "--devices", "0,0,0,0,0,7,0,0", "--n", "4000",
                        "--iters", "3000", "--pairs", "8", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and time.time() - t0 < 180          # rank 5 has no device 7 on this box
