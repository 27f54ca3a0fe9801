"""bench.py's launcher and collective plumbing on CPU: `python bench.py --gpus 2 --dry-run` must start one process per rank by
itself (no torch.distributed.run), rendezvous over gloo on 127.0.0.1, run its own step()/all_gather/max-over-ranks code with
fake result rows, and print exactly one JSON line on rank 0."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--steps", "2", "--warmup", "1", "--pairs", "8"] + extra,
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_self_launch_two_ranks_gloo():
    line = _run(["--gpus", "2"])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["parallelism"] == "pair-sharded x2" and line["value"] > 0
    assert line["cpu_baseline"] is None and line["roofline"] is None       # dry run: nothing measured


def test_single_rank_line_has_the_contract_keys():
    line = _run([])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["unit"] == "pairs/s" and "workload" in line["config"]
    # what makes two records comparable: present in every line, null where nothing was measured (dry run)
    for k in ("ms_per_step_minmedmax", "clock_MHz", "value_at_2.0GHz"):
        assert k in line and line[k] is None, k


def test_process_parallel_cpu_baseline_worker_protocol():
    """bench.py's cpu_baseline also runs the reference-style path as P processes side by side (the reference's own sharding,
    test_parallel.sh:18-20): oracle/cpu_worker.py says READY after its untimed warm-up, starts on GO and reports DONE <pairs> <seconds>."""
    env = dict(os.environ, OMP_NUM_THREADS="2", HIP_VISIBLE_DEVICES="", PYTHONPATH=ROOT)
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", "2", "1500", "MNN", "500", "2", str(100 * (w + 1))], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                              text=True, env=env, cwd=ROOT) for w in range(2)]
    try:
        assert all(p.stdout.readline().strip() == "READY" for p in procs)
        for p in procs:
            p.stdin.write("GO\n"); p.stdin.flush()
        outs = [p.stdout.readline().split() for p in procs]
        assert all(o[0] == "DONE" and o[1] == "2" and float(o[2]) > 0 for o in outs), outs
        assert all(p.wait(timeout=60) == 0 for p in procs)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def test_under_an_external_launcher_environment():
    # what torch.distributed.run provides: the script must not spawn again
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    line = _run(["--gpus", "1"], dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    assert line["n_gpus"] == 1


def test_traffic_figures_belong_to_one_workload():
    """profiles/pmc_traffic.json is keyed by workload: bench.py reports `traffic` / `hbm_bytes_per_pair` only for the workload the
    counters were collected on, and null for everything else (round 4 reported the 30k launch's bytes for a launch of 8 x 100k points)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    doc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    t, pair, src = b.lookup_traffic(doc, b.DEFAULT_TRAFFIC_KEY)
    assert t and t > 1e8 and pair and pair > 3e7 and b.DEFAULT_TRAFFIC_KEY in src
    for key in (b.traffic_key(100000, "MNN", "open3D", 50000, 8), b.traffic_key(30000, "MNN", "open3D", 50000, 16), b.traffic_key(30000, "no_filter", "open3D", 50000, 32),
                b.traffic_key(30000, "MNN", "open3D", 1000, 32), b.traffic_key(12345, "GPF", "GC", 50000, 32)):
        if key not in doc.get("workloads", {}):
            assert b.lookup_traffic(doc, key) == (None, None, None), key
    assert b.traffic_key(30000, "MMN", "open3D", 50000, 32) == b.DEFAULT_TRAFFIC_KEY          # the reference's alias
    fake = {"workloads": {"k1": {"nn16_passb_kernel<true>": {"hbm_bytes_per_launch": 100, "launches": 2}, "_pair": {"hbm_bytes_per_pair": 7}, "_meta": {"commit": "abc"}}}}
    assert b.lookup_traffic(fake, "k1")[:2] == (100, 7) and "abc" in b.lookup_traffic(fake, "k1")[2] and b.lookup_traffic(fake, "k2") == (None, None, None)
    assert b.lookup_traffic({}, "k1") == (None, None, None)
