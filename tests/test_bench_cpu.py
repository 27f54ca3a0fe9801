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


def test_under_an_external_launcher_environment():
    # what torch.distributed.run provides: the script must not spawn again
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    line = _run(["--gpus", "1"], dict(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
    assert line["n_gpus"] == 1
