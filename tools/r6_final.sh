#!/bin/bash
# GPU box, round 6: what the driver runs at round end, on the final tree -- smoke, the whole GPU suite, the default bench line
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_final; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/gpu_suite.txt
( time python bench.py > $O/bench_line.json 2> $O/bench_stderr.log ) 2>&1 | tail -3 | tee $O/bench_time.txt
python - $O/bench_line.json <<'PY' | tee $O/bench_fields.txt
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "ms_per_step_minmedmax", "clock_MHz", "value_at_2.0GHz", "sustained")}, d["config"]["batched_calls_in_flight_per_gpu"], d["roofline"]["frac"])
c = d["cpu_baseline"]; print({k: c.get(k) for k in ("value", "cores", "value_process_parallel", "oracle_port_pairs_per_s")}, d.get("speedup_vs_cpu_baseline"))
e = (d.get("extra") or {}).get("list_A") or {}; print({k: e.get(k) for k in ("value", "pairs", "recall_5deg_0.6m", "hard")})
PY
