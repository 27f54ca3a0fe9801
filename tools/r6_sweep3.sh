#!/bin/bash
# GPU box, round 6: (a) round 5's library against the final one at TWO calls in flight; (b) calls in flight for the lists
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_sweep3; mkdir -p $O; cd $R
LIBS="r5 shipped" REPS="1 2 3" tools/r4_ab.sh --streams 2 2>&1 | tail -8 | tee $O/ab_streams2.txt
for rep in 1 2; do for L in A B; do for s in 6 4 5 8; do
  v=$(python bench.py --list $L --streams $s --hard 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  echo "rep $rep list $L calls in flight $s: $v"
done; done; done 2>&1 | tee $O/lists_inflight.txt
python bench.py > $O/bench_line.json 2>/dev/null; python - $O/bench_line.json <<'PY' | tee $O/bench_fields.txt
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "ms_per_step_minmedmax", "clock_MHz", "value_at_2.0GHz", "sustained")}, d["config"]["batched_calls_in_flight_per_gpu"], d["roofline"]["frac"])
PY
