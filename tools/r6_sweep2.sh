#!/bin/bash
# GPU box, round 6: calls in flight (streams) for the headline workload, alternating, four repetitions; then the same for the GPF / GC variants
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_sweep2; mkdir -p $O; cd $R
run() { python bench.py --no-cpu-baseline --sustain-s ${SUS:-0} --extra-list none "$@" > $O/l.json 2>/dev/null
  python - $O/l.json "$*" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; s = d.get("sustained") or {}
print(f"{sys.argv[2]:42s} pairs/s {d['value']:9.1f}  sustained {s.get('pairs_per_s', 0):9.1f}  clock {d['clock_MHz']:7.1f}  step min/med/max {d['ms_per_step_minmedmax']}")
PY
}
for rep in 1 2 3 4; do for s in 3 2 1; do run --streams $s; done; done 2>&1 | tee $O/streams.txt
SUS=10 run --streams 2 2>&1 | tee -a $O/streams.txt
SUS=10 run --streams 3 2>&1 | tee -a $O/streams.txt
for rep in 1 2; do for s in 3 2; do run --mode GPF --streams $s; run --codebase GC --streams $s; run --codebase GC --streams 4; done; done 2>&1 | tee $O/variants.txt
