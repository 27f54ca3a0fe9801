#!/bin/bash
# GPU box, round 6: pairs per batched call x calls in flight for the headline workload (192 pairs per step), two repetitions
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_sweep; mkdir -p $O; cd $R
for rep in 1 2; do
for cfg in "32 3" "48 2" "48 3" "48 4" "64 2" "64 3" "24 4" "32 4" "32 2"; do set -- $cfg
  python bench.py --no-cpu-baseline --sustain-s 0 --extra-list none --batch $1 --streams $2 > $O/l.json 2>/dev/null
  python - $O/l.json $1 $2 $rep <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"rep {sys.argv[4]} batch {sys.argv[2]:>3s} streams {sys.argv[3]}: pairs/s {d['value']:9.1f}  at 2.0 GHz {d['value_at_2.0GHz']:9.1f}  clock {d['clock_MHz']:7.1f}  call {r['call_ms']:.3f} ms  fwd {r['forward_launch_ms']:.3f}")
PY
done; done | tee $O/sweep.txt
