#!/bin/bash
# GPU box, round 6: the functional pipeline (bench.py --pipeline stages: NN + filter stages of successive calls on one stream, RANSAC + refit on another,
# (needs the sources of commit 7e1fd1b: lr_workspace_ransac_stream and bench.py --pipeline were removed again by the commit after it)
# lr_workspace_ransac_stream) against whole calls in flight -- its test, then alternating runs
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_stages; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q 2>&1 | tail -5 | tee $O/tests.txt
run() { python bench.py --no-cpu-baseline --sustain-s ${SUS:-0} --extra-list none "$@" > $O/l.json 2>$O/err.txt || tail -3 $O/err.txt
  python - $O/l.json "$*" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; s = d.get("sustained") or {}
print(f"{sys.argv[2]:42s} pairs/s {d['value']:9.1f}  sustained {s.get('pairs_per_s', 0):9.1f}  clock {d['clock_MHz']:7.1f}  step min/med/max {d['ms_per_step_minmedmax']}  recall {d['recall_2deg_0.6m']}")
PY
}
for rep in 1 2 3 4; do run --pipeline calls --streams 2; run --pipeline stages; run --pipeline calls --streams 3; done 2>&1 | tee $O/ab.txt
SUS=10 run --pipeline stages 2>&1 | tee -a $O/ab.txt
SUS=10 run --pipeline calls --streams 2 2>&1 | tee -a $O/ab.txt
for rep in 1 2; do run --mode GPF --pipeline stages; run --mode GPF --pipeline calls --streams 3; run --codebase GC --pipeline stages; run --codebase GC --pipeline calls --streams 4; done 2>&1 | tee $O/variants.txt
