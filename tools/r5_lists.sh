#!/bin/bash
# GPU box, round 5: bench.py --list A / B alternating between the shipped library and tools/bin/liblidarreg_<name>.so (same box)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_lists; mkdir -p $O; cd $R
for rep in ${REPS:-1 2}; do
for L in ${LISTS:-A B}; do
for lib in ${LIBS:-base shipped}; do
  if [ $lib = shipped ]; then unset LIDARREG_LIB; else export LIDARREG_LIB=$R/tools/bin/liblidarreg_$lib.so; fi
  python bench.py --list $L --no-cpu-baseline ${EXTRA} > $O/line_${L}_$lib.json 2>/dev/null
  python - $O/line_${L}_$lib.json $L $lib <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"list {sys.argv[2]} {sys.argv[3]:8s} pairs/s {d['value']:9.1f}  recall5 {d['recall_5deg_0.6m']:.4f}  hard {(d.get('hard') or {}).get('recall_5deg_0.6m')}")
PY
done; done; done | tee $O/lists.txt
