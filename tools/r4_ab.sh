#!/bin/bash
# GPU box: the same bench with the shipped library and with a build whose filter pass always uses the plain (v_max3) test
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_ab; mkdir -p $O; cd $R
for rep in ${REPS:-1 2}; do
for lib in ${LIBS:-plain shipped}; do
  if [ $lib = shipped ]; then unset LIDARREG_LIB; else export LIDARREG_LIB=$R/tools/bin/liblidarreg_$lib.so; fi
  python bench.py --no-cpu-baseline --sustain-s 0 --extra-list none "$@" > $O/line_$lib.json 2>/dev/null
  python - $O/line_$lib.json $lib <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"{sys.argv[2]:8s} pairs/s {d['value']:9.1f}  forward launch {r['forward_launch_ms']:.4f} ms  reverse {r['reverse_launch_ms']:.4f}  call {r['call_ms']:.4f}  frac {r['frac']:.4f}")
PY
done; done | tee $O/ab.txt
