#!/bin/bash
# GPU box: the bench under different workspace options (same box, same library)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_opts; mkdir -p $O; cd $R
for rep in 1 2 3; do for o in "${@}"; do
  python bench.py --no-cpu-baseline --sustain-s 0 $o > $O/line.json 2>/dev/null
  python - $O/line.json "$o" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"{sys.argv[2]:40s} pairs/s {d['value']:9.1f}  forward {r['forward_launch_ms']:.4f} ms  reverse {r['reverse_launch_ms']:.4f}  fwd NN {r['forward_nn_ms']:.4f}  call {r['call_ms']:.4f}  frac {r['frac']:.4f}")
PY
done; done | tee $O/opts.txt
