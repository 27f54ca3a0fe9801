#!/bin/bash
# GPU box: pairs per batched call / calls in flight for the synthetic bench, interleaved repeats
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_batch; mkdir -p $O; cd $R
b() { python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-s 0 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.0f pairs/s  (frac %.4f)' % (d['value'], d['roofline']['frac']))"; }
{ for rep in 1 2; do
  for cfg in "32 3" "48 2" "48 4" "64 3" "24 4" "16 6"; do set -- $cfg; echo "== batch $1 streams $2 (pairs 192)"; b --batch $1 --streams $2 --pairs 192; done
done; } 2>&1 | tee $O/batch.txt
