#!/bin/bash
# GPU box: calls in flight -- the synthetic bench (open3D / GC codebase) and the CLI over the lists, interleaved repeats
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_streams; mkdir -p $O; cd $R
b() { python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --sustain-s 0 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   %.0f pairs/s' % d['value'])"; }
{
for rep in 1 2; do
  for s in 3 4 5 6; do echo "== bench open3D streams $s"; b --streams $s; done
  for s in 4 6 8; do echo "== bench GC streams $s"; b --codebase GC --streams $s; done
done
cd $R/Experiments
for rep in 1 2; do
  for s in 3 6; do
    echo "== CLI A streams $s"; python -m test --dataset A --algo RANSAC --mode GPF --iters 50000 --streams $s 2>&1 | grep "process 0:" | sed 's/.*registration region/   registration region/'
    echo "== CLI B streams $s"; python -m test --dataset B --algo RANSAC --mode MNN --iters 1000000 --GC_conf 0.9995 --streams $s 2>&1 | grep "process 0:" | sed 's/.*registration region/   registration region/'
  done
done
rm -rf $R/Experiments/outputs
} 2>&1 | tee $O/streams.txt
