#!/bin/bash
# GPU box: repeat one pytest selection N times (flaky-failure hunting)
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/r4_loop; cd $R
N=${N:-5}
for i in $(seq 1 $N); do timeout 600 python -m pytest "$@" -x -q -m gpu -s 2>&1 | grep -E "passed|failed|Error|assert|contention|^E " | head -12; done | tee $R/gpurun_out/r4_loop/out.txt
