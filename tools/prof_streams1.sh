#!/bin/bash
# GPU box: single-stream rocprofv3 kernel summary of the bench workload (written to gpurun_out/)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_s1 -o s1 -- python3 $R/bench.py --streams 1 --pairs 32 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/prof_s1.log 2>&1
f=$(find /tmp/prof_s1 -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out; cp "$f" $R/gpurun_out/streams1_kernel_stats.csv
head -25 "$f" | cut -d, -f1-5
tail -1 /tmp/prof_s1.log | cut -c1-200
