#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python bench.py --no-cpu-baseline --mode GPF 2>&1 | tail -1 | cut -c1-200
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_g -o g -- python3 $R/bench.py --mode GPF --streams 1 --pairs 32 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/p_g.log 2>&1
mkdir -p $R/gpurun_out; cp "$(find /tmp/p_g -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/gpf_streams1_kernel_stats.csv
