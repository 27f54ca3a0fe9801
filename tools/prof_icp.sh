#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_i -o i -- python3 $R/tools/icp_prof.py > /tmp/p_i.log 2>&1
mkdir -p $R/gpurun_out; cp "$(find /tmp/p_i -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/icp_kernel_stats.csv
