#!/bin/bash
# GPU box: kernel trace of the list-driven runs (configs[2] / [3]), every 8th row, hard surrogate off
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_listprof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for L in ${LISTS:-A B}; do
  rm -rf /tmp/lp_$L
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/lp_$L -o lp --output-format csv -- python3 $R/bench.py --list $L --list-stride 8 --hard 0 --no-cpu-baseline > $O/list_${L}_line.txt 2>&1
  f=$(find /tmp/lp_$L -name "*kernel_stats.csv" | head -1); cp "$f" $O/list_${L}_kernel_stats.csv
  tail -1 $O/list_${L}_line.txt | cut -c1-400
done
