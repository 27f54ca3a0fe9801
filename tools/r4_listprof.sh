#!/bin/bash
# GPU box: kernel trace of the list-driven runs (configs[2] / [3]); STREAMS=1 gives uncontended kernel times
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_listprof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for L in ${LISTS:-A B}; do
  rm -rf /tmp/lp_$L
  timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/lp_$L -o lp --output-format csv -- python3 $R/bench.py --list $L --list-stride ${STRIDE:-8} --hard 0 --no-cpu-baseline --streams ${STREAMS:-4} > $O/list_${L}_line.txt 2>&1
  f=$(find /tmp/lp_$L -name "*kernel_stats.csv" | head -1); cp "$f" $O/list_${L}_s${STREAMS:-4}_kernel_stats.csv
  grep '^{' $O/list_${L}_line.txt | tail -1 | cut -c1-300
done
