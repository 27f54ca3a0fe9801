#!/bin/bash
# GPU box: everything profiles/ holds for a round.  Outputs under gpurun_out/round/.
#   usage: bash tools/prof_round.sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round
mkdir -p $O
cd $R
python bench.py > $O/bench_line.json 2> $O/bench_stderr.log
tail -1 $O/bench_line.json | cut -c1-200
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_s1 -o s1 -- python3 $R/bench.py --streams 1 --pairs 32 --steps 3 --warmup 1 --no-cpu-baseline > /tmp/p_s1.log 2>&1
cp "$(find /tmp/p_s1 -name '*kernel_stats.csv' | head -1)" $O/streams1_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_def -o def -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/p_def.log 2>&1
cp "$(find /tmp/p_def -name '*kernel_stats.csv' | head -1)" $O/default_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_f -o f -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --pairs 8 --no-cpu-baseline > /tmp/p_f.log 2>&1
cp "$(find /tmp/p_f -name '*counter_collection.csv' | head -1)" $O/pmc_fetch.csv
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_w -o w -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --pairs 8 --no-cpu-baseline > /tmp/p_w.log 2>&1
cp "$(find /tmp/p_w -name '*counter_collection.csv' | head -1)" $O/pmc_write.csv
ls -la $O
head -3 $O/pmc_fetch.csv
