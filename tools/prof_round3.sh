#!/bin/bash
# GPU box: everything profiles/ holds for round 3 (pair-batched launches).  Outputs under gpurun_out/round3/.
#   usage: bash tools/prof_round3.sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/round3
mkdir -p $O
cd $R
python bench.py > $O/bench_line.json 2> $O/bench_stderr.log
tail -1 $O/bench_line.json | cut -c1-160
python bench.py --n 100000 --pairs 32 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_line_n100k.json 2>> $O/bench_stderr.log
python bench.py --mode GPF --no-cpu-baseline > $O/bench_line_gpf.json 2>> $O/bench_stderr.log
python bench.py --codebase GC --no-cpu-baseline > $O/bench_line_gc.json 2>> $O/bench_stderr.log
cd /tmp
prof() {   # tag, then bench args
  tag=$1; shift
  rm -rf /tmp/p_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -o $tag -- python3 $R/bench.py --no-cpu-baseline "$@" > /tmp/p_$tag.log 2>&1
  cp "$(find /tmp/p_$tag -name '*kernel_stats.csv' | head -1)" $O/${tag}_kernel_stats.csv
}
prof bench_streams1 --streams 1 --pairs 64 --steps 3 --warmup 1
prof bench_default --steps 3 --warmup 1          # (3 streams x 192 pairs since the end of round 3)
prof bench_gpf_streams1 --mode GPF --streams 1 --pairs 64 --steps 3 --warmup 1
prof bench_n100k_streams1 --n 100000 --streams 1 --pairs 16 --batch 8 --steps 3 --warmup 1
prof bench_gc_streams1 --codebase GC --streams 1 --pairs 64 --steps 3 --warmup 1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -o c -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --pairs 32 --no-cpu-baseline > /tmp/p_$c.log 2>&1
  cp "$(find /tmp/p_$c -name '*counter_collection.csv' | head -1)" /tmp/pmc_$c.csv
done
python3 $R/tools/pmc_to_json.py /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv $O/pmc_traffic.json > $O/pmc_traffic.txt
bash $R/tools/pmc_bench.sh > $O/pmc_sq_summary.txt 2>&1
bash $R/tools/pmc_passb.sh > $O/pmc_passb_summary.txt 2>&1
for l in A B; do python $R/bench.py --list $l --no-cpu-baseline > $O/list_${l}_bench_line.json 2>> $O/bench_stderr.log; done
ls -la $O
