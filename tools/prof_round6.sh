#!/bin/bash
# GPU box: everything profiles/r06_* holds for the final code of round 6.  Outputs under gpurun_out/round6/ (copied to profiles/r06_* by hand).
#   usage: tools/prof_round6.sh <commit>      (the GPU box has no .git: the commit the tree was built from is passed in.  Build the micro-harness
#   tools/bin/pb_micro_r6 and the probe library tools/bin/liblidarreg_probe.so first, here: hipcc ... tools/pb_micro.hip ; tools/r4_loprobe.sh)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
C=${1:-unrecorded}
O=$R/gpurun_out/round6
mkdir -p $O
cd $R
echo "commit $C" > $O/commit.txt
python bench.py > $O/bench_line.json 2> $O/bench_stderr.log
tail -1 $O/bench_line.json | cut -c1-200
python bench.py --n 100000 --pairs 32 --steps 3 --warmup 1 --no-cpu-baseline --sustain-s 0 > $O/bench_line_n100k.json 2>> $O/bench_stderr.log
python bench.py --mode GPF --no-cpu-baseline --sustain-s 0 --extra-list none > $O/bench_line_gpf.json 2>> $O/bench_stderr.log
python bench.py --codebase GC --no-cpu-baseline --sustain-s 0 > $O/bench_line_gc.json 2>> $O/bench_stderr.log
cd /tmp
prof() {   # tag, then bench args
  tag=$1; shift
  rm -rf /tmp/p_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -o $tag -- python3 $R/bench.py --no-cpu-baseline "$@" > /tmp/p_$tag.log 2>&1
  cp "$(find /tmp/p_$tag -name '*kernel_stats.csv' | head -1)" $O/${tag}_kernel_stats.csv
}
prof bench_streams1 --streams 1 --pairs 64 --steps 3 --warmup 1 --sustain-s 0
prof bench_default --steps 3 --warmup 1 --sustain-s 0
prof bench_sustained --steps 10 --warmup 2 --sustain-s 10          # the >= 10 s run: AverageNs of the filter pass over ~600 steps
grep -o '"sustained": {[^}]*}' /tmp/p_bench_sustained.log > $O/bench_sustained_under_profiler.txt
prof bench_gpf_streams1 --mode GPF --streams 1 --pairs 64 --steps 3 --warmup 1 --sustain-s 0
prof bench_n100k_streams1 --n 100000 --streams 1 --pairs 16 --batch 8 --steps 3 --warmup 1 --sustain-s 0
prof bench_gc_streams1 --codebase GC --streams 1 --pairs 64 --steps 3 --warmup 1 --sustain-s 0
cp $R/profiles/pmc_traffic.json $O/pmc_traffic.json
( cd $R; for f in "" "--n 100000"; do timeout 900 tools/pmc_traffic.sh $C $O/pmc_traffic.json $f; done ) > $O/pmc_traffic.txt 2>&1
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
( cd $R; python bench.py --no-cpu-baseline --sustain-s 0 --extra-list none > $O/bench_line_after_pmc.json 2>> $O/bench_stderr.log
  python bench.py --n 100000 --pairs 32 --steps 3 --warmup 1 --no-cpu-baseline --sustain-s 0 > $O/bench_line_n100k.json 2>> $O/bench_stderr.log )
bash $R/tools/pmc_passb.sh > $O/pmc_passb_summary.txt 2>&1
for l in A B; do python $R/bench.py --list $l --no-cpu-baseline > $O/list_${l}_bench_line.json 2>> $O/bench_stderr.log; done
( cd $R && STREAMS=1 LISTS="A B" STRIDE=8 tools/r4_listprof.sh > /dev/null 2>&1; cp gpurun_out/r4_listprof/list_A_s1_kernel_stats.csv $O/list_A_streams1_kernel_stats.csv; cp gpurun_out/r4_listprof/list_B_s1_kernel_stats.csv $O/list_B_streams1_kernel_stats.csv )
( cd $R && [ -f tools/bin/liblidarreg_probe.so ] && STRIDE=8 tools/r4_loprobe.sh run > /dev/null 2>&1 && cp gpurun_out/r4_loprobe/probe.txt $O/lo_probe.txt )
bash $R/tools/single_pair_prof.sh 2>&1 | grep -v "rocprim\|at::" > $O/single_pair_kernels.txt
python $R/tools/fr_latency.py 2>/dev/null > $O/fr_latency.txt
( cd $R && tools/bin/pb_micro_r6 30000 32 1 > $O/pb_micro.txt 2>&1; tools/bin/pb_micro_r6 30000 1 6 > $O/pb_micro_single.txt 2>&1 )
ls -la $O
