#!/bin/bash
# GPU box, round 6: can the exact verification run NEXT TO the filter pass instead of in its place?  nn16_exact_kernel built to need 46 / 34
# registers (column row fetched in 2 / 4 pieces) so that a wave of it fits beside three filter-pass waves on a SIMD: pipeline A/B + its
# stand-alone time
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_sixth; mkdir -p $O; cd $R
LIBS="shipped lean2 lean4" REPS="1 2 3" tools/r4_ab.sh 2>&1 | tail -12 | tee $O/ab.txt
for lib in lean2 lean4; do
  export LIDARREG_LIB=$R/tools/bin/liblidarreg_$lib.so
  cd /tmp; rm -rf /tmp/p_$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$lib -o s1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --pairs 64 --steps 3 --warmup 1 --sustain-s 0 > /tmp/p_$lib.log 2>&1
  cp "$(find /tmp/p_$lib -name '*kernel_stats.csv' | head -1)" $O/${lib}_streams1_kernel_stats.csv
  grep "nn16_exact" $O/${lib}_streams1_kernel_stats.csv | cut -c1-30,200-400
  cd $R
done 2>&1 | tee $O/exact_alone.txt
unset LIDARREG_LIB
LIBS="shipped lean2" REPS="1 2" tools/r5_lists.sh 2>&1 | tail -10 | tee $O/lists.txt
