#!/bin/bash
# GPU box: same-box A/B of two builds (tools/bin/liblidarreg_old.so against the shipped one): bench, single-pair kernels, FR() latency
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_ab2; mkdir -p $O; cd $R
{
LIBS="old shipped" tools/r4_ab.sh
for lib in old shipped; do
  if [ $lib = shipped ]; then unset LIDARREG_LIB; else export LIDARREG_LIB=$R/tools/bin/liblidarreg_$lib.so; fi
  echo "== $lib"; python tools/fr_latency.py 2>/dev/null | head -3
  bash tools/single_pair_prof.sh 2>/dev/null | grep "total us"
done
} 2>&1 | tee $O/ab2.txt
