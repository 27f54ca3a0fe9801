#!/bin/bash
# GPU box: same-box A/B of an alternative build tools/bin/liblidarreg_$1.so against the shipped library: NN parity subset with the
# alternative first, then bench alternation, FR() latency
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_ab3; mkdir -p $O; cd $R
ALT=${1:-kp}
{
LIDARREG_LIB=$R/tools/bin/liblidarreg_$ALT.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -2
LIBS="$ALT shipped" tools/r4_ab.sh
for lib in $ALT shipped; do
  if [ $lib = shipped ]; then unset LIDARREG_LIB; else export LIDARREG_LIB=$R/tools/bin/liblidarreg_$lib.so; fi
  echo "== $lib"; python tools/fr_latency.py 2>/dev/null | head -3
done
} 2>&1 | tee $O/ab3_$ALT.txt
