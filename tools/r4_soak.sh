#!/bin/bash
# GPU box: the random soaks against the oracle (not part of the suite), sized for ~10 minutes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_soak; mkdir -p $O; cd $R
{ for j in "soak_nn_big.py ${NB:-60}" "soak.py ${NS:-300}" "soak_fr.py ${NF:-1500}" "soak_nonfinite.py ${NN:-600}" "soak_misc.py ${NM:-500}" "soak_gc.py ${NG:-4000}"; do
  set -- $j; t0=$(date +%s); echo "== $1 $2"; timeout 1500 python tools/$1 $2 2>&1 | grep -v amdgpu.ids | tail -3; echo "   ($(( $(date +%s) - t0 )) s)"; done; } | tee $O/soak.txt
