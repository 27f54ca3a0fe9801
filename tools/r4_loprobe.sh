#!/bin/bash
# container: build tools/bin/liblidarreg_probe.so (lr_ransac.hip with -DLR_LO_PROBE); GPU box (argument "run"): list runs with it
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/lidarregistration_amd/csrc
if [ "$1" != "run" ]; then
  mkdir -p $R/tools/bin; make -C $C -s -j4 liblidarreg.so
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 --offload-arch=gfx950 -fvisibility=hidden -DLR_LO_PROBE -c $C/lr_ransac.hip -o /tmp/lr_ransac_probe.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/bin/liblidarreg_probe.so $C/lr_api.o $C/lr_nn16.o $C/lr_filter.o /tmp/lr_ransac_probe.o $C/lr_icp.o $C/lr_voxel.o && echo built
  exit
fi
O=$R/gpurun_out/r4_loprobe; mkdir -p $O; cd $R
export LIDARREG_LIB=$R/tools/bin/liblidarreg_probe.so
for L in ${LISTS:-A B}; do python3 tools/lo_probe.py $L ${STRIDE:-8}; done 2>&1 | grep -v amdgpu.ids | tee $O/probe.txt
