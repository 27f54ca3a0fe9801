#!/bin/bash
# usage: r5_buildlib.sh <name> [-D...]: an alternative build of the library -> tools/bin/liblidarreg_<name>.so (for tools/r4_ab*.sh, LIDARREG_LIB)
cd "$(dirname "$0")/../lidarregistration_amd/csrc"; name=$1; shift
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 --offload-arch=gfx950 -fvisibility=hidden"
T=$(mktemp -d); for f in lr_api lr_nn16 lr_filter lr_ransac lr_icp lr_voxel; do /opt/rocm/bin/hipcc $FLAGS "$@" -c $f.hip -o $T/$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/liblidarreg_$name.so $T/*.o && rm -rf $T && ls -la ../../tools/bin/liblidarreg_$name.so
