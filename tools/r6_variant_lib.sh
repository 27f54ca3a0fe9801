#!/bin/bash
# usage: r6_variant_lib.sh <name> <sed-script> [file=lr_nn16.hip]: a copy of csrc/ with the sed script applied to one file, built as
# tools/bin/liblidarreg_<name>.so (for the LIDARREG_LIB A/B scripts: tools/r4_ab.sh, tools/r5_lists.sh, tools/fr_latency.py).  The shipped
# sources carry no experiment switches: a variant is a patch.
set -e
cd "$(dirname "$0")/.."; name=$1; script=$2; file=${3:-lr_nn16.hip}
T=$(mktemp -d); cp lidarregistration_amd/csrc/*.hip lidarregistration_amd/csrc/*.h $T/; mkdir -p $T/../../include 2>/dev/null || true
sed -i 's|#include "../../include/lidarreg.h"|#include "'$PWD'/include/lidarreg.h"|' $T/lr_internal.h
sed -i "$script" $T/$file
if cmp -s $T/$file lidarregistration_amd/csrc/$file; then echo "the sed script changed nothing" >&2; exit 1; fi
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 --offload-arch=gfx950 -fvisibility=hidden -w"
for f in lr_api lr_nn16 lr_filter lr_ransac lr_icp lr_voxel; do /opt/rocm/bin/hipcc $FLAGS -c $T/$f.hip -o $T/$f.o & done; wait
mkdir -p tools/bin; /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/liblidarreg_$name.so $T/*.o && rm -rf $T && ls -la tools/bin/liblidarreg_$name.so
