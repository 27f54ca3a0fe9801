#!/bin/bash
# The ablation variants of the filter pass behind profiles/r05_pb_ablation.txt were -D switches inside lr_nn16.hip through round 5
# (LR_PB_EXP bits, LR_PB_GEO, LR_PB_JOINT, LR_PB_P1FOLD, LR_PB_WAVES, LR_PB_CH).  The shipped file no longer carries them; this script
# rebuilds any of them from the source of the commit that measured them:
#   usage: pb_variant.sh name "-DLR_PB_EXP=4 ..." [commit=8207e5a]   ->  tools/bin/pb_micro_<name>
# New variants of the CURRENT kernel: copy lr_nn16.hip to tools/bin/, patch it (sed / patch), build with -DPB_SRC='"bin/<copy>.hip"'.
set -e
cd "$(dirname "$0")/.."; mkdir -p tools/bin
name=$1; flags=$2; commit=${3:-8207e5a}
git show $commit:lidarregistration_amd/csrc/lr_nn16.hip | sed 's|#include "lr_internal.h"|#include "../../lidarregistration_amd/csrc/lr_internal.h"|' > tools/bin/lr_nn16_$commit.hip
git show $commit:tools/pb_micro.hip > tools/bin/pb_micro_$commit.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -w $flags \
    -DPB_SRC="\"lr_nn16_$commit.hip\"" tools/bin/pb_micro_$commit.hip -o tools/bin/pb_micro_$name
ls -la tools/bin/pb_micro_$name
