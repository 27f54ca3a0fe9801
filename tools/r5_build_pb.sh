#!/bin/bash
# build the micro-harness variants tools/r5_pb.sh runs (here, in the build container: hipcc cross-compiles)
#   usage: r5_build_pb.sh name:"-Dflags" ...
cd "$(dirname "$0")/.."; mkdir -p tools/bin
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form=1 -w"
for spec in "$@"; do name=${spec%%:*}; flags=${spec#*:}; $CC $flags tools/pb_micro.hip -o tools/bin/pb_micro_$name & done; wait
ls -la tools/bin/pb_micro_* | awk '{print $5, $9}'
