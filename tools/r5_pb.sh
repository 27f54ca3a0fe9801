#!/bin/bash
# GPU box, round 5: filter-pass micro-harness variants side by side on one box (tools/pb_micro.hip built with different -D switches
# by tools/r5_build_pb.sh) -> gpurun_out/r5_pb/pb_micro.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_pb; mkdir -p $O; cd $R
for rep in 1 2; do
for b in ${PB:-new p1 geo p1geo}; do echo "== pb_micro_$b (rep $rep)"; timeout 300 tools/bin/pb_micro_$b 30000 32 1 | grep -v "stride  [248]:\|stride 64:\|need=1" ; done
done 2>&1 | tee $O/pb_micro.txt
for b in ${PBX:-x1 x2 x3}; do echo "== pb_micro_$b"; timeout 300 tools/bin/pb_micro_$b 30000 32 1 | grep -E "need=2 sample stride 16|walk only|shader|EXP" ; done 2>&1 | tee -a $O/pb_micro.txt
for b in ${PB1:-new p1geo}; do echo "== pb_micro_$b single pair, 6 strips"; timeout 300 tools/bin/pb_micro_$b 30000 1 6 | grep -E "need=2 sample stride  8|need=2 sample stride 16|walk only" ; done 2>&1 | tee -a $O/pb_micro.txt
