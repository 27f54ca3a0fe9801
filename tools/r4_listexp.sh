#!/bin/bash
# GPU box: list-driven run under a few settings (streams in flight, LO helper groups), interleaved repeats
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_listexp; mkdir -p $O; cd $R
run() { echo "== list ${L:-A} streams $S $*"; env "$@" python3 bench.py --list ${L:-A} --list-stride ${STRIDE:-1} --hard 0 --no-cpu-baseline --streams $S 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); t = d['time_per_pair_us']
print('   pairs/s %.0f  throughput %.1f us  serialised %.1f  fwd nn %.1f  in flight %s' % (d['value'], t['throughput'], t['whole_call_serialised'], t['forward_nn'], d['config']['batched_calls_in_flight_per_gpu']))"; }
for rep in 1 2; do
  for S in ${STREAMS:-4 6}; do for g in ${GROUPS_:-1 2 4 8}; do run LIDARREG_EXP_LO_GROUPS=$g; done; done
done 2>&1 | tee $O/exp_${L:-A}.txt
