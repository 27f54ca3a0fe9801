#!/bin/bash
# GPU box: filter-pass micro-harness (old r3 kernel next to the current one); with "test": NN parity subset first
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_quick; mkdir -p $O; cd $R
if [ "$1" = "test" ]; then timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_fr_golden.py tests/test_gpu_soak.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt; fi
for b in ${PB:-r3 new}; do echo "== pb_micro_$b"; timeout 300 tools/bin/pb_micro_$b 30000 32 1 | grep -v "stride  [248]:\|stride 64:" ; done 2>&1 | tee $O/pb_micro.txt
for b in ${PB1:-new}; do echo "== pb_micro_$b single pair, 6 strips"; timeout 300 tools/bin/pb_micro_$b 30000 1 6 | grep -E "need=2 sample stride  8|need=2 sample stride 16|walk only" ; done 2>&1 | tee -a $O/pb_micro.txt
