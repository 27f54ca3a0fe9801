#!/bin/bash
# GPU box: filter-pass micro-harness (old r3 kernel next to the current one); with "test": NN parity subset first
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_quick; mkdir -p $O; cd $R
if [ "$1" = "test" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_fr_golden.py tests/test_gpu_soak.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt; fi
for b in ${PB:-r3 new new16}; do echo "== pb_micro_$b"; timeout 300 tools/bin/pb_micro_$b 30000 32 1 | grep -v "stride  [248]:\|stride 64:" ; done 2>&1 | tee $O/pb_micro.txt
