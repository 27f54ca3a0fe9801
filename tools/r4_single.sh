#!/bin/bash
# GPU box: single-pair filter pass (micro harness, P = 1) and FR() latency
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_single; mkdir -p $O; cd $R
if [ "$1" = "test" ]; then timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_fr_golden.py tests/test_gpu_soak.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest.txt; fi
{ echo "== r3 kernel, 5 strips"; tools/bin/pb_micro_r3 30000 1 5 | grep -E "need=2|walk only"
  for s in 4 5 6 7 8; do echo "== new kernel, $s strips"; tools/bin/pb_micro_new 30000 1 $s | grep -E "need=2|walk only"; done; } 2>&1 | tee $O/pb_micro_single.txt
python tools/fr_latency.py 2>&1 | grep -v amdgpu.ids | tee $O/fr_latency.txt
