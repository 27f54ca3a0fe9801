#!/bin/bash
# GPU box, round 6: the fused verification -- its tests, the whole GPU suite, the pipeline A/B against round 5's library, the micro-harness
# (as run on commit 8027fcb, where the fused verification and tests/test_gpu_fused_verify.py existed; today the test inputs live in tests/test_gpu_nn_strips.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_first; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_fused_verify.py -x -q 2>&1 | tail -15 | tee $O/fused_tests.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/gpu_suite.txt
LIBS="r5 shipped" REPS="1 2 3" tools/r4_ab.sh 2>&1 | tail -8 | tee $O/ab.txt
for b in r5 r6; do echo "== pb_micro_$b"; timeout 300 tools/bin/pb_micro_$b 30000 32 1 | grep -E "need=2 sample stride 16|walk only"; done 2>&1 | tee $O/pb_micro.txt
