#!/bin/bash
# GPU box: the drop-in CLI through the batched engine -- tests, then the two README commands over the FULL balanced test lists
# (list-driven surrogate), end-to-end and registration-region pairs/s -> gpurun_out/r4_cli/
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_cli; mkdir -p $O; cd $R
if [ "$1" = "test" ]; then timeout 1200 python -m pytest tests/test_gpu_cli.py tests/test_gpu_gc.py -x -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt; fi
cd $R/Experiments
( time python -m test --dataset A --algo RANSAC --mode GPF --iters 50000 $CLI_EXTRA ) > $O/cli_A.log 2>&1; grep -E "process 0|recall|real" $O/cli_A.log
( time ./test_parallel.sh --dataset B --algo RANSAC --mode MNN --iters 1000000 --GC_conf 0.9995 $CLI_EXTRA ) > $O/cli_B.log 2>&1; grep -E "process 0|recall|real" $O/cli_B.log
( time python -m test --dataset A --algo RANSAC --mode GPF --iters 50000 --icp False $CLI_EXTRA ) > $O/cli_A_noicp.log 2>&1; grep -E "process 0|recall|real" $O/cli_A_noicp.log
cd $R; python bench.py --list A --no-cpu-baseline > $O/list_A_bench_line.json 2>/dev/null; cut -c1-300 $O/list_A_bench_line.json
rm -rf $R/Experiments/outputs
