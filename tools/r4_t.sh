#!/bin/bash
# GPU box: run the given pytest selection
R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out/r4_t; cd $R
timeout ${T:-1500} python -m pytest "$@" -x -q -m gpu -s 2>&1 | tail -${TAIL:-25} | tee $R/gpurun_out/r4_t/pytest.txt
