#!/bin/bash
# GPU box: full -m gpu suite, default bench line, SQ counters of the filter pass
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_check; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest.txt
python bench.py > $O/bench_line.json 2> $O/bench_stderr.log; cut -c1-400 $O/bench_line.json
bash tools/pmc_passb.sh > $O/pmc_passb_summary.txt 2>&1; grep -E "== filter|SQ_INSTS_VALU|SQ_INSTS_MFMA|MFMA_BUSY|GRBM" $O/pmc_passb_summary.txt
