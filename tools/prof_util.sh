#!/bin/bash
# GPU box: derived utilisation counters of the bench kernels, one --pmc pass each (outputs under gpurun_out/util/)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/util
mkdir -p $O
cd /tmp
for c in MfmaUtil VALUBusy LDSBankConflict; do
  rocprofv3 --pmc $c --output-format csv -d /tmp/p_$c -o u -- python3 $R/bench.py --steps 2 --warmup 1 --streams 1 --pairs 8 --no-cpu-baseline > /tmp/p_$c.log 2>&1
  f=$(find /tmp/p_$c -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $O/$c.csv; echo "$c ok $(wc -l < $f)"; else echo "$c FAILED"; tail -3 /tmp/p_$c.log; fi
done
