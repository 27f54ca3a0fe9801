#!/bin/bash
# GPU box: HBM traffic of one bench workload -> profiles-style JSON keyed by the workload (bench.py reports `traffic` only for a matching key).
#   usage: pmc_traffic.sh <commit> <out.json> [bench flags, e.g. --n 100000 | --mode GPF | --codebase GC]
# Two separate --pmc passes (FETCH_SIZE, WRITE_SIZE), --kernel-trace-free, as MI355X_MICROARCH.md prescribes; one batched call in flight.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$1; OUT=$(realpath -m "$2"); shift 2
KEY=$(python3 $R/bench.py "$@" --traffic-key); PPL=${KEY##*pairs_per_launch=}
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pt_$c -o c -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --streams 1 --pairs $PPL --no-cpu-baseline --sustain-s 0 > /tmp/pt_$c.log 2>&1
  f=$(find /tmp/pt_$c -name '*counter_collection.csv' | head -1)
  if [ -z "$f" ]; then echo "$c pass FAILED"; tail -5 /tmp/pt_$c.log; exit 1; fi
  cp "$f" /tmp/ptc_$c.csv
done
python3 $R/tools/pmc_to_json.py /tmp/ptc_FETCH_SIZE.csv /tmp/ptc_WRITE_SIZE.csv $OUT "$C" "$KEY" $PPL | tail -3
