#!/bin/bash
# GPU box, round 5: bench.py --list A|B over calls in flight x pairs per call, two repetitions (-> docs/HISTORY.md)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for cfg in "6 32" "8 32" "6 48" "8 48" "10 32" "5 32"; do set -- $cfg
  for L in A B; do
  v=$(python bench.py --list $L --streams $1 --batch $2 --hard 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
  echo "rep $rep streams $1 batch $2 list $L: $v"
  done
done; done
