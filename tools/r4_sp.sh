#!/bin/bash
# GPU box: single-pair kernel profile under workspace options (LIDARREG_OPTS)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp
for opts in "$@"; do
  export LIDARREG_OPTS="$opts"
  rm -rf /tmp/p_sp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_sp -o sp -- python3 $R/tools/single_pair_prof.py open3D MNN > /tmp/sp.log 2>&1
  f=$(find /tmp/p_sp -name '*kernel_stats.csv' | head -1)
  echo "== opts: $opts"
  python3 - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows:
    n=r['Name']
    if 'at::' in n or 'rocprim' in n or 'Cijk' in n: continue
    per=float(r['TotalDurationNs'])/40/1e3
    tot+=per
    if 'passb' in n or 'exact' in n or 'rev_' in n: print('  ', n[:44].ljust(44), int(r['Calls'])//40, round(per,1), ' avg', round(float(r['AverageNs'])/1e3,1))
print('   total us per pair', round(tot,1))
PY
done
