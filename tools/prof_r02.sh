#!/bin/bash
# GPU box: per-kernel summary of the batched bench workload (one batched call in flight, so kernel times do not overlap)
#   usage: bash tools/prof_r02.sh <tag> [extra bench args]      (env vars pass through)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-s1}; shift
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp
rm -rf /tmp/p_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG -o $TAG -- python3 $R/bench.py --streams 1 --pairs 64 --steps 3 --warmup 1 --no-cpu-baseline "$@" > /tmp/p_$TAG.log 2>&1
f=$(find /tmp/p_$TAG -name '*kernel_stats.csv' | head -1)
t=$(find /tmp/p_$TAG -name '*kernel_trace.csv' | head -1)
cp "$f" $O/${TAG}_kernel_stats.csv
echo "== $TAG $@"
grep "^{" /tmp/p_$TAG.log | tail -1 > /tmp/p_$TAG.json
python3 - /tmp/p_$TAG.json <<'PY'
import sys, json
try:
    l = json.load(open(sys.argv[1]))
    print('value', l['value'], 'redone', l.get('nn_rows_redone_by_full_scan_per_pair'), 'roof', l['roofline']['frac'], 'launch_ms', l['roofline']['launch_ms'], 'pair', l['pair_roofline'])
except Exception as e:
    print('no json', e)
PY
python3 - "$f" "$t" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n = r['Name']
    if n.startswith('void at::') or 'rocprim' in n or 'Cijk' in n or 'rocclr' in n: continue
    print(f"{n[:48]:48s} calls {int(r['Calls']):5d} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f} pct {float(r['Percentage']):6.2f}")
# forward / reverse split of pass B: launches alternate forward, reverse
tr = [r for r in csv.DictReader(open(sys.argv[2])) if 'nn16_passb' in r['Kernel_Name']]
tr.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in tr]
if d:
    f, b = d[0::2], d[1::2]
    print(f"passb forward avg_us {sum(f)/len(f):.1f}  reverse avg_us {sum(b)/len(b):.1f}  (launches {len(d)})")
PY
