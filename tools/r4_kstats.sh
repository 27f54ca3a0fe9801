#!/bin/bash
# GPU box: single-stream kernel summary of the bench (one batched call in flight) -> per-kernel average durations
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_kstats; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- python3 $R/bench.py --no-cpu-baseline --streams 1 --pairs 64 --steps 3 --warmup 1 --sustain-s 0 "$@" > $O/line.txt 2>&1
cp "$(find /tmp/ks -name '*kernel_stats.csv' | head -1)" $O/kernel_stats.csv
python3 - $O/kernel_stats.csv <<'P'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("at::", "rocprim", "Cijk", "rocclr")): continue
    tot += float(r["TotalDurationNs"])
    print(f"{n.split('(')[0][:44]:44s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
print("library kernels, total ms", tot / 1e6)
P
