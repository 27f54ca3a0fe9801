#!/bin/bash
# GPU box: per-launch kernel trace of a list-driven run (which filter-pass instantiation does the work, launch by launch)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_listtrace; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
L=${1:-A}
rm -rf /tmp/lt
timeout 900 rocprofv3 --kernel-trace -d /tmp/lt -o lt --output-format csv -- python3 $R/bench.py --list $L --list-stride ${STRIDE:-8} --hard 0 --no-cpu-baseline > $O/line.txt 2>&1
f=$(find /tmp/lt -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/passb_launches.txt <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
for r in rows:
    n = r["Kernel_Name"]
    if "passb" in n or "nn16_range" in n or "nn16_prep" in n:
        tag = "T" if "ILb1E" in n else ("F" if "ILb0E" in n else n[:14])
        print(f'{(int(r["Start_Timestamp"])-t0)/1e3:12.1f} us  {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:9.1f} us  {tag:14s} grid {r["Grid_Size_X"]}x{r["Grid_Size_Y"]}x{r["Grid_Size_Z"]} wg {r["Workgroup_Size_X"]} stream {r.get("Stream_Id","?")} queue {r.get("Queue_Id","?")}')
P
head -80 $O/passb_launches.txt
