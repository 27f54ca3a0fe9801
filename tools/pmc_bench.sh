#!/bin/bash
# GPU box: SQ counters per kernel of the batched bench workload, one --pmc pass per group -> gpurun_out/pmc_bench/summary.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=/tmp/pmc_bench_csv            # raw counter CSVs (10 MB per pass) stay on the box; only the summary is merged back
S=$R/gpurun_out/pmc_bench
rm -rf $O; mkdir -p $O $S
cd /tmp
k=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM"; do
  k=$((k+1))
  rm -rf /tmp/pb_$k
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pb_$k -o m -- python3 $R/bench.py --streams 1 --pairs 32 --steps 2 --warmup 1 --no-cpu-baseline "$@" > /tmp/pb_$k.log 2>&1
  f=$(find /tmp/pb_$k -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $O/g$k.csv; else echo "group $k FAILED"; tail -3 /tmp/pb_$k.log; fi
done
python3 - $O <<'PY' | tee $S/summary.txt
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
def short(n):
    m = re.match(r"^_Z(\d+)", n)
    if m: return n[m.end():m.end() + int(m.group(1))]
    return re.sub(r"^void ", "", n).split("(")[0].split("<")[0]
for f in sorted(glob.glob(sys.argv[1] + "/g*.csv")):
    for r in csv.DictReader(open(f)):
        n = short(r["Kernel_Name"])
        if n.startswith("at::") or "rocprim" in n or "Cijk" in n: continue
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("per-launch averages; quad = 4-cycle issue slots; a batched launch covers 32 pairs")
for n in sorted(acc, key=lambda n: -sum(acc[n].get("GRBM_GUI_ACTIVE", [0]))):
    c = {k: sum(v) / len(v) for k, v in acc[n].items()}
    if "GRBM_GUI_ACTIVE" not in c: continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0                       # summed over the 8 XCDs
    quads = cyc / 4.0 * 1024                               # issue slots of all 1024 SIMDs
    print(f"{n[:28]:28s} cycles {cyc:10.0f}  VALU-active {100*c.get('SQ_ACTIVE_INST_VALU',0)/quads:5.1f}%  MFMA-busy {100*c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(cyc*1024):5.1f}%  "
          f"wave-occupancy {c.get('SQ_WAVE_CYCLES',0)/quads:4.2f}/SIMD  wait-inst {100*c.get('SQ_WAIT_INST_ANY',0)/max(1,c.get('SQ_WAVE_CYCLES',1)):4.1f}%  "
          f"insts VALU {c.get('SQ_INSTS_VALU',0)/1e6:8.2f}M MFMA {c.get('SQ_INSTS_MFMA',0)/1e6:7.2f}M SALU {c.get('SQ_INSTS_SALU',0)/1e6:7.2f}M SMEM {c.get('SQ_INSTS_SMEM',0)/1e6:7.2f}M VMEM {c.get('SQ_INSTS_VMEM',0)/1e6:6.2f}M LDS {c.get('SQ_INSTS_LDS',0)/1e6:7.2f}M")
PY
