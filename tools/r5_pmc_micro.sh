#!/bin/bash
# GPU box, round 5: SQ counters of the candidate-free walk (real accumulators) of micro-harness variants, one --pmc pass per group
#   usage: r5_pmc_micro.sh name1 name2 ...   (tools/bin/pb_micro_<name>)  -> gpurun_out/r5_pmc/summary.txt
export TMPDIR=/tmp PB_ONLY=nohit
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5_pmc; mkdir -p $O; cd /tmp
for name in "$@"; do
k=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_COEXEC_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INSTS_VMEM"; do
  k=$((k+1)); rm -rf /tmp/pm_$k
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pm_$k -o m -- $R/tools/bin/pb_micro_$name 30000 32 1 > /tmp/pm_$k.log 2>&1
  f=$(find /tmp/pm_$k -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" /tmp/pmg_${name}_$k.csv; else echo "$name group $k FAILED"; tail -3 /tmp/pm_$k.log; fi
done
done
python3 - "$@" <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
names = sys.argv[1:]
tab = collections.OrderedDict()
for name in names:
    acc = {}
    for f in sorted(glob.glob(f"/tmp/pmg_{name}_*.csv")):
        rows = [r for r in csv.DictReader(open(f)) if "nn16_passb" in r["Kernel_Name"] and ("ILb1E" in r["Kernel_Name"] or "<true>" in r["Kernel_Name"])]
        by = collections.defaultdict(list)
        for r in rows: by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for c, v in by.items():
            v.sort(); v = [x[1] for x in v][-8:]          # the timed repetitions of the candidate-free walk
            acc[c] = sum(v) / len(v)
    tab[name] = acc
keys = sorted({k for a in tab.values() for k in a})
print("per MFMA instruction (SQ_INSTS_MFMA), candidate-free walk on real accumulators, 32 pairs x 30k")
print(f"{'counter':30s}" + "".join(f"{n:>14s}" for n in names))
for k in keys:
    print(f"{k:30s}" + "".join(f"{tab[n].get(k, float('nan')) / max(tab[n].get('SQ_INSTS_MFMA', 1), 1):14.4f}" for n in names))
print(f"{'GRBM_GUI_ACTIVE/8 (cycles)':30s}" + "".join(f"{tab[n].get('GRBM_GUI_ACTIVE', 0) / 8:14.0f}" for n in names))
PY
