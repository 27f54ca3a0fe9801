#!/bin/bash
# GPU box: SQ counters of the batched pass-B / pass-A kernels on the micro harness (tools/pb_micro.hip), one --pmc pass per group.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_micro
mkdir -p $O
cd /tmp
k=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY"; do
  k=$((k+1))
  rm -rf /tmp/pm_$k
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pm_$k -o m -- $R/tools/bin/pb_micro_0 30000 32 1 > /tmp/pm_$k.log 2>&1
  f=$(find /tmp/pm_$k -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $O/g$k.csv; else echo "group $k FAILED"; tail -3 /tmp/pm_$k.log; fi
done
python3 - $O <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/g*.csv")):
    rows = list(csv.DictReader(open(f)))
    byk = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        name = "passa" if "passa" in r["Kernel_Name"] else ("passb" if "passb" in r["Kernel_Name"] else None)
        if name: byk[name][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for name, cs in byk.items():
        for c, vals in cs.items():
            vals.sort()
            v = [x[1] for x in vals]
            if name == "passb":          # dispatch groups: 11 x need 2, 11 x need 1, 11 x no candidates
                acc["passb need2"][c] = sum(v[2:11]) / 9; acc["passb need1"][c] = sum(v[13:22]) / 9; acc["passb nocand"][c] = sum(v[24:33]) / max(1, len(v[24:33]))
            else:
                acc["passa"][c] = sum(v[2:]) / max(1, len(v[2:]))
for name in sorted(acc):
    print("==", name)
    for c in sorted(acc[name]): print(f"   {c:28s} {acc[name][c]:16.0f}")
PY
