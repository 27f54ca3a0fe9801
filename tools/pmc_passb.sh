#!/bin/bash
# GPU box: more SQ counters for the filter pass (co-execution of the vector and matrix pipes, LDS, waits, scalar / branch issue)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=/tmp/pmc_passb_csv            # raw counter CSVs (10 MB per pass) stay on the box; only the summary is merged back
S=$R/gpurun_out/pmc_passb
rm -rf $O; mkdir -p $O $S
cd /tmp
k=0
for grp in "SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU"; do
  k=$((k+1))
  rm -rf /tmp/pp_$k
  rocprofv3 --pmc $grp --output-format csv -d /tmp/pp_$k -o m -- python3 $R/bench.py --streams 1 --pairs 32 --steps 2 --warmup 1 --no-cpu-baseline --sustain-s 0 > /tmp/pp_$k.log 2>&1
  f=$(find /tmp/pp_$k -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $O/g$k.csv; else echo "group $k FAILED"; tail -3 /tmp/pp_$k.log; fi
done
python3 - $O <<'PY' | tee $S/summary.txt
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/g*.csv")):
    # (the kernel has two instantiations, both launched; on the bench's unit-norm descriptors the blocks of <false> return at once)
    rows = [r for r in csv.DictReader(open(f)) if "nn16_passb" in r["Kernel_Name"] and ("ILb1E" in r["Kernel_Name"] or "<true>" in r["Kernel_Name"])]
    # launches alternate forward, reverse
    by = collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in by.items():
        acc["fwd"][c] = sum(v[0::2]) / max(1, len(v[0::2])); acc["rev"][c] = sum(v[1::2]) / max(1, len(v[1::2]))
for d in ("fwd", "rev"):
    c = acc[d]
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    quads = cyc / 4.0 * 1024
    print(f"== filter pass {d}: cycles {cyc:.0f}")
    for k in sorted(c):
        v = c[k]
        print(f"   {k:32s} {v:16.0f}   per SIMD-quad {v / max(quads, 1):7.3f}   per MFMA {v / max(c.get('SQ_INSTS_MFMA', 1), 1):7.3f}")
PY
