export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for cfg in "open3D MNN" "GC MNN"; do
  set -- $cfg
  rm -rf /tmp/p_sp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_sp -o sp -- python3 $R/tools/single_pair_prof.py $1 $2 > /tmp/sp.log 2>&1
  f=$(find /tmp/p_sp -name '*kernel_stats.csv' | head -1)
  echo "== $cfg"
  python3 - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=0
for r in rows:
    n=r['Name']
    if 'at::' in n or 'rocprim' in n or 'Cijk' in n: continue
    per=float(r['TotalDurationNs'])/40/1e3
    tot+=per
    if per>2: print(n[:50].ljust(50), int(r['Calls'])//40, round(per,1))
print('total us per pair', round(tot,1))
PY
done
