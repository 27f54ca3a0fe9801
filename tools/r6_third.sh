#!/bin/bash
# GPU box, round 6: one ordering kernel for the reverse pass + closed-form Kabsch: the whole GPU suite, pipeline and list A/B against round 5's
# library, the single-pair kernels and FR() latency (shipped / tightening mark staggered per block), the bench line with its new fields
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_third; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/gpu_suite.txt
LIBS="r5 shipped" REPS="1 2 3" tools/r4_ab.sh 2>&1 | tail -8 | tee $O/ab.txt
LIBS="r5 shipped" REPS="1 2" tools/r5_lists.sh 2>&1 | tail -10 | tee $O/lists.txt
for lib in shipped stagger; do
  if [ $lib = shipped ]; then unset LIDARREG_LIB; else export LIDARREG_LIB=$R/tools/bin/liblidarreg_$lib.so; fi
  echo "==== $lib"; bash tools/single_pair_prof.sh 2>&1 | grep -v "^$"; python tools/fr_latency.py 2>/dev/null | head -2
done 2>&1 | tee $O/single_pair.txt
unset LIDARREG_LIB
cd /tmp; rm -rf /tmp/p_s1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_s1 -o s1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --pairs 64 --steps 3 --warmup 1 --sustain-s 0 > /tmp/p_s1.log 2>&1
cp "$(find /tmp/p_s1 -name '*kernel_stats.csv' | head -1)" $O/bench_streams1_kernel_stats.csv
cd $R
python bench.py > $O/bench_line.json 2> $O/bench_stderr.log; python - $O/bench_line.json <<'PY' | tee $O/bench_fields.txt
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "ms_per_step_minmedmax", "clock_MHz", "value_at_2.0GHz", "sustained")})
c = d["cpu_baseline"]; print({k: c.get(k) for k in ("value", "cores", "value_process_parallel", "process_parallel", "host_hardware_threads", "oracle_port_pairs_per_s")}, d.get("speedup_vs_cpu_baseline"))
PY
