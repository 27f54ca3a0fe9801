#!/bin/bash
# GPU box, round 6: the shipped state -- fused verification removed again, closed-form Kabsch, one ordering kernel, device checks: the whole suite,
# pipeline A/B against round 5's library, lists, single-stream kernel summary, single pair, the bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_fifth; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/gpu_suite.txt
LIBS="r5 shipped" REPS="1 2 3 4" tools/r4_ab.sh 2>&1 | tail -12 | tee $O/ab.txt
LIBS="r5 shipped" REPS="1 2" tools/r5_lists.sh 2>&1 | tail -10 | tee $O/lists.txt
cd /tmp; rm -rf /tmp/p_s1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_s1 -o s1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --pairs 64 --steps 3 --warmup 1 --sustain-s 0 > /tmp/p_s1.log 2>&1
cp "$(find /tmp/p_s1 -name '*kernel_stats.csv' | head -1)" $O/bench_streams1_kernel_stats.csv
cd $R
bash tools/single_pair_prof.sh 2>&1 | grep -v "^$\|rocprim\|at::" | tee $O/single_pair.txt
python tools/fr_latency.py 2>/dev/null | tee $O/fr_latency.txt
for b in r5 r6; do echo "== pb_micro_$b"; timeout 300 tools/bin/pb_micro_$b 30000 32 1 | grep -E "need=2 sample stride 16|walk only"; done 2>&1 | tee $O/pb_micro.txt
python bench.py > $O/bench_line.json 2> $O/bench_stderr.log; python - $O/bench_line.json <<'PY' | tee $O/bench_fields.txt
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "ms_per_step_minmedmax", "clock_MHz", "value_at_2.0GHz", "sustained")})
c = d["cpu_baseline"]; print({k: c.get(k) for k in ("value", "cores", "value_process_parallel", "process_parallel", "host_hardware_threads", "oracle_port_pairs_per_s")}, d.get("speedup_vs_cpu_baseline"))
e = (d.get("extra") or {}).get("list_A") or {}; print({k: e.get(k) for k in ("value", "pairs", "recall_5deg_0.6m", "hard")})
PY
