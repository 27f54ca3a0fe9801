#!/bin/bash
# GPU box, round 6: ordering kernel with 8 keys in flight -- suite, three-way pipeline A/B (round 5 / closed-form Kabsch with the old three
# ordering kernels / shipped), single-stream kernel summary, HBM traffic of the 100k workload with two column strips, the full bench line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_fourth; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/gpu_suite.txt
LIBS="r5 k3 shipped" REPS="1 2 3" tools/r4_ab.sh 2>&1 | tail -12 | tee $O/ab.txt
cd /tmp; rm -rf /tmp/p_s1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_s1 -o s1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --pairs 64 --steps 3 --warmup 1 --sustain-s 0 > /tmp/p_s1.log 2>&1
cp "$(find /tmp/p_s1 -name '*kernel_stats.csv' | head -1)" $O/bench_streams1_kernel_stats.csv
cd $R
bash tools/single_pair_prof.sh 2>&1 | grep -v "^$" | head -16 | tee $O/single_pair.txt
timeout 900 tools/pmc_traffic.sh r6-strips2 $O/pmc_100k_strips2.json --n 100000 --opt nn_blocks_batch=6256 2>&1 | tail -5 | tee $O/pmc_100k_strips2.txt
timeout 900 tools/pmc_traffic.sh r6-strips1 $O/pmc_100k_strips1.json --n 100000 2>&1 | tail -5 | tee $O/pmc_100k_strips1.txt
python bench.py > $O/bench_line.json 2> $O/bench_stderr.log; python - $O/bench_line.json <<'PY' | tee $O/bench_fields.txt
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "ms_per_step_minmedmax", "clock_MHz", "value_at_2.0GHz", "sustained")})
c = d["cpu_baseline"]; print({k: c.get(k) for k in ("value", "cores", "value_process_parallel", "process_parallel", "host_hardware_threads", "oracle_port_pairs_per_s")}, d.get("speedup_vs_cpu_baseline"))
print(json.dumps(d.get("extra"))[:1500])
PY
