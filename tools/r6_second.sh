#!/bin/bash
# GPU box, round 6: closed-form Kabsch eigenvector (oracle + kernels) -- the whole GPU suite, pipeline A/B against round 5's library, lists A/B,
# kernel summary of the single-stream bench, and two zero-code sweeps: column strips at 100k points, strips / reverse strips of the headline
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6_second; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/gpu_suite.txt
LIBS="r5 shipped" REPS="1 2 3" tools/r4_ab.sh 2>&1 | tail -8 | tee $O/ab.txt
LIBS="r5 shipped" REPS="1 2" tools/r5_lists.sh 2>&1 | tail -10 | tee $O/lists.txt
cd /tmp; rm -rf /tmp/p_s1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_s1 -o s1 -- python3 $R/bench.py --no-cpu-baseline --streams 1 --pairs 64 --steps 3 --warmup 1 --sustain-s 0 > /tmp/p_s1.log 2>&1
cp "$(find /tmp/p_s1 -name '*kernel_stats.csv' | head -1)" $O/bench_streams1_kernel_stats.csv
cd $R
for nb in 0 6256 9384 12512; do
  python bench.py --n 100000 --pairs 32 --steps 3 --warmup 1 --no-cpu-baseline --sustain-s 0 --opt nn_blocks_batch=$nb > $O/n100k_$nb.json 2>/dev/null
  python - $O/n100k_$nb.json $nb <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"100k nn_blocks_batch={sys.argv[2]:6s} pairs/s {d['value']:8.1f} fwd {r['forward_launch_ms']:.3f} rev {r['reverse_launch_ms']:.3f} call {r['call_ms']:.3f} frac {r['frac']:.4f} clock {d.get('clock_MHz')}")
PY
done 2>&1 | tee $O/n100k_strips.txt
python bench.py > $O/bench_line.json 2> $O/bench_stderr.log; python - $O/bench_line.json <<'PY' | tee $O/bench_fields.txt
import json, sys
d = json.load(open(sys.argv[1]))
print({k: d.get(k) for k in ("value", "ms_per_step", "ms_per_step_minmedmax", "clock_MHz", "value_at_2.0GHz", "sustained")})
c = d["cpu_baseline"]; print({k: c.get(k) for k in ("value", "cores", "value_process_parallel", "process_parallel", "host_hardware_threads", "oracle_port_pairs_per_s")}, d.get("speedup_vs_cpu_baseline"))
PY
