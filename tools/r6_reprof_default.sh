export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/round6b; mkdir -p $O; cd /tmp
prof() { tag=$1; shift; rm -rf /tmp/p_$tag; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$tag -o $tag -- python3 $R/bench.py --no-cpu-baseline "$@" > /tmp/p_$tag.log 2>&1; cp "$(find /tmp/p_$tag -name '*kernel_stats.csv' | head -1)" $O/${tag}_kernel_stats.csv; }
prof bench_default --steps 3 --warmup 1 --sustain-s 0
prof bench_sustained --steps 10 --warmup 2 --sustain-s 10
grep -o '"sustained": {[^}]*}' /tmp/p_bench_sustained.log > $O/bench_sustained_under_profiler.txt
cat $O/bench_sustained_under_profiler.txt
