"""One worker process of bench.py's process-parallel CPU baseline -- TEST INFRASTRUCTURE / CPU BASELINE ONLY (like everything under oracle/).

The reference shards its pair list over OS processes (Experiments/test_parallel.sh:18-20, one per GPU); its host-cores equivalent is P
processes of the reference-style path (oracle/torch_cpu.py), each with its own torch / OpenMP thread pool.  bench.py starts P of these,
waits until all say READY (imports, thread pools and one small pair are untimed), sends GO to all at once and takes the wall time until
the last one reports.

    python -m oracle.cpu_worker <threads> <n_points> <mode> <iters> <pairs> <seed0> [<cpu list "0,1,2" or "-">]
"""
import sys
import time


def main(argv):
    threads, n, mode, iters, pairs, seed0 = int(argv[0]), int(argv[1]), argv[2], int(argv[3]), int(argv[4]), int(argv[5])
    if len(argv) > 6 and argv[6] != "-":          # this worker's own cores, before any thread pool exists
        import os
        os.sched_setaffinity(0, {int(c) for c in argv[6].split(",")})
    import torch
    torch.set_num_threads(threads)
    from lidarregistration_amd import synth
    from oracle import oracle as orc, torch_cpu
    orc.build()
    w = synth.make_pair(N=2000, seed=1)
    torch_cpu.register_pair(w["xyz0"], w["xyz1"], w["feats0"], w["feats1"], mode=mode, iters=1000)          # page-in / thread pools: untimed
    data = [synth.make_pair(N=n, seed=seed0 + k) for k in range(pairs)]
    print("READY", flush=True)
    if sys.stdin.readline().strip() != "GO":
        return 1
    t0 = time.perf_counter()
    for p in data:
        torch_cpu.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode=mode, iters=iters, sample_size=3, seed=51)
    print("DONE %d %.6f" % (pairs, time.perf_counter() - t0), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
