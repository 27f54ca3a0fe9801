"""Development: what runs when.  Reads a rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp per dispatch) of a multi-stream run
and prints: the window between the first and last filter-pass launch, the share of it in which NO kernel of the library runs (idle
gaps), in which exactly k kernels overlap, and per kernel the time during which it runs ALONE vs next to others.

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python3 bench.py --list A --list-stride 8 --hard 0 --no-cpu-baseline
    python tools/timeline.py /tmp/tl/.../tl_kernel_trace.csv
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
own = [r for r in rows if not r["Kernel_Name"].startswith(("at::", "void at::", "rocprim", "void rocprim", "__amd", "Cijk", "void (anonymous"))]
ev = []
for r in own:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:40]
    ev.append((s, e, name))
pb = [x for x in ev if "passb" in x[2]]
t0, t1 = min(x[0] for x in pb), max(x[1] for x in pb)
ev = [x for x in ev if x[1] > t0 and x[0] < t1]
pts = []
for s, e, n in ev:
    pts.append((max(s, t0), 1, n)); pts.append((min(e, t1), -1, n))
pts.sort(key=lambda p: (p[0], -p[1]))
depth = 0
last = t0
hist = defaultdict(int)
active = defaultdict(int)
alone = defaultdict(int); shared = defaultdict(int)
for t, d, n in pts:
    dt = t - last
    if dt > 0:
        hist[depth] += dt
        names = [k for k, v in active.items() if v > 0]
        for k in names:
            (alone if depth == 1 else shared)[k] += dt
    last = t
    depth += d
    active[n] += d
W = t1 - t0
print(f"window {W / 1e6:.1f} ms, {len(ev)} dispatches of the library's kernels")
for k in sorted(hist):
    print(f"   {k} kernels in flight: {100 * hist[k] / W:5.1f} %")
print("per kernel: ms alone / ms next to others")
for k in sorted(set(alone) | set(shared), key=lambda k: -(alone[k] + shared[k]))[:14]:
    print(f"   {k:42s} {alone[k] / 1e6:8.2f} {shared[k] / 1e6:8.2f}")
