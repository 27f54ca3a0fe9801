"""Development: phase times of ransac_lo_kernel's master block over a list-driven run (needs the -DLR_LO_PROBE build, tools/r4_loprobe.sh)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from lidarregistration_amd import _ext, harness   # noqa: E402

which, stride = sys.argv[1], int(sys.argv[2])
L = harness.load_list_fixture(which)
rows = list(range(0, len(L["session"]), stride))


class A:      # bench.py list_run's settings
    codebase = "GC"; prosac = True; fast_rejection = "ELC"; GC_LO = True; GPF_factor = 2.0; GPF_grid_wid = 10


if which == "A":
    A.mode, A.iters, A.GC_conf = "GPF", 50000, 0.999
else:
    A.mode, A.iters, A.GC_conf = "MMN", 1000000, 0.9995
lib = _ext.lib()
lib.lr_debug_lo_probe.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
for serial in (1, 6):
    lib.lr_debug_lo_probe(None, 1)
    res = harness.eval_list_batched(L, rows, A, n=30000, batch=32, nstreams=serial, device=torch.device("cuda", 0))
    torch.cuda.synchronize()
    lib.lr_debug_lo_probe(buf, 0)
    v = [int(x) for x in buf]
    n = len(rows)
    us = lambda t: t / 100.0
    print(f"list {which}: {n} pairs, {serial} call(s) in flight; LO launches mode0 {v[9]} (pair-blocks), final {v[10]}; rounds {v[8]} ({v[12]} scored over the near list only, {v[13]} near-list copies), polish iterations {v[11]}")
    names = ["build list", "sample + fit", "score 20", "select", "polish: build list", "polish: fit all", "polish: score 1"]
    for k, nm in enumerate(names):
        cnt = v[8] if k < 4 else v[11]
        print(f"   {nm:20s} {us(v[k]) / n:8.2f} us per pair   {us(v[k]) / max(cnt, 1):8.2f} us per round")
    print(f"   {'whole kernel (master)':20s} {us(v[7]) / n:8.2f} us per pair")
