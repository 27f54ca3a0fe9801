"""GPU box: recall of the HIP pipeline on the list-driven surrogate under harder settings (to pick one where a regression can show)."""
import itertools
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lidarregistration_amd import harness
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import Args

stride = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
for ds in ("A", "B"):
    L = harness.load_list_fixture(ds)
    idx = list(range(0, len(L["session"]), stride))
    a = Args(mode="GPF", codebase="GC", iters=50000, prosac=True, GC_conf=0.999) if ds == "A" else Args(mode="MMN", codebase="GC", iters=1000000, prosac=True, GC_conf=0.9995)
    for rs, s, noise in [(0.5, 1.7, 0.15), (0.45, 1.7, 0.15), (0.45, 1.75, 0.15), (0.4, 1.7, 0.15), (0.4, 1.75, 0.15), (0.35, 1.8, 0.2), (0.3, 1.8, 0.2), (0.3, 1.85, 0.2), (0.3, 1.9, 0.2)]:
        r = harness.eval_list_batched(L, idx, a, n=n, s=s, batch=16, nstreams=2, rho_scale=rs, noise=noise)
        ok5 = (r["re_deg"] < 5) & (r["te_m"] < 0.6); ok2 = (r["re_deg"] < 2) & (r["te_m"] < 0.6)
        print(f"list {ds} rows {len(idx)} n={n} rho_scale {rs} s {s} noise {noise}: recall5 {ok5.mean():.3f} recall2 {ok2.mean():.3f}  n_corr mean {r['n_corr'].mean():.0f} ids mean {r['n_ids'].mean():.0f}  "
              f"RE ok-mean {r['re_deg'][ok5].mean() if ok5.any() else float('nan'):.3f} TE ok-mean cm {100 * r['te_m'][ok5].mean() if ok5.any() else float('nan'):.2f}", flush=True)
