"""Soak (not part of the suite): the HARD list-driven surrogate on other rows than tests/test_gpu_lists.py takes -- the HIP pipeline and the
oracle pipeline must succeed and fail on exactly the same rows, and agree on the transform where both succeed.
usage: python tools/soak_hard.py A|B stride offset"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from lidarregistration_amd import harness, metrics, synth      # noqa: E402
from oracle import oracle                                       # noqa: E402
from tests.conftest import Args, gc_oracle_kwargs               # noqa: E402

dataset, stride, offset = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
a = Args(mode="GPF", codebase="GC", iters=50000, prosac=True, GC_conf=0.999) if dataset == "A" else Args(mode="MMN", codebase="GC", iters=1000000, prosac=True, GC_conf=0.9995)
L = harness.load_list_fixture(dataset)
idx = list(range(offset, len(L["session"]), stride))
hs = harness.HARD[dataset]
t0 = time.time()
res = harness.eval_list_batched(L, idx, a, n=30000, batch=8, nstreams=2, **hs)
ok_hip = (res["re_deg"] < metrics.RE_THRE_DEG) & (res["te_m"] * 100 < metrics.TE_THRE_CM)
ok_orc = np.zeros(len(idx), bool)
worst = [0.0, 0.0]
for j, k in enumerate(idx):
    rho = float(np.clip(L["overlap"][k], 0.05, 0.95)) * hs["rho_scale"]
    p = synth.make_pair_dev(N=30000, rho=rho, s=hs["s"], seed=51 + k, device=torch.device("cuda", 0), T_gt=L["T_gt"][k], noise=hs["noise"])
    h = {q: p[q].cpu().numpy() for q in ("xyz0", "xyz1", "feats0", "feats1")}
    e = oracle.register_pair(h["xyz0"], h["xyz1"], h["feats0"], h["feats1"], mode=a.mode, iters=a.iters, seed=51, args=a, **gc_oracle_kwargs(a))
    ok_orc[j] = metrics.is_success(e["T"], L["T_gt"][k])
    if ok_orc[j] and ok_hip[j]:
        dr, dt = np.radians(oracle.rotation_error_deg(res["T"][j], e["T"])), oracle.translation_error_cm(res["T"][j], e["T"]) / 100
        worst = [max(worst[0], dr), max(worst[1], dt)]
        assert dr <= 1e-4 and dt <= 1e-3, (k, dr, dt)
assert np.array_equal(ok_hip, ok_orc), (np.flatnonzero(ok_hip != ok_orc), res["re_deg"], res["te_m"])
print(f"hard soak ok: list {dataset}, rows {offset}::{stride} ({len(idx)} rows), recall HIP = oracle = {ok_orc.mean():.3f} ({int((~ok_orc).sum())} fail on both), "
      f"largest |dR| {worst[0]:.1e} rad |dt| {worst[1]:.1e} m where both succeed, {time.time() - t0:.0f} s")
