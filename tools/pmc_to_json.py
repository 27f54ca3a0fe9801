"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/pmc_traffic.json.

    python tools/pmc_to_json.py pmc_fetch.csv pmc_write.csv profiles/pmc_traffic.json [commit] [workload key] [pairs per launch]

The output file is keyed by workload (`python bench.py <flags> --traffic-key`): {"workloads": {key: {kernel: ..., "_pair": ..., "_meta": ...}}};
a run for one workload replaces that entry and leaves the others.  (A file from before round 5 -- one unkeyed workload, the default
one -- is converted on the way.)

hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: both counters are KiB, and on gfx950 FETCH_SIZE reports half of a wide
coalesced read (/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section).
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    m = re.match(r"^_Z(\d+)", name)          # rocprofv3 leaves some names mangled: _Z<len><name>...
    if m:
        base = name[m.end():m.end() + int(m.group(1))]
        rest = name[m.end() + int(m.group(1)):]
        t = re.match(r"^I((?:Lb[01]E)+)E", rest)   # bool template arguments: the instantiations of the filter pass are different kernels
        return base + "<" + ",".join("true" if b == "1" else "false" for b in re.findall(r"Lb([01])E", t.group(1))) + ">" if t else base
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    return re.sub(r"<.*$", "", name) if name.startswith("at::") else name


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}, {k: v[1] for k, v in acc.items()}


DEFAULT_KEY = "n=30000,mode=MNN,codebase=open3D,iters=50000,pairs_per_launch=32"


def main(fetch_csv, write_csv, out, commit="commit unrecorded", key=DEFAULT_KEY, ppl=32):
    import os
    ppl = int(ppl)
    (f, nf), (w, _) = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    doc = json.load(open(out)) if os.path.exists(out) else {}
    if "workloads" not in doc:
        doc = {"workloads": ({DEFAULT_KEY: {k: v for k, v in doc.items() if k != "_how"}} if doc else {})}
    doc["_how"] = ("per workload key (python bench.py <flags> --traffic-key): rocprofv3 --pmc FETCH_SIZE (and, in a separate run, --pmc WRITE_SIZE) "
                   "--output-format csv -- python3 bench.py <flags> --steps 2 --warmup 1 --streams 1 --pairs <pairs per launch> --no-cpu-baseline "
                   "--sustain-s 0 ; per-launch averages over all launches of the run (a launch of the batched path covers the pairs of one "
                   "lr_register_batch call); FETCH_SIZE/WRITE_SIZE are KiB; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE "
                   "reports half of a wide coalesced read, MI355X_MICROARCH.md section HBM); tools/pmc_traffic.sh + tools/pmc_to_json.py")
    res = {"_meta": {"commit": commit, "pairs_per_launch": ppl,
                     "note": "source tree the counters were collected on (passed to the profiling script from `git rev-parse`: the GPU box has no .git)"}}
    for k in sorted(set(f) | set(w)):
        if not k.strip():
            continue
        fk, wk = f.get(k, 0.0), w.get(k, 0.0)
        res[k] = {"FETCH_SIZE_KiB": round(fk, 1), "WRITE_SIZE_KiB": round(wk, 1), "hbm_bytes_per_launch": int((2 * fk + wk) * 1024),
                  "launches": nf.get(k, 0)}
    # whole pair: every launch of the library's own kernels (not the input generation, torch / rocprim / rocBLAS / runtime fills),
    # over the pairs the profiled run registered: one batched call of 32 pairs per prep launch (warm-up, timed steps and the
    # repetitions bench.py adds for the roofline events)
    own = [k for k in res if not k.startswith("_") and not k.startswith(("at::", "rocprim::", "Cijk_", "__amd_rocclr", "void at::")) and k.strip()]
    pairs = res["nn16_prep_kernel"]["launches"] * ppl         # one prep launch per batched call of `ppl` pairs
    total = sum(res[k]["hbm_bytes_per_launch"] * res[k]["launches"] for k in own)
    res["_pair"] = {"hbm_bytes_per_pair": int(total / pairs), "pairs_in_run": pairs,
                    "kernels": {k: int(res[k]["hbm_bytes_per_launch"] * res[k]["launches"] / pairs) for k in sorted(own, key=lambda k: -res[k]["hbm_bytes_per_launch"] * res[k]["launches"])[:12]}}
    doc["workloads"][key] = res
    json.dump(doc, open(out, "w"), indent=1)
    print("workload:", key)
    for k, v in res.items():
        if not k.startswith("_"):
            print(f"{k:36s} {v['hbm_bytes_per_launch'] / 1e6:9.2f} MB")
    print("per pair:", res["_pair"]["hbm_bytes_per_pair"] / 1e6, "MB")


if __name__ == "__main__":
    main(*sys.argv[1:7])
