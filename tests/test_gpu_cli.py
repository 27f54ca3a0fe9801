"""The CLI surface end to end on the GPU: Experiments/test.py (single process and the test_parallel protocol), the demo,
and shard equivalence (the merged table does not depend on the world size)."""
import importlib
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def cli(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    sys.path.insert(0, os.path.join(ROOT, "Experiments"))
    mod = importlib.import_module("test")
    yield mod
    sys.path.remove(os.path.join(ROOT, "Experiments"))


COMMON = ["--dataset", "synthetic", "--num_pairs", "6", "--synthetic_n", "4000", "--algo", "RANSAC", "--iters", "2000"]


def _outputs(tmp_path):
    d = sorted((tmp_path / "outputs").iterdir())[-1]
    from lidarregistration_amd import io_lists
    ids, T = io_lists.read_coarse_motions(str(d / "coarse_motions.txt"))
    return np.load(d / "raw_stats.npy"), ids, T, (d / "log.txt").read_text()


@pytest.mark.parametrize("mode_args", [["--mode", "MMN"], ["--mode", "GPF", "--GPF_factor", "0.5"], ["--mode", "no_filter", "--codebase", "open3D"],
                                       ["--mode", "MNN", "--fast_rejection", "SPRT", "--GC_LO", "False"]])
def test_single_process_run(cli, tmp_path, oracle, mode_args):
    stats = cli.main(COMMON + mode_args)
    raw, ids, T, log = _outputs(tmp_path)
    assert raw.shape == (6, 22) and np.array_equal(raw, stats, equal_nan=True)
    assert (raw[:, 0] == 1).all() and (raw[:, 1] < 1.0).all() and (raw[:, 2] < 30).all()
    assert (raw[:, 12] == 1).all() and (raw[:, 13] < 1.0).all() and (raw[:, 14] < 30).all() and (raw[:, 11] > 0).all()    # ICP columns
    assert ids[:, 1].tolist() == list(range(6)) and "recall: 100.00%" in log and "mode = " in log
    # pair 2 of the run against the oracle pipeline on the same synthetic pair (the CLI's source draws its pairs on the device)
    import torch
    from lidarregistration_amd import synth
    from tests.conftest import Args
    pd = synth.make_pair_dev(N=4000, seed=51 + 2, device=torch.device("cuda", torch.cuda.current_device()))
    p = {k: pd[k].cpu().numpy() for k in ("xyz0", "xyz1", "feats0", "feats1")}
    mode = mode_args[1]
    from tests.conftest import gc_oracle_kwargs
    a = Args(GPF_factor=0.5, prosac=True)          # the CLI's defaults: --codebase GC --prosac True --GC_LO True
    if "SPRT" in mode_args:
        a.fast_rejection = "SPRT"; a.GC_LO = False
    kw = dict(sample_size=4, confidence=0.9995, refit_on_orig=1, scoring=0) if "open3D" in mode_args else gc_oracle_kwargs(a)
    e = oracle.register_pair(p["xyz0"], p["xyz1"], p["feats0"], p["feats1"], mode=mode, iters=2000, seed=51, args=a, **kw)
    assert np.radians(oracle.rotation_error_deg(T[2], e["T"])) <= 1e-4 and oracle.translation_error_cm(T[2], e["T"]) / 100 <= 1e-3
    assert raw[2, 17] == len(e["idx0"])


@pytest.mark.parametrize("extra", [["--mode", "MNN"], ["--mode", "GPF", "--codebase", "open3D"]])
def test_batched_engine_equals_the_pair_by_pair_path(cli, tmp_path, extra):
    """The CLI registers its list through lr_register_batch (windows of batched calls); --serial True is the reference harness' call
    pattern (one lr_register_pair per row).  Same rows, same statistics: every result column and every transform identical."""
    common = ["--dataset", "synthetic", "--num_pairs", "11", "--synthetic_n", "5000", "--algo", "RANSAC", "--iters", "3000"]
    a = cli.main(common + extra + ["--batch", "4", "--in_flight", "2"])
    _, ids_a, T_a, log_a = _outputs(tmp_path)
    cols = (sorted((tmp_path / "outputs").iterdir())[-1] / "raw_stats.columns.txt").read_text()
    assert "9: model_time / reg time (s): ATTRIBUTED, not measured" in cols and "--serial True" in cols          # the batched engine's time columns are attributed shares of a window, and the file says so
    b = cli.main(common + extra + ["--serial", "True"])
    _, ids_b, T_b, _ = _outputs(tmp_path)
    assert "9: model_time / reg time (s): measured per pair" in (sorted((tmp_path / "outputs").iterdir())[-1] / "raw_stats.columns.txt").read_text()
    for c in (0, 1, 2, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21):
        assert np.array_equal(a[:, c], b[:, c]), c
    assert np.array_equal(T_a, T_b) and np.array_equal(ids_a, ids_b)
    assert (a[:, 9] > 0).all() and (a[:, 11] > 0).all() and (a[:, 9] < 0.05).all()
    assert "registration region" in log_a and "pairs/s" in log_a


def test_parallel_protocol_is_shard_invariant(cli, tmp_path):
    ref = cli.main(COMMON + ["--mode", "MNN"])
    base = str(tmp_path / "pp")
    for r in range(2):
        cli.main(["test_parallel", "20260101_00_00_00", base, "2", str(r)] + COMMON + ["--mode", "MNN"])
    merged = cli.main(["test_parallel", "20260101_00_00_00", base, "2", "analysis"] + COMMON + ["--mode", "MNN"])
    for c in (0, 1, 2, 15, 16, 17, 18, 19, 20, 21):
        assert np.array_equal(ref[:, c], merged[:, c]), c       # identical results whatever the world size
    # 3 ranks over 6 pairs with --max_samples 5: wrap-around padding is dropped by the analysis pass
    base = str(tmp_path / "qq")
    for r in range(3):
        cli.main(["test_parallel", "20260101_00_00_01", base, "3", str(r)] + COMMON + ["--mode", "MNN", "--max_samples", "5"])
    merged = cli.main(["test_parallel", "20260101_00_00_01", base, "3", "analysis"] + COMMON + ["--mode", "MNN", "--max_samples", "5"])
    assert merged.shape == (5, 22) and np.array_equal(merged[:, 1], ref[:5, 1])


def test_demo_registration(cli, capsys):
    demo = importlib.import_module("demo_registration")
    T = demo.main(["--algo", "RANSAC", "--mode", "MMN", "--iters", "1000"])
    out = capsys.readouterr().out
    assert T.shape == (4, 4) and "RE = " in out and "filtered pairs" in out


@pytest.mark.parametrize("dataset,extra", [("A", ["--mode", "GPF", "--iters", "50000"]),
                                           ("B", ["--mode", "MNN", "--iters", "1000000", "--GC_conf", "0.9995"])])
def test_list_driven_surrogate_configs(cli, tmp_path, monkeypatch, dataset, extra):
    """BASELINE configs 3 and 4 with the list-driven synthetic surrogate: ground-truth motion and overlap of every pair come from
    rows of the reference's balanced lists (64-row excerpts under tests/golden/), clouds and descriptors are synthetic."""
    monkeypatch.setenv("LIDARREG_BALANCED_SETS", os.path.join(ROOT, "tests", "golden", "balanced_sets_excerpt"))
    stats = cli.main(["--dataset", dataset, "--algo", "RANSAC", "--synthetic_n", "6000", "--max_samples", "12"] + extra)
    raw, ids, T, log = _outputs(tmp_path)
    assert raw.shape == (12, 22)
    from lidarregistration_amd import io_lists
    lst = io_lists.read_pair_list(os.path.join(ROOT, "tests", "golden", "balanced_sets_excerpt", io_lists.DATASET_NAMES[dataset], "test.txt"))
    # session / source / target ids of the list rows are carried through to the output files
    assert sorted(map(tuple, ids.tolist())) == sorted(zip(lst["session"][:12].tolist(), lst["src"][:12].tolist(), lst["tgt"][:12].tolist()))
    assert raw[:, 0].mean() >= 0.9 and np.nanmean(raw[:, 12]) >= 0.9            # recall of RANSAC and of RANSAC+ICP on the surrogate


def test_batched_harness_on_ragged_clouds_equals_the_serial_one():
    """harness.eval_pairs on pairs of DIFFERENT cloud sizes inside one batched call (real data: every scan has its own point count),
    with a pair that cannot be registered among them: every result column equal to eval_pairs_serial's."""
    import torch
    from lidarregistration_amd import harness, synth
    from tests.conftest import Args

    class Ragged:
        sizes = [(3000, 2600), (1500, 4000), (4096, 4096), (900, 700), (2500, 2500), (3333, 1024), (64, 80)]

        def __len__(self):
            return len(self.sizes)

        def ids(self, k):
            return 7, k, k + 1

        def get_dev(self, k, device):
            n0, n1 = self.sizes[k]
            p = synth.make_pair(N=n0, N1=n1, rho=0.5, s=0.8, seed=400 + k)
            if k == 3:                       # unrelated clouds: nothing to find
                p["xyz1"] = (np.random.default_rng(1).uniform(-80, 80, p["xyz1"].shape)).astype(np.float32)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)
            return dict(xyz0=t(p["xyz0"]), xyz1=t(p["xyz1"]), feats0=t(p["feats0"]), feats1=t(p["feats1"]), T_gt=p["T_gt"])

    src = Ragged()
    idx = list(range(len(src)))
    for a in (Args(mode="MNN", codebase="GC", iters=4000, prosac=True, icp=True), Args(mode="GPF", codebase="open3D", iters=4000, ransac_n=3, icp=True)):
        sb, Tb = harness.eval_pairs(src, idx, a, batch=3, in_flight=2)
        ss, Ts = harness.eval_pairs_serial(src, idx, a, in_flight=2)
        for c in (0, 1, 2, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21):
            assert np.array_equal(sb[:, c], ss[:, c], equal_nan=True), (a.mode, c, sb[:, c], ss[:, c])
        assert np.array_equal(Tb, Ts)
        assert sb[:, 15].tolist() == [s[0] for s in src.sizes] and (sb[[0, 1, 2, 4, 5], 0] == 1).all()


def test_file_backed_source_is_read_ahead_by_worker_threads(tmp_path):
    """A feature cache on disk (io_lists.save_cloud) through harness.CacheSource: eval_pairs reads the next window's files on worker
    threads while the current window registers -- same rows as with the reads on the caller's thread, in list order, over several windows."""
    from lidarregistration_amd import harness, io_lists, synth
    from tests.conftest import Args
    n_pairs = 9
    lst = dict(session=np.full(n_pairs, 3), src=np.arange(n_pairs) * 2, tgt=np.arange(n_pairs) * 2 + 1, T_gt=[None] * n_pairs, overlap=None)
    for k in range(n_pairs):
        p = synth.make_pair(N=1500 + 100 * k, N1=1400 + 50 * k, rho=0.5, s=0.8, seed=900 + k)
        io_lists.save_cloud(str(tmp_path), 3, 2 * k, p["xyz0"], p["feats0"]); io_lists.save_cloud(str(tmp_path), 3, 2 * k + 1, p["xyz1"], p["feats1"])
        lst["T_gt"][k] = p["T_gt"]
    src = harness.CacheSource(lst, str(tmp_path))
    order = [4, 0, 8, 2, 6, 1, 7, 3, 5]
    a = Args(mode="MNN", codebase="GC", iters=4000, prosac=True, icp=False)
    s1, T1 = harness.eval_pairs(src, order, a, batch=2, in_flight=2, workers=4)      # windows of 4 rows: 4 + 4 + 1
    s0, T0 = harness.eval_pairs(src, order, a, batch=2, in_flight=2, workers=0)
    for c in (0, 1, 2, 15, 16, 17, 18, 19, 20, 21):
        assert np.array_equal(s1[:, c], s0[:, c], equal_nan=True), c
    assert np.array_equal(T1, T0) and s1[:, 20].tolist() == [2 * k for k in order] and (s1[:, 0] == 1).all()


# ------------------------------------------------------------------ the launcher: ./test_parallel.sh = python -m test launch
def _launch(tmp_path, gpus, flags, timeout=1500):
    env = dict(os.environ, LIDARREG_GPUS=gpus)
    return subprocess.run(["bash", os.path.join(ROOT, "Experiments", "test_parallel.sh")] + flags, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=timeout)


def test_launcher_two_ranks_on_this_gpu_equals_one_process(cli, tmp_path):
    ref = cli.main(COMMON + ["--mode", "MNN"])
    for d in (tmp_path / "outputs").iterdir():
        shutil.rmtree(d)
    r = _launch(tmp_path, "0 0", COMMON + ["--mode", "MNN"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    raw, ids, T, log = _outputs(tmp_path)
    for c in (0, 1, 2, 15, 16, 17, 18, 19, 20, 21):
        assert np.array_equal(ref[:, c], raw[:, c]), c
    assert "process 0:" in log and "process 1:" in log


def test_launcher_default_device_list_on_this_box(cli, tmp_path):
    """./test_parallel.sh with LIDARREG_GPUS UNSET -- what the first run on an 8-GPU node does: the launcher finds this box's GPUs from the
    inherited *_VISIBLE_DEVICES or from the KFD topology in sysfs (never through HIP in the parent), starts one rank per device and runs the
    analysis; one rank per visible device, same rows as the in-process run."""
    import torch
    ref = cli.main(COMMON + ["--mode", "MNN"])
    for d in (tmp_path / "outputs").iterdir():
        shutil.rmtree(d)
    env = {k: v for k, v in os.environ.items() if k != "LIDARREG_GPUS"}
    r = subprocess.run(["bash", os.path.join(ROOT, "Experiments", "test_parallel.sh")] + COMMON + ["--mode", "MNN"], cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    raw, ids, T, log = _outputs(tmp_path)
    for c in (0, 1, 2, 15, 16, 17, 18, 19, 20, 21):
        assert np.array_equal(ref[:, c], raw[:, c]), c
    world = torch.cuda.device_count()
    assert world >= 1 and all(f"process {k}:" in log for k in range(world)) and f"process {world}:" not in log


def test_launcher_eight_ranks_on_this_gpu_over_list_rows(cli, tmp_path):
    """World size 8 (the reference's README command for NuScenes-Boston, test_parallel.sh:18-24) -- on ONE GPU here, so this says nothing
    about scaling: it exercises the 8-way shard (DistributedSampler order), eight concurrent ranks and the merge.  64 list rows."""
    flags = ["--dataset", "B", "--algo", "RANSAC", "--mode", "MNN", "--iters", "1000000", "--GC_conf", "0.9995", "--synthetic_n", "6000", "--max_samples", "64", "--batch", "8", "--in_flight", "2"]
    ref = cli.main(flags)
    for d in (tmp_path / "outputs").iterdir():
        shutil.rmtree(d)
    r = _launch(tmp_path, "0 0 0 0 0 0 0 0", flags)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    raw, ids, T, log = _outputs(tmp_path)
    assert raw.shape == (64, 22)
    for c in (0, 1, 2, 15, 16, 17, 18, 19, 20, 21):
        assert np.array_equal(ref[:, c], raw[:, c]), c       # identical results whatever the world size, rows back in list order
    assert all(f"process {k}:" in log for k in range(8))


def test_launcher_stops_when_a_rank_dies(tmp_path):
    import time
    t0 = time.time()
    r = _launch(tmp_path, "0 0 0 0 0 0 0 7", COMMON + ["--mode", "MNN"], timeout=600)          # rank 7 sees no device
    assert r.returncode != 0 and time.time() - t0 < 300
    assert "no analysis" in r.stdout + r.stderr
    out = tmp_path / "outputs"
    assert not out.exists() or not any((d / "raw_stats.npy").exists() for d in out.iterdir())
