"""ctypes binding of liblidarreg.so (the C ABI in include/lidarreg.h).

There is no fallback: if the library is missing or a call fails, an exception is raised.  PyTorch
is only used by the callers for device memory and streams; no torch type crosses this boundary.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("LIDARREG_LIB") or os.path.join(CSRC, "liblidarreg.so")      # LIDARREG_LIB: development hook (alternate builds)

LR_MODE_NO_FILTER, LR_MODE_MNN, LR_MODE_GPF = 0, 1, 2

SYMBOLS = [
    "lr_version", "lr_last_error", "lr_workspace_create", "lr_workspace_destroy", "lr_workspace_bytes", "lr_workspace_poison",
    "lr_nn_top2", "lr_nn_to_mutual", "lr_feat_ratio", "lr_gpf", "lr_gpf_bb_first", "lr_ransac", "lr_refit", "lr_icp", "lr_kabsch",
    "lr_register_pair", "lr_workspace_lists", "lr_workspace_timing", "lr_workspace_timing_read",
    "lr_workspace_create_batch", "lr_register_batch", "lr_workspace_lists_at", "lr_inlier_mask", "lr_workspace_mask_at",
    "lr_voxel_dedup_scratch_bytes", "lr_voxel_dedup", "lr_workspace_option", "lr_workspace_stage_times", "lr_icp_batch", "lr_workspace_lists_batch",
    "lr_workspace_clock", "lr_debug_fake_current_device",
]

# lr_workspace_option ids (include/lidarreg.h).  DEFAULT_OPTIONS is applied to every Workspace this module creates (a hook for
# tuning experiments and for the test that no option changes a result; the library itself reads no environment variable).
OPTIONS = {"nn_blocks": 1, "nn_blocks_batch": 2, "nn_sample_stride": 3, "rev_strips": 4, "nn_second_auto": 5, "clock_probe": 6}
DEFAULT_OPTIONS = {}


class LidarRegError(RuntimeError):
    pass


class RansacParams(ctypes.Structure):
    """lr_ransac_params.  Positional arguments start at sample_size: struct_size (the first field of the C struct) is filled in."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("sample_size", ctypes.c_int32), ("use_elc", ctypes.c_int32), ("thr2", ctypes.c_float),
                ("iters", ctypes.c_int32), ("seed", ctypes.c_uint64), ("confidence", ctypes.c_float), ("batch", ctypes.c_int32),
                ("sampler", ctypes.c_int32), ("prosac_growth", ctypes.c_int32), ("scoring", ctypes.c_int32), ("local_opt", ctypes.c_int32),
                ("lo_rounds", ctypes.c_int32), ("lo_trials", ctypes.c_int32), ("lo_max_calls", ctypes.c_int32), ("min_iters", ctypes.c_int32)]

    def __init__(self, *args, **kw):
        kw.pop("struct_size", None)          # (always this build's size: an explicit keyword used to end in ctypes' opaque "duplicate values" TypeError)
        super().__init__(ctypes.sizeof(type(self)), *args, **kw)

    def effective_thr2(self):
        """The squared threshold the estimator really tests inliers against: scoring 2 (MSAC as GC-RANSAC runs it) uses the
        truncated threshold (3/2 thr)^2 -- what lr_inlier_mask has to be given to reproduce the estimator's own inlier set."""
        return float(self.thr2) * 2.25 if self.scoring == 2 else float(self.thr2)


class RansacResult(ctypes.Structure):
    _fields_ = [("best_h", ctypes.c_int64), ("best_count", ctypes.c_uint32), ("pad0", ctypes.c_uint32),
                ("best_ssq", ctypes.c_uint64), ("n_valid", ctypes.c_int64), ("n_ids", ctypes.c_int64)]


class IcpResult(ctypes.Structure):
    _fields_ = [("fitness", ctypes.c_double), ("inlier_rmse", ctypes.c_double), ("n_corr", ctypes.c_int32),
                ("iterations", ctypes.c_int32)]


class PairResult(ctypes.Structure):
    _fields_ = [("T", ctypes.c_double * 16), ("T_ransac", ctypes.c_double * 16), ("ransac", RansacResult),
                ("n_corr", ctypes.c_int32), ("n_refit", ctypes.c_int32), ("n_nn_fixed", ctypes.c_int32),
                ("status", ctypes.c_int32), ("reserved", ctypes.c_int32 * 8), ("T_icp", ctypes.c_double * 16), ("icp", IcpResult)]


class PairParams(ctypes.Structure):
    """lr_pair_params.  struct_size (here and in the nested lr_ransac_params) is filled in."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("mode", ctypes.c_int32), ("refit", ctypes.c_int32), ("ransac", RansacParams),
                ("gpf_grid_wid", ctypes.c_int32), ("icp", ctypes.c_int32), ("gpf_factor", ctypes.c_double),
                ("refit_thr2", ctypes.c_double)]

    def __init__(self, *args, **kw):
        kw.pop("struct_size", None)
        super().__init__(ctypes.sizeof(type(self)), *args, **kw)
        if self.ransac.struct_size == 0:
            self.ransac.struct_size = ctypes.sizeof(RansacParams)


assert ctypes.sizeof(PairResult) == 496 and ctypes.sizeof(RansacParams) == 72 and ctypes.sizeof(PairParams) == 112

_lib = None


def build(force=False):
    """hipcc --offload-arch=gfx950 -> csrc/liblidarreg.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    srcs.append(os.path.join(_HERE, "..", "include", "lidarreg.h"))
    stale = not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j4", "liblidarreg.so"])
    return LIB_PATH


def lib():
    """Load the HIP library; raises LidarRegError when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LidarRegError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # torch ships its own libamdhip64; it must be in the process before this library is, so that both use the same
        # HIP runtime (loaded the other way round, the library binds to a second runtime that sees no device)
        import torch  # noqa: F401
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise LidarRegError(f"cannot load {LIB_PATH}: {e}") from e
        L.lr_last_error.restype = ctypes.c_char_p
        L.lr_workspace_bytes.restype = ctypes.c_size_t
        L.lr_workspace_bytes.argtypes = [ctypes.c_void_p]
        L.lr_workspace_poison.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.lr_workspace_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.lr_workspace_destroy.argtypes = [ctypes.c_void_p]
        vp, ci = ctypes.c_void_p, ctypes.c_int
        L.lr_nn_top2.argtypes = [vp, vp, ci, vp, ci, ci, vp, vp, vp, vp, vp]
        L.lr_nn_to_mutual.argtypes = [vp, vp, ci, vp, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]
        L.lr_feat_ratio.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, vp]
        L.lr_gpf.argtypes = [vp, vp, ci, vp, ci, ci, vp, vp, vp, ci, ctypes.c_double, vp, vp, vp, vp, vp, vp]
        L.lr_gpf_bb_first.argtypes = [vp, vp, ci, vp, ci, ci, vp, vp, vp, ci, ctypes.c_double, vp, vp, vp, vp, vp, vp, vp]
        L.lr_ransac.argtypes = [vp, vp, vp, ci, vp, ctypes.POINTER(RansacParams), vp, vp, vp]
        L.lr_refit.argtypes = [vp, vp, ci, vp, vp, vp, ctypes.c_double, vp, vp, vp]
        L.lr_icp.argtypes = [vp, vp, ci, vp, ci, vp, ctypes.c_double, ci, ctypes.c_double, ctypes.c_double, vp, vp, vp]
        L.lr_kabsch.argtypes = [vp, vp, vp, ci, vp, vp]
        L.lr_icp_batch.argtypes = [vp, ctypes.c_double, ci, ctypes.c_double, ctypes.c_double, vp, vp]
        L.lr_register_pair.argtypes = [vp, vp, vp, vp, vp, ci, ci, ci, ctypes.POINTER(PairParams), vp, vp]
        L.lr_workspace_lists.argtypes = [vp, ci, vp, vp, vp, vp, vp]
        L.lr_workspace_lists_at.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp]
        L.lr_workspace_lists_batch.argtypes = [vp, ci, ci, vp, vp, vp, vp, vp]
        L.lr_inlier_mask.argtypes = [vp, vp, vp, ci, vp, ctypes.c_float, vp, vp, vp]
        L.lr_workspace_mask_at.argtypes = [vp, ci, vp, vp, ci, ctypes.c_float, vp, vp, vp]
        L.lr_voxel_dedup_scratch_bytes.restype = ctypes.c_size_t
        L.lr_voxel_dedup_scratch_bytes.argtypes = [ci]
        L.lr_voxel_dedup.argtypes = [vp, ci, vp, vp, vp, vp, ctypes.c_size_t, vp]
        L.lr_workspace_create_batch.argtypes = [ctypes.POINTER(ctypes.c_void_p), ci, ci, ci, ci, ci]
        pp, ip = ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int32)
        L.lr_register_batch.argtypes = [vp, ci, pp, pp, pp, pp, ip, ip, ci, ctypes.POINTER(PairParams), vp, vp]
        L.lr_workspace_option.argtypes = [vp, ci, ci]
        L.lr_workspace_stage_times.argtypes = [vp, ctypes.POINTER(ctypes.c_float * 8), ctypes.POINTER(ci)]
        L.lr_workspace_timing.argtypes = [vp, ci]
        if hasattr(L, "lr_workspace_clock"):      # (absent only from older builds loaded through the LIDARREG_LIB development hook)
            L.lr_workspace_clock.argtypes = [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong), ci]
        L.lr_workspace_timing_read.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ci)]
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise LidarRegError(f"liblidarreg error {rc}: {lib().lr_last_error().decode()}")


class Workspace:
    """Owns one lr_workspace: device scratch for one in-flight pair, or -- max_pairs > 1 -- for one in-flight batch of pairs
    registered by a single lr_register_batch call."""

    def __init__(self, max_n0, max_n1, dim=32, max_iters=50000, max_pairs=1):
        self._h = ctypes.c_void_p()
        self.max_n0, self.max_n1, self.dim, self.max_iters = int(max_n0), int(max_n1), int(dim), int(max_iters)
        self.max_pairs = int(max_pairs)
        check(lib().lr_workspace_create_batch(ctypes.byref(self._h), self.max_pairs, self.max_n0, self.max_n1, self.dim, self.max_iters))
        for name, value in DEFAULT_OPTIONS.items():
            self.set_option(name, value)

    def set_option(self, name, value):
        """Tuning option of this workspace (OPTIONS; none changes a result)."""
        check(lib().lr_workspace_option(self._h, OPTIONS[name], int(value)))

    def stage_times(self):
        """(sums in ms [whole call, forward NN, forward filter pass, reverse filter pass, RANSAC gen+score, reverse NN], timed calls)
        since lr_workspace_timing(ws, 1); the stream must be synchronised."""
        out = (ctypes.c_float * 8)(); n = ctypes.c_int()
        check(lib().lr_workspace_stage_times(self._h, out, ctypes.byref(n)))
        return list(out[:6]), n.value

    def timing(self, enable):
        check(lib().lr_workspace_timing(self._h, int(bool(enable))))

    def clock(self, reset=False):
        """(MHz, shader cycles, 100 MHz ticks) of the filter-pass blocks since the last reset (option clock_probe); the streams that used the
        workspace must be synchronised."""
        mhz = ctypes.c_double(); cyc = ctypes.c_ulonglong(); tk = ctypes.c_ulonglong()
        check(lib().lr_workspace_clock(self._h, ctypes.byref(mhz), ctypes.byref(cyc), ctypes.byref(tk), int(bool(reset))))
        return mhz.value, cyc.value, tk.value

    @property
    def handle(self):
        return self._h

    def fits(self, n0, n1, iters):
        return n0 <= self.max_n0 and n1 <= self.max_n1 and iters <= self.max_iters

    @property
    def nbytes(self):
        return lib().lr_workspace_bytes(self._h)

    def poison(self, byte, stream=None):
        """Test hook: fill the scratch arena with one byte value (results must not depend on it)."""
        check(lib().lr_workspace_poison(self._h, int(byte), stream))

    def close(self):
        if self._h:
            lib().lr_workspace_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
