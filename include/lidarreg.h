/*
 * lidarreg.h -- C ABI of liblidarreg.so (hand-written HIP for gfx950 / MI355X).
 *
 * Drop-in boundary for the registration hot path of AmnonDrory/LidarRegistration.  Each entry point
 * names the reference interface it replaces (paths relative to the reference tree).  Conventions:
 *
 *   - every data pointer is a caller-owned DEVICE pointer (e.g. torch tensor .data_ptr()), contiguous,
 *     row-major; the library allocates nothing behind the caller's back except inside an explicit
 *     lr_workspace;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all calls are asynchronous on
 *     it and do not synchronise with the host; the data-path entry points only launch kernels (and
 *     1-D memsets) on that stream, so a caller may capture them in a HIP graph and replay it on new
 *     data in the same buffers (timing off: lr_workspace_timing records events);
 *   - return value 0 = LR_OK, negative = error (lr_last_error() gives the text); nothing throws;
 *   - 4x4 transforms are row-major float64, column-vector convention, cloud 0 -> cloud 1
 *     (the reference's pygcransac binding returns the transpose, GC_RANSAC.py:55 -- not here);
 *   - a workspace may be used by one stream at a time; use one workspace per in-flight pair;
 *   - a workspace belongs to the device that was current when it was created (hipGetDevice); every entry point that takes one returns
 *     LR_EINVAL -- before launching anything -- when another device is current or the stream belongs to another device.  The library
 *     holds gfx950 code objects only: lr_workspace_create refuses any other architecture (gcnArchName) and sizes its launches from the
 *     device's compute-unit count (multiProcessorCount).
 */
#ifndef LIDARREG_H
#define LIDARREG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LR_API __attribute__((visibility("default")))

enum { LR_OK = 0, LR_EINVAL = -1, LR_ENOMEM = -2, LR_EHIP = -3, LR_ESIZE = -4 };

/* filter modes of FR(): Experiments/algorithms/FR.py:48-56 ("MNN" | "GPF" | "no_filter") */
enum { LR_MODE_NO_FILTER = 0, LR_MODE_MNN = 1, LR_MODE_GPF = 2 };

typedef struct lr_workspace lr_workspace;

/* RANSAC knobs.  Replaces the parameter dict of GC_RANSAC.py:12-37 and the keyword arguments of
 * FR.py:128-137, with explicit flags instead of the reference's sentinel overloading.            */
typedef struct lr_ransac_params {
    uint32_t struct_size;   /* = sizeof(lr_ransac_params) of the header the caller was built against; every entry point that
                               takes the struct checks it first and returns LR_EINVAL on a mismatch (lr_version 102)   */
    int32_t  sample_size;   /* 3 = GC-RANSAC minimal solver, 4 = FR.py:134 ransac_n                  */
    int32_t  use_elc;       /* pre-verification (--fast_rejection, GC_RANSAC.py:29-34): 0 none; 1 edge-length check of the
                               sample, similarity 0.9 (preemption_edge_length.h:82); 2 SPRT on the estimated model
                               (gcransac_python.cpp:534-568, min_inlier_ratio_for_sprt 0.1): Wald's sequential test over
                               the first 256 correspondences, design (eps, delta, A) updated between batches          */
    float    thr2;          /* squared inlier threshold; (2*voxel)^2 = 0.36 (FR.py:85,95)             */
    int32_t  iters;         /* hypothesis ids 0..iters-1 (--iters, FR.py:65-67)                       */
    uint64_t seed;          /* Philox4x32-10 key; sample of hypothesis h = philox(seed, h)            */
    float    confidence;    /* early exit (--GC_conf / FR.py:136): ids are evaluated in batches of `batch`; after a
                               batch ending at id e the run stops when e >= log(1-confidence) / log(1 - (inl/M)^sample_size)
                               for the best model so far.  >= 1 (or <= 0): every id is evaluated.             */
    int32_t  batch;         /* batch length of the early-exit test, constant; 0 -> 1024, 8192, 65536, ... (eightfold) */
    int32_t  sampler;       /* 0: uniform WITH replacement (Open3D's RANSAC, FR.py:128-137); 2: uniform, unique indices
                               (GC_RANSAC.py:19 'sampler': 0 -> GC-RANSAC's UniformSampler); 1: PROSAC (--prosac,
                               GC_RANSAC.py:24,39-43): the correspondences must come best quality first; hypothesis id h =
                               PROSAC draw h+1: sample_size-1 indices uniformly from the first n-1 correspondences plus the
                               n-th, n from the growth function of Chum & Matas 2005 (as in USAC / GC-RANSAC's
                               prosac_sampler.h); ids past prosac_growth fall back to uniform sampling over all
                               correspondences.  Samplers 1 and 2 reject a draw with a repeated index (the id is consumed
                               like a failed pre-check)                                                              */
    int32_t  prosac_growth; /* T_N of the growth function (0 -> 100000, GC-RANSAC's default)          */
    int32_t  scoring;       /* which model wins: 0 = more inliers, then lower squared-error sum (Open3D: fitness, then
                               inlier RMSE); 1 = MSAC, the truncated quadratic cost: larger sum over inliers of (thr2 - d^2),
                               evaluated as count * (uint32)(thr2 * 2^20) - best_ssq; 2 = MSAC as GC-RANSAC runs it
                               (MSACScoringFunction behind gcransac_python.cpp:507-512; --codebase GC): inlier test, cost, exit
                               rule, local optimisation, final least squares and inlier mask all use the TRUNCATED threshold
                               (3/2 thr)^2 = 2.25 thr2 (upstream-recalled) -- i.e. scoring 1 at 1.5 x the threshold */
    int32_t  local_opt;     /* 0: none -- the winning minimal-sample model is returned (Open3D); 1: GC-RANSAC's local
                               optimisation (--GC_LO True, GC_RANSAC.py:36-37; gcransac_python.cpp:418-423,508-515): every new
                               best model is re-estimated by an inner RANSAC over its inliers (<= 10 rounds of 20 least-squares
                               fits on 21 inliers each, scored over all correspondences; spatial coherence weight 0), at the
                               granularity of the early-exit batches, plus the final iterated least squares; 2: the final
                               iterated least squares over the inliers only (--GC_LO False)                           */
    /* Three settings of gcransac_python.cpp:513-517 (553-556, 579-582) whose meaning lives in the un-vendored library; 0 = default.
     * `max_local_optimization_number = 20` (50 without a pre-verification) admits two readings -- least-squares fits per round of
     * one optimisation (lo_trials) or optimisations per run (lo_max_calls); both are knobs, both default to that number.         */
    int32_t  lo_rounds;     /* rounds of one local optimisation (upstream max_graph_cut_number), default 10             */
    int32_t  lo_trials;     /* least-squares fits per round, 1..20, default 20                                          */
    int32_t  lo_max_calls;  /* local optimisations per run, default 20 (use_elc != 0) / 50 (use_elc == 0)               */
    int32_t  min_iters;     /* the exit rule is not consulted before this many ids (min_iteration_number), default 20 / 50;
                               only matters with batches shorter than that                                              */
} lr_ransac_params;

/* Written to device memory by lr_ransac / lr_register_pair. */
typedef struct lr_ransac_result {
    int64_t  best_h;        /* winning hypothesis id, -1 when none had an inlier                     */
    uint32_t best_count;    /* its inlier count over the M correspondences                            */
    uint32_t pad0;          /* diagnostic: see lr_pair_result.reserved[1] (0 unless a hand-off wait of the local optimisation timed out) */
    uint64_t best_ssq;      /* sum over its inliers of (uint32)(d^2 * 2^20)                           */
    int64_t  n_valid;       /* hypotheses that passed the pre-check and were scored                   */
    int64_t  n_ids;         /* hypothesis ids examined before the run stopped (== iters without early exit) */
} lr_ransac_result;

/* Written by lr_icp / lr_register_pair(icp != 0). */
typedef struct lr_icp_result {
    double   fitness;       /* correspondences within max_dist / source points, last evaluation       */
    double   inlier_rmse;   /* sqrt(mean squared distance) over those correspondences                  */
    int32_t  n_corr;
    int32_t  iterations;    /* transform updates applied                                               */
} lr_icp_result;

/* Per-pair result block of lr_register_pair (device memory, 496 bytes). */
typedef struct lr_pair_result {
    double   T[16];         /* final transform (after the LS refit when refit != 0)                   */
    double   T_ransac[16];  /* winning minimal-sample model before the refit                          */
    lr_ransac_result ransac;
    int32_t  n_corr;        /* correspondences after filtering (num_pairs_filtered, FR.py:60)         */
    int32_t  n_refit;       /* inliers used by the refit (FR.py:104-108)                              */
    int32_t  n_nn_fixed;    /* NN rows/cols that needed the exact sqrt tie-break path                 */
    int32_t  status;        /* 0 ok, 1 = no valid hypothesis (T = identity, GC_RANSAC.py:51-52)       */
    int32_t  reserved[8];   /* [0]: diagnostic, like n_nn_fixed -- (model, correspondence) evaluations of the scoring passes in ppm of
                               scanning every list in full (0: not recorded); may differ between runs (the pilot among models with
                               equal head counts depends on scheduling), no result does.  [1]: diagnostic -- waits of the local optimisation's
                               helper-block hand-off that hit their 0.2 s bound (low 16 bits: the master block recomputed a scoring job alone;
                               high 16 bits: a helper block left without a job); integer sums make the result the same either way, a non-zero
                               value means time was lost (also in lr_ransac_result.pad0).  [2]: diagnostic -- a single-pair call launched only the form of the
                               filter pass the previous calls' norms asked for and this pair's asked for the other (bit 0: forward, bit 1: reverse pass):
                               every row then went through the exact scan, correct but slow; two calls in a row must agree before a form is launched
                               alone.  [3..7]: 0                                                                                  */
    double   T_icp[16];     /* T refined by point-to-point ICP (test.py:183-189) when icp != 0, else = T */
    lr_icp_result icp;
} lr_pair_result;

typedef struct lr_pair_params {
    uint32_t struct_size;   /* = sizeof(lr_pair_params), checked like lr_ransac_params.struct_size (which must be set too) */
    int32_t  mode;          /* LR_MODE_*                                                             */
    int32_t  refit;         /* 0 none; 1: LS refit on the ORIGINAL NN pairs within thr (FR.py:99-111, codebase open3D);
                               2: on the FILTERED pairs RANSAC ran on (GC-RANSAC's final least squares over its inliers);
                               3: as 1 but weighted by the inverse feature distance of each pair (DGR register_FCGF,
                                  DGR/core/deep_global_registration.py:531-537)                                  */
    lr_ransac_params ransac;
    /* GPF (matching.py:100-205), only read when mode == LR_MODE_GPF */
    int32_t  gpf_grid_wid;  /* --GPF_grid_wid, default 10                                            */
    int32_t  icp;           /* 1: refine T by ICP (max distance 0.6 m, 30 updates, 1e-6 criteria; test.py:183-189) */
    double   gpf_factor;    /* --GPF_factor,   default 2.0 (a Python float in the reference)         */
    double   refit_thr2;    /* fp64 squared threshold of the refit's inlier test, (2*0.3)**2 (FR.py:105) */
} lr_pair_params;

/* ---- library ------------------------------------------------------------------------------- */
LR_API int         lr_version(void);    /* 100 * major + minor; 103: lr_workspace_clock, LR_OPT_CLOCK_PROBE, lr_debug_fake_current_device, device checks; 102: params structs start with struct_size (72 / 112 bytes), descriptors of 1..32 dimensions; 101: lr_ransac_params 64 bytes / lr_pair_params 96 bytes (round 3), lr_icp_batch */
LR_API const char *lr_last_error(void);

/* Scratch for clouds up to (max_n0, max_n1) points x dim (1 <= dim <= 32; matching.py:22-65 takes any width) and up to max_iters hypotheses. */
LR_API int    lr_workspace_create(lr_workspace **ws, int max_n0, int max_n1, int dim, int max_iters);
/* The same for up to max_pairs (<= 64) pairs registered by ONE call of lr_register_batch: max_pairs arenas of identical
 * layout in one allocation.  The single-pair entry points below work on such a workspace too (they use arena 0).        */
LR_API int    lr_workspace_create_batch(lr_workspace **ws, int max_pairs, int max_n0, int max_n1, int dim, int max_iters);
LR_API int    lr_workspace_destroy(lr_workspace *ws);
LR_API size_t lr_workspace_bytes(const lr_workspace *ws);
/* Test hook (no reference counterpart): fill the scratch arena with one byte value; results must not depend on it. */
LR_API int    lr_workspace_poison(lr_workspace *ws, int byte, void *stream);
/* Test hook (no reference counterpart): the entry points take `device` for the current device from now on (-1: ask HIP again), so that a
 * one-GPU box can exercise the wrong-device refusal above. */
LR_API int    lr_debug_fake_current_device(int device);
/* Tuning options of a workspace (no reference counterpart; NONE of them changes a result, tests/test_gpu_parity.py; the library
 * reads no environment variable).  value 0 restores the default.                                                          */
enum {
    LR_OPT_NN_BLOCKS        = 1,  /* a single-pair filter pass is cut into column strips so that it launches about this many blocks (512) */
    LR_OPT_NN_BLOCKS_BATCH  = 2,  /* the same for a batched call, over all its pairs (3072)                                  */
    LR_OPT_NN_SAMPLE_STRIDE = 3,  /* the filter pass samples every k-th column tile for its start thresholds (default: strip tiles / 32, at most 32) */
    LR_OPT_REV_STRIPS       = 4,  /* column strips offered to each row block of the reverse NN pass (default 48 / pairs, within 2..8) */
    LR_OPT_NN_SECOND_AUTO   = 5,  /* 1: lr_register_pair / _batch compute the second neighbour only when a stage of the call reads it */
    LR_OPT_CLOCK_PROBE      = 6   /* 1: the filter-pass blocks sum their shader cycles and 100 MHz ticks into the workspace (lr_workspace_clock) */
};
LR_API int    lr_workspace_option(lr_workspace *ws, int option, int value);
/* Measurement hook (no reference counterpart): the shader clock the filter-pass blocks ran at since the last reset, *mhz = 100 * cycles / ticks
 * (0 when nothing was recorded: LR_OPT_CLOCK_PROBE off).  The caller has synchronised the streams that used the workspace.  Any pointer may be NULL. */
LR_API int    lr_workspace_clock(lr_workspace *ws, double *mhz, unsigned long long *cycles, unsigned long long *ticks, int reset);

/* ---- a1/a2: find_nn / find_2nn  (Experiments/algorithms/matching.py:6-65) ------------------------
 * For every row of F0 [n0,dim] the nearest and second nearest row of F1 [n1,dim] under L2, first
 * minimal value wins.  idx2/s1/s2 may be NULL.  s = sqrt(max(d2,1e-30)) as matching.py:30.        */
LR_API int lr_nn_top2(lr_workspace *ws, const float *F0, int n0, const float *F1, int n1, int dim,
                      int32_t *idx1, int32_t *idx2, float *s1, float *s2, void *stream);

/* ---- a3-a5: nn_to_mutual / mark_best_buddies  (matching.py:67-87, 207-239) -----------------------
 * Runs the reverse NN (F1 -> F0) and intersects: is_bb[i] = (rev[idx1[i]] == i).  The surviving
 * pairs are written in ascending i (torch coalesce order) to out_idx0/out_idx1[/out_idx2]; their
 * number to *n_out (device int32).  is_bb, out_* and idx2 may be NULL.                             */
LR_API int lr_nn_to_mutual(lr_workspace *ws, const float *F0, int n0, const float *F1, int n1, int dim,
                           const int32_t *idx1, const int32_t *idx2,
                           uint8_t *is_bb, int32_t *out_idx0, int32_t *out_idx1, int32_t *out_idx2,
                           int32_t *n_out, void *stream);

/* ---- a6: calc_distance_ratio_in_feature_space  (matching.py:89-98) ------------------------------- */
LR_API int lr_feat_ratio(const float *F0, const float *F1, int dim, int m,
                         const int32_t *i0, const int32_t *i1, const int32_t *i2, float *out, void *stream);

/* ---- a7: Grid_Prioritized_Filter, BB_first=False  (matching.py:100-205) --------------------------
 * idx1/idx2 are the NN lists of all n0 rows; xyz0 [n0,3].  Kept pairs (ascending i) go to out_*,
 * their count to *n_out (device), their score (norm_feat_dist, matching.py:124,134) to out_score. */
LR_API int lr_gpf(lr_workspace *ws, const float *F0, int n0, const float *F1, int n1, int dim,
                  const int32_t *idx1, const int32_t *idx2, const float *xyz0,
                  int grid_wid, double factor,
                  int32_t *out_idx0, int32_t *out_idx1, int32_t *out_idx2, float *out_score,
                  int32_t *n_out, void *stream);

/* ---- a7, BB_first=True  (matching.py:109-113,126; the reference's TEASER wrapper, TEASER_plus_plus.py:109-110) ----
 * Mutual pairs first, then the grid filter over them with TOTAL_NUM = max_matches and no best-buddy shift.
 * *has_score (device int32) is 0 when the mutual set was already <= max_matches (the reference returns None). */
LR_API int lr_gpf_bb_first(lr_workspace *ws, const float *F0, int n0, const float *F1, int n1, int dim,
                           const int32_t *idx1, const int32_t *idx2, const float *xyz0,
                           int grid_wid, double max_matches,
                           int32_t *out_idx0, int32_t *out_idx1, int32_t *out_idx2, float *out_score,
                           int32_t *n_out, int32_t *has_score, void *stream);

/* ---- a10/a12: RANSAC over M correspondences src[i] <-> tgt[i]  ([M,3] float32 each) ---------------
 * Replaces pygcransac.findRigidTransform(x1y1z1, x2y2z2, ...) (GC_RANSAC.py:46-49; native side
 * gcransac_python.cpp:404-416) and o3d registration_ransac_based_on_correspondence (FR.py:128-137).
 * m_dev, if not NULL, is a device int32 holding the live M (<= m).  Writes T_out[16] and *res.    */
LR_API int lr_ransac(lr_workspace *ws, const float *src, const float *tgt, int m, const int32_t *m_dev,
                     const lr_ransac_params *p, double *T_out, lr_ransac_result *res, void *stream);

/* The inlier mask findRigidTransform returns next to the pose (gcransac_python.cpp:594-603): mask[c] = 1 when
 * |T src[c] - tgt[c]|^2 < thr2 in the fp32 arithmetic of the scoring kernel; *n_inliers (device, may be NULL) their number.
 * T is a DEVICE pointer (e.g. T_out of lr_ransac); with T = NULL the model of the last lr_ransac / lr_register_pair on ws.  */
LR_API int lr_inlier_mask(lr_workspace *ws, const float *src, const float *tgt, int m, const double *T, float thr2,
                          uint8_t *mask, int32_t *n_inliers, void *stream);
/* ... over the filtered pairs (corr_idx0[c], corr_idx1[c]) of pair `pair` of the last lr_register_pair / lr_register_batch,
 * with that pair's RANSAC model (lr_pair_result.T_ransac); mask has room for n0 entries, the first n_corr are written.      */
LR_API int lr_workspace_mask_at(lr_workspace *ws, int pair, const float *xyz0, const float *xyz1, int n0, float thr2,
                                uint8_t *mask, int32_t *n_inliers, void *stream);

/* ---- a11: LS refit on the original NN pairs within thr of T_in  (FR.py:99-111) -------------------- */
LR_API int lr_refit(lr_workspace *ws, const float *xyz0, int n0, const float *xyz1, const int32_t *idx1,
                    const double *T_in, double thr2, double *T_out, int32_t *n_inliers, void *stream);

/* ---- f1: point-to-point ICP refinement (Experiments/test.py:183-189; Open3D registration_icp) ---------------
 * src = xyz0 [n0,3], tgt = xyz1 [n1,3] float32, T_init[16] device float64.  Open3D defaults: max_iter 30,
 * rel_fitness = rel_rmse = 1e-6.                                                                   */
LR_API int lr_icp(lr_workspace *ws, const float *xyz0, int n0, const float *xyz1, int n1, const double *T_init,
                  double max_dist, int max_iter, double rel_fitness, double rel_rmse,
                  double *T_out, lr_icp_result *res, void *stream);

/* The same refinement for EVERY pair of the last lr_register_batch call on this workspace, as one set of launches: the
 * harness runs ICP after the timed registration and times it on its own (Experiments/test.py:183-193, stats columns 11-14).
 * Starts from that call's final transforms (still in the arenas), over the clouds that call was given (they must still be
 * alive); fills T_icp / icp of out[k], k < npairs of that call (out = the result blocks of that call, or a copy).       */
LR_API int lr_icp_batch(lr_workspace *ws, double max_dist, int max_iter, double rel_fitness, double rel_rmse,
                        lr_pair_result *out, void *stream);

/* ---- a13: least-squares rigid fit of n point pairs  (models/common.py:7-45) -------------------------
 * P, Q [n,3] float64, optional weights w [n]; T_out[16].                                           */
LR_API int lr_kabsch(const double *P, const double *Q, const double *w, int n, double *T_out, void *stream);

/* ---- a9: FR() end to end on device  (Experiments/algorithms/FR.py:16-119) -------------------------
 * NN -> filter (mode) -> RANSAC -> optional refit, one call, no host synchronisation.             */
LR_API int lr_register_pair(lr_workspace *ws, const float *xyz0, const float *xyz1,
                            const float *F0, const float *F1, int n0, int n1, int dim,
                            const lr_pair_params *p, lr_pair_result *out, void *stream);

/* ---- a9 for many pairs: what Experiments/test.py:165-167 does pair after pair (one FR() per list row), as ONE sequence of
 * kernel launches over `npairs` pairs: every kernel of the path is launched once with the pair as a grid dimension, so the
 * GPU is filled by the batch instead of by many concurrent streams.  xyz0/xyz1/F0/F1/n0/n1 are HOST arrays of length npairs
 * (device pointers / cloud sizes, which may differ from pair to pair); out is a DEVICE array of npairs result blocks.  The
 * result of pair k is bit-identical to lr_register_pair on that pair.
 * ONE CALL IN FLIGHT PER WORKSPACE: the call's descriptor table and scratch arenas belong to the workspace, so a second call on
 * the same workspace -- from any stream -- may only be enqueued behind the first on the SAME stream; concurrent batches need one
 * workspace each (bench.py: one per stream).  The *_at accessors read pair < npairs of the LAST call only.                  */
LR_API int lr_register_batch(lr_workspace *ws, int npairs, const float *const *xyz0, const float *const *xyz1,
                             const float *const *F0, const float *const *F1, const int32_t *n0, const int32_t *n1, int dim,
                             const lr_pair_params *p, lr_pair_result *out, void *stream);

/* Copies the correspondence lists of the last lr_register_pair on this workspace into caller-owned
 * device buffers (any may be NULL): the NN lists over all n0 rows (FR.py's corres_idx1_orig / idx1_2nd_orig)
 * and the filtered lists, of which the first out->n_corr entries are live (buffers sized n0).     */
LR_API int lr_workspace_lists(lr_workspace *ws, int n0, int32_t *nn_idx1, int32_t *nn_idx2,
                              int32_t *corr_idx0, int32_t *corr_idx1, void *stream);
/* ... and of pair `pair` of the last lr_register_batch */
LR_API int lr_workspace_lists_at(lr_workspace *ws, int pair, int n0, int32_t *nn_idx1, int32_t *nn_idx2,
                                 int32_t *corr_idx0, int32_t *corr_idx1, void *stream);
/* ... and of its first `npairs` pairs at once: buffers [npairs][width] int32, row k = pair k (what the harness reads for the
 * statistics of Experiments/test.py:200-208 after a batched call: one strided copy per list)                             */
LR_API int lr_workspace_lists_batch(lr_workspace *ws, int npairs, int width, int32_t *nn_idx1, int32_t *nn_idx2,
                                    int32_t *corr_idx0, int32_t *corr_idx1, void *stream);

/* ---- f2: voxel de-duplication of a raw cloud -- ME.utils.sparse_quantize(xyz / voxel_size, return_index=True) as the
 * reference's loaders call it (Experiments/dataloader/generic_balanced_loader.py:62-63; voxel_size 0.3).
 * coords [n,3] float64 device = xyz / voxel_size (the division is the caller's, as in the reference); one point per occupied
 * integer cell floor(coords) is kept -- the first in input order -- and the kept point indices are written to sel in ascending
 * order, their number to *n_sel (device).  cells (optional, [n,3] int32) receives the integer cell of every kept point.
 * Points with a non-finite coordinate or |cell| >= 2^20 are dropped.  scratch: lr_voxel_dedup_scratch_bytes(n) device bytes.  */
LR_API size_t lr_voxel_dedup_scratch_bytes(int n);
LR_API int    lr_voxel_dedup(const double *coords, int n, int32_t *sel, int32_t *n_sel, int32_t *cells, void *scratch,
                             size_t scratch_bytes, void *stream);

/* ---- measurement hook for bench.py: duration of the last NN distance kernel(s) on this workspace,
 * from HIP events recorded on the launch stream.  Enable, run, synchronise, then read.            */
LR_API int lr_workspace_timing(lr_workspace *ws, int enable);
LR_API int lr_workspace_timing_read(lr_workspace *ws, float *nn_ms, float *ransac_ms, int *n_samples);
/* Stage times of the timed lr_register_pair / _batch calls since lr_workspace_timing(ws, 1), sums in ms over *n_samples calls
 * (read after synchronising the stream): out[0] whole call; out[1] forward NN = find_nn of matching.py:22-65 incl. the second
 * neighbour (norms + f16 copies, filter pass, exact verification); out[2] / out[3] the forward / reverse filter-pass launch;
 * out[4] hypothesis generation + scoring of the first RANSAC batch; out[5] the reverse NN (ordering, filter pass, exact
 * verification; 0 with --mode no_filter); out[6..7] reserved (0).  The registration time FR.py:117
 * bills -- filter + RANSAC + refit + the second neighbour's surcharge, matching.py:12-18 -- is out[0] - out[1] + that surcharge.  */
LR_API int lr_workspace_stage_times(lr_workspace *ws, float out[8], int *n_samples);

#ifdef __cplusplus
}
#endif
#endif /* LIDARREG_H */
