// Internal declarations shared by the translation units of liblidarreg.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/lidarreg.h"

void lr_set_error(const char *fmt, ...);

#define LR_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            lr_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return LR_EHIP;                                                               \
        }                                                                                 \
    } while (0)

#define LR_LAUNCH_CHECK() LR_HIP(hipGetLastError())

#define LR_REQUIRE(cond, code, msg)                 \
    do {                                            \
        if (!(cond)) { lr_set_error("%s", msg); return code; } \
    } while (0)

// Candidate store of the f16 filter: every wave of a pass-B block (64 query rows x one column strip) owns a private segment
// of { column | lane group, row mask | filter value } entries (lr_nn16.hip, LR_PB_*); the segments of one wave's rows over all strips hold at most
// LR_NN16_SEG entries (8 bytes each).  Columns must be < 2^22.
#define LR_NN16_SEG 4096
#define LR_NN16_SEG_INTS(n) ((size_t)((n) / 256 + 1) * 4 * LR_NN16_SEG * 2)      // int32 words of the store for n query rows
#define LR_NN16_CNT_INTS(n) ((size_t)((n) / 256 + 1) * 4 * 65)                   // segment counters (up to 64 strips) + strips used
static inline int lr_cdiv(int a, int b) { return (a + b - 1) / b; }
// entries per (row block, wave, strip) segment when a pass runs with `strips` column strips
__host__ __device__ static inline int lr_seg_cap(int strips) { const int c = (LR_NN16_SEG / strips) & ~63; return c < 64 ? 64 : c; }

#define LR_TRY_HIP(x) do { int rc__ = (x); if (rc__ != LR_OK) return rc__; } while (0)

// Column strips of the NN distance kernel: each wave owns 32 query rows x one strip.
#define LR_NN_MAX_STRIPS 8
#define LR_FEAT_DIM 32

// Waves of the scoring kernel (four per block); this many are launched and map themselves onto
// (hypothesis group, correspondence chunk) from the live counts on device.
#ifndef LR_SCORE_BLOCKS
#define LR_SCORE_BLOCKS 2048      // 512 blocks per pair (8192: one work item per wave, +4 % time: measured round 3)
#endif
#define LR_GPF_MAX_CELLS 4096
#define LR_PR_BUCKETS 8192        // PROSAC ordering: linear buckets over the quality range (lr_filter.hip); its offsets / fill / range reuse gpf_cells
#define LR_SC_INFO_BYTES 2048
#define LR_LO_CTL_BYTES 20480      // lr_lo_ctl: control block of the local optimisation's helper blocks
#define LR_NEV 10

// ---- pair-batched launches -------------------------------------------------------------------------------------------
// A workspace holds `max_pairs` arenas of identical layout, `stride` bytes apart; every scratch pointer below refers to
// arena 0.  A batched call launches each kernel ONCE for all pairs: the pair is the grid's z index (or decoded from a
// 1-D XCD-aware grid), scratch pointers are moved by pair * stride, and the pair's inputs / sizes come from a descriptor
// table in device memory.  The single-pair operators pass z = { 0, nullptr }: pointers and sizes are used as given.
struct lr_pair_desc {
    const float *xyz0, *xyz1, *F0, *F1;
    int32_t n0, n1;
};
struct lr_zargs {
    size_t stride;               // bytes between the arenas of consecutive pairs (0: single pair)
    const lr_pair_desc *descs;   // [pairs] device table, or nullptr: inputs and sizes are the kernel's own arguments
};
#define LR_MAX_BATCH 64          // pairs per batched call (the descriptor table travels as a kernel argument: 64 * 40 B)
struct lr_desc_table { lr_pair_desc d[LR_MAX_BATCH]; };

#ifdef __HIPCC__
// scratch pointer of pair `pair` (null stays null: optional outputs).  A macro: the kernels' pointer parameters are
// __restrict__-qualified, which a `T *&` template parameter does not bind to.  The offset is applied as pointer arithmetic on
// the parameter itself (never through an integer): the compiler then still knows that the address is global memory, uniform
// and not clobbered by the kernel's own stores -- which is what lets wave-uniform streams stay on scalar loads.
// Branch-free (a select on the offset): with an `if` per pointer the compiler fetched every kernel argument in its own basic block --
// up to twenty dependent s_load / s_waitcnt pairs at the head of every block of every kernel.
#define lr_z(p, z, pair) \
    do { const size_t lr_zo_ = (size_t)(pair) * (z).stride; p = (decltype(p))((const char *)(p) + (lr_zo_ & ((size_t)0 - (size_t)((p) != nullptr)))); } while (0)
// 1-D grid of 8 * ceil(total / 8) blocks -> logical block id such that the blocks an XCD receives (hardware ids congruent
// mod 8) form one contiguous range of logical ids (cdna_hip_programming.md T1).  false: padding block.
__device__ __forceinline__ bool lr_xcd_block(int total, int &logical)
{
    const int chunk = (int)gridDim.x >> 3;
    logical = ((int)blockIdx.x & 7) * chunk + ((int)blockIdx.x >> 3);
    return logical < total;
}
#endif

struct lr_workspace {
    int max_n0, max_n1, max_n, dim, max_iters;
    int max_pairs;               // arenas in this workspace
    int device, n_cus;           // the device the workspace lives on (checked against the current device at every entry point), its compute units
    int last_npairs;             // pairs of the last lr_register_pair / _batch call (0: none yet): what the *_at accessors may read
    int last_batch;              // 1: that call was lr_register_batch (descs[] still describes its pairs: lr_icp_batch)
    int last_mx0, last_mx1;      // largest cloud sizes of that call (they size the grids of a follow-up stage)
    const double *last_T_final;  // arena-0 pointer of that call's final transform (T_tmp or T_tmp + 16)
    hipStream_t last_stream;     // the stream that call ran on (lr_icp_batch must follow on the same one)
    size_t stride;               // bytes per arena
    size_t bytes;
    char *base;                  // one hipMalloc
    lr_pair_desc *descs;         // [LR_MAX_BATCH] descriptor table of the batched call in flight (arena 0 only)
    // call context of the entry point in flight (one stream at a time per workspace): grid z extent and the kernels' z argument
    int zP;
    lr_zargs z;
    // --- NN (both directions share these) ---
    float *nrm0, *nrm1;          // row norms
    _Float16 *H0, *H1;           // [n,32] f16 copies for the matrix-core filter passes
    float *P0, *P1;              // [n,32] zero-padded fp32 copies of descriptors narrower than 32 (nullptr at dim 32)
    float *tau;                  // [max_n] per-row candidate threshold
    uint32_t *yshare;            // [max_n] the rows' tightest threshold so far over all column strips of the filter pass in flight (order-preserving
                                 // integer image of y, lowered with atomicMin): how the strips of one row block learn from each other
    float *yfin;                 // [LR_NN_MAX_STRIPS][max_n] the rows' final thresholds (y = tau/2) of the forward filter pass, per strip
    int32_t *cand_cnt, *cand;    // segment counters [row blocks][4 waves][strips] and the candidate store (LR_NN16_SEG_INTS)
    float *bmax0, *bmax1;        // per-32-row maxima of the norms (f16 filter error bound)
    float *bmin0, *bmin1;        // ... and minima
    float *nn_range;             // [4] largest / smallest squared norm of cloud 0, of cloud 1 (are the norms all alike?  then the walk's candidate test is a sign test)
    int form_prev[2];            // what lr_nn16_form_hint read at the previous single-pair call (a hint counts when two readings agree)
    int32_t *form_host, *form_dev;   // pinned, device-visible: [0] / [1] = the form the norms of cloud 0 / cloud 1 asked for in the last single-pair call (0 unknown, 1 sign, 2 plain)
    uint32_t *rev_seed;          // [max_n1] best forward distance pointing at each cloud-1 row (bit pattern)
    unsigned long long *rev_seed64;  // [max_n1] ... and who: (distance bits << 32) | smallest cloud-0 index at that distance
    int32_t *rev_rows;           // [max_n1] the cloud-1 rows that have one, by descending seed
    int32_t *rev_cols;           // [max_n0] cloud-0 points by ascending NN distance (columns of the reverse pass)
    int32_t *rev_pos;            // [max_n0] position of every cloud-0 point in rev_cols order (scratch between the rank and the copy kernel)
    float *rev_s1;               // [max_n0] column key of the reverse pass per cloud-0 point: its 2nd-NN distance (NN distance when no 2nd was asked for)
    float *rev_tmin;             // [max_n0/32+1] smallest key of each column tile
    int32_t *rev_hist;           // [2][4096] counting-sort offsets
    _Float16 *Hs; float *nrms;   // [max_n0] f16 rows and norms of cloud 0 in rev_cols order
    // tuning options (lr_workspace_option; none of them changes a result)
    int nn_blocks_target;        // LR_OPT_NN_BLOCKS: column strips are chosen so that a filter pass launches about this many blocks
    int nn_blocks_batch;         // LR_OPT_NN_BLOCKS_BATCH: the same for a batched call (all pairs together)
    int nn_sample_stride;        // LR_OPT_NN_SAMPLE_STRIDE: phase 1 of the filter pass samples every this-many-th column tile (0: by the strip length)
    int rev_strips;              // LR_OPT_REV_STRIPS: strips offered to every row block of the reverse pass (0: by the number of pairs)
    int nn_second_auto;          // LR_OPT_NN_SECOND_AUTO: the pair pipeline computes the 2nd neighbour only when a stage reads it
    int clock_probe;             // LR_OPT_CLOCK_PROBE: the filter-pass blocks sum their shader cycles / 100 MHz ticks into clk_dev
    unsigned long long *clk_dev; // [2] device words behind lr_workspace_clock (outside the arenas: one per workspace)
    int32_t *counters;           // small int block, see LR_CNT_*
    // --- per-pair lists ---
    int32_t *nn_idx1, *nn_idx2;  // forward NN over n0 rows
    float *nn_s1, *nn_s2;
    int32_t *rev_idx1;           // reverse NN over n1 rows
    uint8_t *is_bb;
    int32_t *blk_cnt;            // per-256-element survivor counts of the ordered compaction
    int32_t *corr_idx0, *corr_idx1, *corr_idx2;   // filtered lists [max_n0]
    float *corr_score;
    float *ratio;                // GPF: ratio / normalised score over n0
    int32_t *cell;               // GPF: cell id per pair
    int32_t *cell_sorted;        // GPF: pair ids bucketed by cell
    double *gpf_quota;           // GPF: per-cell quota [LR_GPF_MAX_CELLS]
    int32_t *gpf_cells;          // GPF: cell_count | cell_fill | cell_off, each [LR_GPF_MAX_CELLS + 8]; PROSAC: offsets | fill [LR_PR_BUCKETS + 8] | range
    uint8_t *gpf_keep;           // GPF: keep mask [max_n0]
    float *gpf_f;                // GPF: min/max scratch
    int32_t *prosac_G;           // [max_n0+2] PROSAC growth function G[n], n = sample_size..M
    int32_t *prosac_rank;        // [max_n0] position of each filtered pair in quality order
    float *corr8;                // packed correspondences, 8 floats each, interleaved in pairs (lr_corr_at)
    // --- RANSAC ---
    float *models;               // [max_iters][12] fp32 R|t rows of hypotheses that passed the pre-check
    double *models64;            // [max_iters][12] the same models in fp64 (the winner is returned from here)
    int32_t *model_h;            // [max_iters] their hypothesis ids
    float *models2; double *models64_2; int32_t *model_h2;   // the same three for the models that survived the SPRT pre-verification
    uint32_t *score_cnt;         // [max_iters]
    unsigned long long *score_ssq; // [max_iters]
    double *refit_part;          // [blocks][16] moment partials
    int32_t *lo_list;            // [max_n0] inlier list of the model under local optimisation
    void *lo_ctl;                // lr_lo_ctl (LR_LO_CTL_BYTES): jobs the master block of a local optimisation publishes for its helper blocks
    // pilot-ordered scoring (lr_ransac.hip, "score"): head / order passes -> main pass
    void *sc_info;               // lr_score_info (LR_SC_INFO_BYTES)
    float *corr8s;               // the records behind the head, sorted by their residual under the pilot model (layout of corr8)
    int32_t *sc_perm;            // [max_iters] model slots by descending reach
    float *models_s;             // [12][max_iters] the fp32 models in that order
    int32_t *sc_glen;            // [max_iters/64+1] records every group of 64 sorted models has to scan
    uint8_t *sc_mb, *sc_cb;      // [max_iters] reach bucket per model, [max_n0] residual bucket per correspondence
    lr_ransac_result *res_tmp;
    double *T_tmp;               // [32]
    // --- ICP ---
    int32_t *icp_ints;           // hist | fill | start of the hashed target grid
    int32_t *icp_bucket;         // [max_n1] bucket of every target point
    float *icp_pts;              // [max_n1][4] target points in bucket order: x y z index-bits
    double *icp_state, *icp_part;
    // --- timing hook ---
    int timing;
    hipEvent_t ev[LR_NEV];       // [0,1] forward filter pass, [2,3] RANSAC gen+score, [4,5] reverse filter pass, [6] call start, [7] forward NN done, [8] call end, [9] reverse NN done
    float nn_ms_acc, ransac_ms_acc, call_ms_acc, fwd_ms_acc, fwd_filter_ms_acc, rev_filter_ms_acc, rev_ms_acc;
    int rev_done_recorded;
    int n_samples;
    int ev_pending;
    int rev_recorded;
};

enum {
    LR_CNT_FIX = 0,      // number of rows in fix_list
    LR_CNT_FIX_TOTAL,    // running total over the pair (stats)
    LR_CNT_NCORR,        // live M
    LR_CNT_NVALID,       // hypotheses appended to models[]
    LR_CNT_NBB,          // best buddies
    LR_CNT_NREV,         // rows of the reverse NN pass (cloud-1 points some query points at)
    LR_CNT_RLO,          // smallest / largest forward NN distance of the pair (float bit patterns)
    LR_CNT_RHI,
    LR_CNT_REFIT_TICKET, // blocks of the refit kernel that have finished (last-block-done; reset by the last block)
    LR_CNT_NVALID2,      // SPRT pre-verification: models that survived it (dense second list, scored in full)
    LR_CNT_FORM_MISS_F,  // single-pair calls launch only the form of the filter pass the last call's norms asked for: set when THIS call's norms
    LR_CNT_FORM_MISS_R,  //   ask for the other one (forward / reverse launch) -- the exact kernel then re-does every row by the full scan
    LR_CNT_COUNT = 16,
    LR_CNT_TOTAL = 64        // counters[16..63] hold lr_ransac_state
};

// corr8 layout: two correspondences share one 64-byte record { px_a px_b py_a py_b pz_a pz_b qx_a qx_b qy_a qy_b qz_a qz_b 0 0 0 0 },
// so the scoring loop reads a pair with one scalar load and feeds the halves to packed fp32 instructions as they lie.
// Float index of component k (0..5 = px py pz qx qy qz) of correspondence c:
__host__ __device__ inline size_t lr_corr_at(int c, int k) { return (size_t)(c >> 1) * 16 + 2 * k + (c & 1); }

// running state of a RANSAC call across its early-exit batches (lives in counters[LR_CNT_COUNT..], zeroed with NVALID)
struct lr_ransac_state {
    double T[12];                 // fp64 model of the best hypothesis so far
    unsigned long long ssq;
    long long n_valid, n_ids;
    uint32_t cnt;
    int32_t h;                    // its id (valid when cnt > 0)
    int32_t done;                 // set when the confidence test says stop: later batches return immediately
    int32_t lo_pending;           // the best model changed in the batch just merged: the local optimisation has to run on it
    int32_t lo_calls;             // local optimisations run so far (part of the key of their sample stream)
    int32_t lo_timeouts;          // diagnostic: waits of the local optimisation's hand-off protocol that hit their 0.2 s bound (low 16 bits: the master
                                  // recomputed a job alone; high 16 bits: a helper block left after 0.2 s without a job) -- 0 in every run seen
    // SPRT pre-verification (use_elc == 2): design of the current batch (0 = not designed yet: eps 0.1, delta 0.01) and the
    // statistics of the models rejected so far
    double sprt_eps, sprt_delta;
    unsigned long long rej_inl, rej_pts;
    // statistics of the scoring passes: (model, correspondence) evaluations done / what scanning every list in full would take
    unsigned long long evals, evals_full;
};
static_assert(sizeof(lr_ransac_state) <= (LR_CNT_TOTAL - LR_CNT_COUNT) * sizeof(int32_t), "lr_ransac_state does not fit");

// lr_api.hip: zero `bytes` of scratch at `p` (arena 0) in the arena of every pair of the call in flight
int lr_zero_scratch(lr_workspace *ws, void *p, size_t bytes, hipStream_t st);

// lr_nn16.hip
int lr_nn16_prep(lr_workspace *ws, const float *F0, int n0, const float *F1, int n1, hipStream_t st, bool zero_counters = false);
int lr_nn16_run(lr_workspace *ws, const float *Fq, const _Float16 *Hq, const float *nQ, int na,
                const float *Fc, const _Float16 *Hc, const float *nC, const float *range_c, int nb,
                int need, int32_t *idx1, int32_t *idx2, float *s1, float *s2, hipStream_t st, bool seed_reverse = false);
int lr_nn16_reverse(lr_workspace *ws, const float *F0, const _Float16 *H0, const float *nrm0, const float *bmax0, int n0,
                    const float *F1, const _Float16 *H1, const float *nrm1, int n1, const int32_t *fwd_idx1,
                    int32_t *rev, hipStream_t st, bool seeded = false);

// lr_filter.hip
int lr_mutual_run(lr_workspace *ws, int n0, const int32_t *idx1, const int32_t *idx2, const int32_t *rev,
                  uint8_t *is_bb, int32_t *o0, int32_t *o1, int32_t *o2, int32_t *n_out, hipStream_t st,
                  const float *xyz0 = nullptr, const float *xyz1 = nullptr, float *corr8 = nullptr);
int lr_identity_corr(lr_workspace *ws, int n0, const int32_t *idx1, const int32_t *idx2,
                     int32_t *o0, int32_t *o1, int32_t *o2, int32_t *n_out, hipStream_t st);
int lr_pack_corr(lr_workspace *ws, const float *xyz0, const float *xyz1, const int32_t *i0, const int32_t *i1, int m_max,
                 const int32_t *m_dev, float *corr8, hipStream_t st, const int32_t *rank = nullptr);
int lr_prosac_order(lr_workspace *ws, const float *F0, const float *F1, int dim, const float *quality, int m_max, const int32_t *m_dev,
                    hipStream_t st);
int lr_gpf_run(lr_workspace *ws, const float *F0, int n0, const float *F1, int dim,
               const int32_t *idx1, const int32_t *idx2, const uint8_t *is_bb, const float *xyz0,
               int grid_wid, double factor, int32_t *o0, int32_t *o1, int32_t *o2, float *oscore,
               int32_t *n_out, hipStream_t st, const float *xyz1 = nullptr, float *corr8 = nullptr);

int lr_gpf_bb_run(lr_workspace *ws, const float *F0, int n0, const float *F1, int dim,
                  const int32_t *b0, const int32_t *b1, const int32_t *b2, const int32_t *mb_dev, const float *xyz0,
                  int G, double max_matches, int32_t *o0, int32_t *o1, int32_t *o2, float *oscore,
                  int32_t *n_out, int32_t *has_score, hipStream_t st);

// lr_icp.hip
int lr_icp_run(lr_workspace *ws, const float *xyz0, int n0, const float *xyz1, int n1, const double *T_init,
               const lr_ransac_result *gate, double max_dist, int max_iter, double rel_fit, double rel_rmse,
               double *T_out, lr_icp_result *res, hipStream_t st);

// lr_ransac.hip
int lr_ransac_run(lr_workspace *ws, const float *corr8, int m_max, const int32_t *m_dev, const lr_ransac_params *p,
                  double *T_out, lr_ransac_result *res, hipStream_t st);
int lr_inlier_mask_run(const float *src, const float *tgt, const int32_t *i0, const int32_t *i1, int m_max, const int32_t *m_dev,
                       const double *T, float thr2, uint8_t *mask, int32_t *n_inliers, hipStream_t st);
int lr_refit_run(lr_workspace *ws, const float *xyz0, int n0, const float *xyz1, const int32_t *idx1,
                 const double *T_in, double thr2, double *T_out, int32_t *n_inl, const lr_ransac_result *gate,
                 hipStream_t st, lr_pair_result *pair_out = nullptr, const int32_t *idx0 = nullptr, const int32_t *m_dev = nullptr,
                 const float *F0 = nullptr, const float *F1 = nullptr);
