// Correspondence filtering on device: mutual-NN intersection, feature-distance ratio, Grid-Prioritized
// Filter, and packing of the surviving pairs for the RANSAC kernels.
//
// Replaces (reference Experiments/algorithms/matching.py): torch_intersect :67-87 (two sparse COO
// matrices + coalesce sort), nn_to_mutual :222-239, mark_best_buddies :207-220,
// calc_distance_ratio_in_feature_space :89-98 and Grid_Prioritized_Filter :100-205 (200 Python
// iterations of N-length numpy masks + per-cell argsort).  Everything here is index/compare work on
// <= N0 elements: HBM/latency-bound, a handful of small launches, no host round trips.
#include "lr_internal.h"
#include <math.h>

#define LR_INF __builtin_huge_valf()

// ------------------------------------------------------------------ ordered compaction
// Two small launches instead of a scan: (1) flags + per-block (256 elements) counts, (2) every block sums the
// counts of the blocks before it (<= n/256 values, one coalesced read) and scatters its survivors in order.
__device__ __forceinline__ void block_count(bool k, int32_t *__restrict__ blk_cnt)
{
    __shared__ int s_wave[4];
    const unsigned long long bal = __ballot(k);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

// keep[i] = rev[idx1[i]] == i  (torch_intersect, matching.py:67-87, reduced to its gather-compare core)
__global__ void __launch_bounds__(256)
mutual_flag_kernel(int n0, const int32_t *__restrict__ idx1, const int32_t *__restrict__ rev,
                   uint8_t *__restrict__ is_bb, int32_t *__restrict__ blk_cnt)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool k = i < n0 && rev[idx1[i]] == i;
    if (i < n0) is_bb[i] = k ? 1 : 0;
    block_count(k, blk_cnt);
}

__global__ void __launch_bounds__(256)
count_flags_kernel(int n0, const int32_t *__restrict__ m_dev, const uint8_t *__restrict__ flags, int32_t *__restrict__ blk_cnt)
{
    if (m_dev) n0 = min(n0, *m_dev);
    const int i = blockIdx.x * 256 + threadIdx.x;
    block_count(i < n0 && flags[i] != 0, blk_cnt);
}

// survivors in ascending i == torch coalesce order (matching.py:80-85) / boolean-mask order (matching.py:197-199)
__global__ void __launch_bounds__(256)
compact_kernel(int n0, const uint8_t *__restrict__ flags, const int32_t *__restrict__ blk_cnt,
               const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2, const float *__restrict__ score,
               int32_t *__restrict__ o0, int32_t *__restrict__ o1, int32_t *__restrict__ o2, float *__restrict__ oscore,
               int32_t *__restrict__ n_out, int32_t *__restrict__ n_out2,
               const float *__restrict__ xyz0, const float *__restrict__ xyz1, float *__restrict__ corr8, int32_t *__restrict__ counters,
               const int32_t *__restrict__ m_dev = nullptr, const int32_t *__restrict__ src0 = nullptr)
{
    __shared__ int s_wave[4];
    __shared__ int s_part[4];
    if (m_dev) n0 = min(n0, *m_dev);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int c = 0;
    for (int b = tid; b < (int)blockIdx.x; b += 256) c += blk_cnt[b];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m);
    const int i = blockIdx.x * 256 + tid;
    const bool k = i < n0 && flags[i] != 0;
    const unsigned long long bal = __ballot(k);
    if (lane == 0) { s_part[wave] = c; s_wave[wave] = __popcll(bal); }
    __syncthreads();
    const int prefix = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_wave[w];
    const int slot = prefix + woff + __popcll(bal & ((1ull << lane) - 1ull));
    if (k) {
        if (o0) o0[slot] = src0 ? src0[i] : i;
        if (o1) o1[slot] = idx1[i];
        if (o2 && idx2) o2[slot] = idx2[i];
        if (oscore && score) oscore[slot] = score[i];
        if (corr8) {        // fused pack_corr_kernel: the survivor's point pair in the RANSAC kernels' record layout
            const int b = idx1[i];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                corr8[lr_corr_at(slot, k)] = xyz0[3 * i + k];
                corr8[lr_corr_at(slot, 3 + k)] = xyz1[3 * b + k];
            }
        }
    }
    if (corr8 && blockIdx.x == 0 && tid < LR_CNT_TOTAL - LR_CNT_COUNT) {     // the RANSAC that follows starts from scratch
        counters[LR_CNT_COUNT + tid] = 0;
        if (tid == 0) counters[LR_CNT_NVALID] = 0;
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        const int total = prefix + s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (n_out) *n_out = total;
        if (n_out2) *n_out2 = total;
    }
}

int lr_mutual_run(lr_workspace *ws, int n0, const int32_t *idx1, const int32_t *idx2, const int32_t *rev,
                  uint8_t *is_bb, int32_t *o0, int32_t *o1, int32_t *o2, int32_t *n_out, hipStream_t st,
                  const float *xyz0, const float *xyz1, float *corr8)
{
    const int nb = lr_cdiv(n0, 256);
    uint8_t *flags = is_bb ? is_bb : ws->is_bb;
    hipLaunchKernelGGL(mutual_flag_kernel, dim3(nb), dim3(256), 0, st, n0, idx1, rev, flags, ws->blk_cnt);
    // the number of best buddies is wanted even when no list is (GPF's TOTAL_NUM, matching.py:115-116)
    hipLaunchKernelGGL(compact_kernel, dim3(nb), dim3(256), 0, st, n0, flags, ws->blk_cnt, idx1, idx2, (const float *)nullptr,
                       o0, o1, o2, (float *)nullptr, n_out, ws->counters + LR_CNT_NBB, xyz0, xyz1, corr8, ws->counters);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// mode "no_filter" (FR.py:53-54): every NN pair survives
__global__ void identity_corr_kernel(int n0, const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2,
                                     int32_t *o0, int32_t *o1, int32_t *o2, int32_t *n_out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n0) {
        o0[i] = i; o1[i] = idx1[i];
        if (o2 && idx2) o2[i] = idx2[i];
    }
    if (i == 0 && n_out) *n_out = n0;
}

int lr_identity_corr(lr_workspace *ws, int n0, const int32_t *idx1, const int32_t *idx2,
                     int32_t *o0, int32_t *o1, int32_t *o2, int32_t *n_out, hipStream_t st)
{
    (void)ws;
    hipLaunchKernelGGL(identity_corr_kernel, dim3(lr_cdiv(n0, 256)), dim3(256), 0, st, n0, idx1, idx2, o0, o1, o2, n_out);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// ------------------------------------------------------------------ pack survivors for RANSAC
// corr8: point pairs in the pair-interleaved record layout of lr_corr_at (one 64-byte scalar load per two
// correspondences in the scoring loop)
__global__ void pack_corr_kernel(const float *__restrict__ xyz0, const float *__restrict__ xyz1,
                                 const int32_t *__restrict__ i0, const int32_t *__restrict__ i1,
                                 int m_max, const int32_t *__restrict__ m_dev, float *__restrict__ corr8,
                                 int32_t *__restrict__ counters, const int32_t *__restrict__ rank)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < LR_CNT_TOTAL - LR_CNT_COUNT) counters[LR_CNT_COUNT + c] = 0;     // the RANSAC that follows starts from scratch
    if (c == 0) counters[LR_CNT_NVALID] = 0;
    int m = m_dev ? min(*m_dev, m_max) : m_max;
    if (c >= m) return;
    int a = i0 ? i0[c] : c, b = i1 ? i1[c] : c;
    const int dst = rank ? rank[c] : c;          // PROSAC: records in quality order
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        corr8[lr_corr_at(dst, k)] = xyz0[3 * a + k];
        corr8[lr_corr_at(dst, 3 + k)] = xyz1[3 * b + k];
    }
}

int lr_pack_corr(lr_workspace *ws, const float *xyz0, const float *xyz1, const int32_t *i0, const int32_t *i1, int m_max,
                 const int32_t *m_dev, float *corr8, hipStream_t st, const int32_t *rank)
{
    hipLaunchKernelGGL(pack_corr_kernel, dim3(lr_cdiv(m_max > 0 ? m_max : 1, 256)), dim3(256), 0, st, xyz0, xyz1, i0, i1, m_max, m_dev, corr8,
                       ws->counters, rank);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// ------------------------------------------------------------------ PROSAC order (GC_RANSAC.py:39-43)
// rank[c] = position of pair c when the pairs are sorted by ascending feature distance (= descending match quality,
// FR.py:74-80), ties by c (numpy's argsort leaves ties unspecified; a stable order keeps the result reproducible), NaN
// last.  Counting rank, O(M^2) compares on wave-uniform loads: M is a few 1e4 and the kernel is tens of microseconds.
__global__ void __launch_bounds__(256)
prosac_rank_kernel(const float *__restrict__ q, int m_max, const int32_t *__restrict__ m_dev, int32_t *__restrict__ rank)
{
    const int m = m_dev ? min(*m_dev, m_max) : m_max;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if ((int)blockIdx.x * 256 >= m) return;
    float v = c < m ? q[c] : 0.0f;
    if (!(v == v)) v = __builtin_huge_valf();
    int r = 0;
    for (int j = 0; j < m; ++j) {
        float w = q[j];                          // wave-uniform address: scalar load
        if (!(w == w)) w = __builtin_huge_valf();
        r += (w < v || (w == v && j < c)) ? 1 : 0;
    }
    if (c < m) rank[c] = r;
}

__global__ void ratio_kernel(const float *__restrict__ F0, const float *__restrict__ F1, int dim, int m, const int32_t *__restrict__ m_dev,
                             const int32_t *__restrict__ i0, const int32_t *__restrict__ i1, const int32_t *__restrict__ i2,
                             float *__restrict__ out);

// quality = feature-distance ratio of the listed pairs (FR.py:77) unless the caller has one already (GPF's
// norm_feat_dist, FR.py:75); then the ranks
int lr_prosac_order(lr_workspace *ws, const float *F0, const float *F1, int dim, const float *quality, int m_max, const int32_t *m_dev,
                    hipStream_t st)
{
    const int nb = lr_cdiv(m_max > 0 ? m_max : 1, 256);
    if (!quality) {
        hipLaunchKernelGGL(ratio_kernel, dim3(nb), dim3(256), 0, st, F0, F1, dim, m_max, m_dev, (const int32_t *)ws->corr_idx0,
                           (const int32_t *)ws->corr_idx1, (const int32_t *)ws->corr_idx2, ws->ratio);
        quality = ws->ratio;
    }
    hipLaunchKernelGGL(prosac_rank_kernel, dim3(nb), dim3(256), 0, st, quality, m_max, m_dev, ws->prosac_rank);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// ------------------------------------------------------------------ ratio (a6)
// ||A - B1|| / (||A - B2|| + 1e-6): direct differences, sequential k, no contraction (== oracle)
__global__ void __launch_bounds__(256)
ratio_kernel(const float *__restrict__ F0, const float *__restrict__ F1, int dim, int m, const int32_t *__restrict__ m_dev,
             const int32_t *__restrict__ i0, const int32_t *__restrict__ i1, const int32_t *__restrict__ i2,
             float *__restrict__ out)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (m_dev) m = min(m, *m_dev);
    if (c >= m) return;
    const float *a = F0 + (size_t)(i0 ? i0[c] : c) * dim;
    const float *b1 = F1 + (size_t)i1[c] * dim;
    const float *b2 = F1 + (size_t)i2[c] * dim;
    float s1 = 0.0f, s2 = 0.0f;
    for (int k = 0; k < dim; ++k) {
        float av = a[k];
        float e1 = av - b1[k], e2 = av - b2[k];
        float q1 = e1 * e1, q2 = e2 * e2;
        s1 = s1 + q1;
        s2 = s2 + q2;
    }
    float d1 = __builtin_sqrtf(s1), d2 = __builtin_sqrtf(s2);
    out[c] = ((d1) / (d2 + 1e-6f));
}

extern "C" int lr_feat_ratio(const float *F0, const float *F1, int dim, int m, const int32_t *i0, const int32_t *i1,
                             const int32_t *i2, float *out, void *stream)
{
    LR_REQUIRE(F0 && F1 && i1 && i2 && out && dim > 0 && m >= 0, LR_EINVAL, "lr_feat_ratio: bad argument");
    if (m == 0) return LR_OK;
    hipLaunchKernelGGL(ratio_kernel, dim3(lr_cdiv(m, 256)), dim3(256), 0, (hipStream_t)stream, F0, F1, dim, m,
                       (const int32_t *)nullptr, i0, i1, i2, out);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// ------------------------------------------------------------------ GPF (a7, BB_first=False)
// Step numbers follow matching.py:100-205.
//
// gpf_f layout: [0] min ratio [1] max ratio [2] min x [3] max x [4] min y [5] max y
// gpf_d layout: [0 .. G*G) cell counts (max_per_quad), [G*G .. 2 G*G) per_quad quota, then cell offsets as int

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fminf(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// min/max of the ratio and of x,y over the n0 pairs (one block; n0 is a few 1e4)
__global__ void __launch_bounds__(1024)
gpf_minmax_kernel(int n0, const float *__restrict__ ratio, const float *__restrict__ xyz0, float *__restrict__ gf,
                  const int32_t *__restrict__ m_dev = nullptr, const int32_t *__restrict__ pidx = nullptr)
{
    __shared__ float sm[6][16];
    if (m_dev) n0 = min(n0, *m_dev);
    float lo[3] = { LR_INF, LR_INF, LR_INF }, hi[3] = { -LR_INF, -LR_INF, -LR_INF };
    for (int i = threadIdx.x; i < n0; i += 1024) {
        const int pi = pidx ? pidx[i] : i;
        float v[3] = { ratio[i], xyz0[3 * pi], xyz0[3 * pi + 1] };
#pragma unroll
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], v[k]); hi[k] = fmaxf(hi[k], v[k]); }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float a = wave_min(lo[k]), b = wave_max(hi[k]);
        if (lane == 0) { sm[2 * k][wave] = a; sm[2 * k + 1][wave] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = sm[threadIdx.x][0];
        for (int w = 1; w < 16; ++w) r = (threadIdx.x & 1) ? fmaxf(r, sm[threadIdx.x][w]) : fminf(r, sm[threadIdx.x][w]);
        gf[threadIdx.x] = r;
    }
}

// normalised score (matching.py:118-134) and grid cell (matching.py:136-146), all in fp32 as torch does
__global__ void __launch_bounds__(256)
gpf_score_cell_kernel(int n0, int G, const float *__restrict__ gf, const uint8_t *__restrict__ is_bb,
                      const float *__restrict__ xyz0, float *__restrict__ ratio_inout, int32_t *__restrict__ cell,
                      int32_t *__restrict__ cell_count, const int32_t *__restrict__ m_dev = nullptr,
                      const int32_t *__restrict__ pidx = nullptr)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (m_dev) n0 = min(n0, *m_dev);
    if (i >= n0) return;
    const int pi = pidx ? pidx[i] : i;
    const float m = gf[0], M = gf[1];
    float nfd = ((ratio_inout[i] - m) / (M - m));
    if (is_bb && is_bb[i]) nfd = nfd - 1.0f;          // BB_first=True has no best-buddy shift (matching.py:126)
    ratio_inout[i] = nfd;
    const float denx = (gf[3] - gf[2]) + 1e-3f, deny = (gf[5] - gf[4]) + 1e-3f;
    float qx = floorf((float)G * ((xyz0[3 * pi] - gf[2]) / (denx)));
    float qy = floorf((float)G * ((xyz0[3 * pi + 1] - gf[4]) / (deny)));
    int c = (int)qx * G + (int)qy;
    cell[i] = c;
    atomicAdd(&cell_count[c], 1);
}

// water-filling bisection in fp64 exactly as matching.py:154-179, then exclusive cell offsets; one thread
// total_fixed >= 0 selects the BB_first=True form: TOTAL = GPF_max_matches, and nothing is filtered (has_score = 0) when
// the mutual set is already that small (matching.py:109-113)
__global__ void gpf_waterfill_kernel(int G, double factor, const int32_t *__restrict__ counters,
                                     const int32_t *__restrict__ cell_count, double *__restrict__ quota,
                                     int32_t *__restrict__ cell_off, double total_fixed = -1.0,
                                     const int32_t *__restrict__ m_dev = nullptr, int32_t *__restrict__ has_score = nullptr)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int C = G * G;
    const double TOTAL = total_fixed >= 0.0 ? total_fixed : factor * (double)counters[LR_CNT_NBB];
    if (total_fixed >= 0.0) {
        const bool keep_all = TOTAL >= (double)*m_dev;
        if (has_score) *has_score = keep_all ? 0 : 1;
        if (keep_all) {
            int off = 0;
            for (int c = 0; c < C; ++c) { quota[c] = (double)cell_count[c]; cell_off[c] = off; off += cell_count[c]; }
            cell_off[C] = off;
            return;
        }
    }
    auto total_at = [&](double h) {
        double s = 0.0;
        for (int c = 0; c < C; ++c) { double m = (double)cell_count[c]; s += (m < h) ? m : h; }
        return s;
    };
    double max_h = TOTAL, min_h = 0.0, cur = (max_h + min_h) / 2;
    while (fabs(max_h - min_h) > 2) {
        double t = total_at(cur);
        if (t == TOTAL) break;
        else if (t < TOTAL) min_h = cur;
        else if (t > TOTAL) max_h = cur;
        cur = (max_h + min_h) / 2;
    }
    const double hr = rint(cur);                      // np.round: half to even
    int off = 0;
    for (int c = 0; c < C; ++c) {
        double m = (double)cell_count[c];
        quota[c] = (m < hr) ? m : hr;
        cell_off[c] = off;
        off += cell_count[c];
    }
    cell_off[C] = off;
}

// bucket pair ids by cell (order inside a bucket is irrelevant: ranks below use (score, id))
__global__ void __launch_bounds__(256)
gpf_bucket_kernel(int n0, const int32_t *__restrict__ cell, const int32_t *__restrict__ cell_off,
                  int32_t *__restrict__ cell_fill, int32_t *__restrict__ bucket, const int32_t *__restrict__ m_dev = nullptr)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (m_dev) n0 = min(n0, *m_dev);
    if (i >= n0) return;
    int c = cell[i];
    int pos = atomicAdd(&cell_fill[c], 1);
    bucket[cell_off[c] + pos] = i;
}

// keep[i] = all of the cell if quota == count, else rank of (score, i) inside the cell < quota
// (matching.py:184-195; ties in torch.argsort are unspecified, resolved towards the lower pair id)
__global__ void __launch_bounds__(256)
gpf_select_kernel(int n0, const int32_t *__restrict__ cell, const int32_t *__restrict__ cell_off,
                  const int32_t *__restrict__ cell_count, const double *__restrict__ quota,
                  const int32_t *__restrict__ bucket, const float *__restrict__ score, uint8_t *__restrict__ keep,
                  const int32_t *__restrict__ m_dev = nullptr)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (m_dev) n0 = min(n0, *m_dev);
    if (i >= n0) return;
    const int c = cell[i];
    const int q = (int)quota[c];
    bool k = false;
    if (q > 0) {
        if (quota[c] == (double)cell_count[c]) k = true;
        else {
            const float s = score[i];
            int rank = 0;
            const int b = cell_off[c], e = b + cell_count[c];
            for (int t = b; t < e; ++t) {
                int j = bucket[t];
                float sj = score[j];
                rank += (sj < s || (sj == s && j < i)) ? 1 : 0;
            }
            k = rank < q;
        }
    }
    keep[i] = k ? 1 : 0;
}

int lr_gpf_run(lr_workspace *ws, const float *F0, int n0, const float *F1, int dim,
               const int32_t *idx1, const int32_t *idx2, const uint8_t *is_bb, const float *xyz0,
               int G, double factor, int32_t *o0, int32_t *o1, int32_t *o2, float *oscore,
               int32_t *n_out, hipStream_t st, const float *xyz1, float *corr8)
{
    LR_REQUIRE(G >= 1 && G <= 64, LR_EINVAL, "lr_gpf: grid width must be in [1,64]");
    int32_t *cell_count = ws->gpf_cells;
    int32_t *cell_fill = cell_count + LR_GPF_MAX_CELLS + 8;
    int32_t *cell_off = cell_fill + LR_GPF_MAX_CELLS + 8;
    double *quota = ws->gpf_quota;
    uint8_t *keep = ws->gpf_keep;
    const int nb = lr_cdiv(n0, 256);
    LR_HIP(hipMemsetAsync(cell_count, 0, sizeof(int32_t) * 2 * (LR_GPF_MAX_CELLS + 8), st));
    // ratio over all n0 NN pairs (corres_idx0 == arange)
    hipLaunchKernelGGL(ratio_kernel, dim3(nb), dim3(256), 0, st, F0, F1, dim, n0, (const int32_t *)nullptr,
                       (const int32_t *)nullptr, idx1, idx2, ws->ratio);
    hipLaunchKernelGGL(gpf_minmax_kernel, dim3(1), dim3(1024), 0, st, n0, ws->ratio, xyz0, ws->gpf_f);
    hipLaunchKernelGGL(gpf_score_cell_kernel, dim3(nb), dim3(256), 0, st, n0, G, ws->gpf_f, is_bb, xyz0, ws->ratio,
                       ws->cell, cell_count);
    hipLaunchKernelGGL(gpf_waterfill_kernel, dim3(1), dim3(64), 0, st, G, factor, ws->counters, cell_count, quota, cell_off);
    hipLaunchKernelGGL(gpf_bucket_kernel, dim3(nb), dim3(256), 0, st, n0, ws->cell, cell_off, cell_fill, ws->cell_sorted);
    hipLaunchKernelGGL(gpf_select_kernel, dim3(nb), dim3(256), 0, st, n0, ws->cell, cell_off, cell_count, quota,
                       ws->cell_sorted, ws->ratio, keep);
    hipLaunchKernelGGL(count_flags_kernel, dim3(nb), dim3(256), 0, st, n0, (const int32_t *)nullptr, keep, ws->blk_cnt);
    hipLaunchKernelGGL(compact_kernel, dim3(nb), dim3(256), 0, st, n0, keep, ws->blk_cnt, idx1, idx2, ws->ratio, o0, o1, o2, oscore,
                       n_out, (int32_t *)nullptr, xyz0, xyz1, corr8, ws->counters);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// Grid_Prioritized_Filter(BB_first=True) (matching.py:100-205 with :109-113,126): the filter runs over the MUTUAL pairs
// (b0,b1,b2 of live length *mb_dev), TOTAL_NUM = GPF_max_matches, no best-buddy shift.  Used by the reference's TEASER
// wrapper only (TEASER_plus_plus.py:109-110).
int lr_gpf_bb_run(lr_workspace *ws, const float *F0, int n0, const float *F1, int dim,
                  const int32_t *b0, const int32_t *b1, const int32_t *b2, const int32_t *mb_dev, const float *xyz0,
                  int G, double max_matches, int32_t *o0, int32_t *o1, int32_t *o2, float *oscore,
                  int32_t *n_out, int32_t *has_score, hipStream_t st)
{
    LR_REQUIRE(G >= 1 && G <= 64, LR_EINVAL, "lr_gpf: grid width must be in [1,64]");
    int32_t *cell_count = ws->gpf_cells;
    int32_t *cell_fill = cell_count + LR_GPF_MAX_CELLS + 8;
    int32_t *cell_off = cell_fill + LR_GPF_MAX_CELLS + 8;
    double *quota = ws->gpf_quota;
    uint8_t *keep = ws->gpf_keep;
    const int nb = lr_cdiv(n0, 256);
    LR_HIP(hipMemsetAsync(cell_count, 0, sizeof(int32_t) * 2 * (LR_GPF_MAX_CELLS + 8), st));
    hipLaunchKernelGGL(ratio_kernel, dim3(nb), dim3(256), 0, st, F0, F1, dim, n0, mb_dev, b0, b1, b2, ws->ratio);
    hipLaunchKernelGGL(gpf_minmax_kernel, dim3(1), dim3(1024), 0, st, n0, ws->ratio, xyz0, ws->gpf_f, mb_dev, b0);
    hipLaunchKernelGGL(gpf_score_cell_kernel, dim3(nb), dim3(256), 0, st, n0, G, ws->gpf_f, (const uint8_t *)nullptr, xyz0, ws->ratio,
                       ws->cell, cell_count, mb_dev, b0);
    hipLaunchKernelGGL(gpf_waterfill_kernel, dim3(1), dim3(64), 0, st, G, 0.0, ws->counters, cell_count, quota, cell_off, max_matches,
                       mb_dev, has_score);
    hipLaunchKernelGGL(gpf_bucket_kernel, dim3(nb), dim3(256), 0, st, n0, ws->cell, cell_off, cell_fill, ws->cell_sorted, mb_dev);
    hipLaunchKernelGGL(gpf_select_kernel, dim3(nb), dim3(256), 0, st, n0, ws->cell, cell_off, cell_count, quota,
                       ws->cell_sorted, ws->ratio, keep, mb_dev);
    hipLaunchKernelGGL(count_flags_kernel, dim3(nb), dim3(256), 0, st, n0, mb_dev, keep, ws->blk_cnt);
    hipLaunchKernelGGL(compact_kernel, dim3(nb), dim3(256), 0, st, n0, keep, ws->blk_cnt, b1, b2, ws->ratio, o0, o1, o2, oscore,
                       n_out, (int32_t *)nullptr, (const float *)nullptr, (const float *)nullptr, (float *)nullptr, ws->counters, mb_dev, b0);
    LR_LAUNCH_CHECK();
    return LR_OK;
}
