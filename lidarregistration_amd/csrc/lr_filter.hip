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
                   uint8_t *__restrict__ is_bb, int32_t *__restrict__ blk_cnt, lr_zargs z)
{
    if (z.descs) n0 = z.descs[blockIdx.z].n0;
    lr_z(idx1, z, blockIdx.z); lr_z(rev, z, blockIdx.z); lr_z(is_bb, z, blockIdx.z); lr_z(blk_cnt, z, blockIdx.z);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool k = i < n0 && rev[idx1[i]] == i;
    if (i < n0) is_bb[i] = k ? 1 : 0;
    block_count(k, blk_cnt);
}

__global__ void __launch_bounds__(256)
count_flags_kernel(int n0, const int32_t *__restrict__ m_dev, const uint8_t *__restrict__ flags, int32_t *__restrict__ blk_cnt, lr_zargs z)
{
    if (z.descs) n0 = z.descs[blockIdx.z].n0;
    lr_z(m_dev, z, blockIdx.z); lr_z(flags, z, blockIdx.z); lr_z(blk_cnt, z, blockIdx.z);
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
               lr_zargs z, const int32_t *__restrict__ m_dev = nullptr, const int32_t *__restrict__ src0 = nullptr)
{
    __shared__ int s_wave[4];
    __shared__ int s_part[4];
    if (z.descs) { const lr_pair_desc d = z.descs[blockIdx.z]; n0 = d.n0; if (xyz0) { xyz0 = d.xyz0; xyz1 = d.xyz1; } }
    lr_z(flags, z, blockIdx.z); lr_z(blk_cnt, z, blockIdx.z); lr_z(idx1, z, blockIdx.z); lr_z(idx2, z, blockIdx.z); lr_z(score, z, blockIdx.z); lr_z(o0, z, blockIdx.z); lr_z(o1, z, blockIdx.z); lr_z(o2, z, blockIdx.z); lr_z(oscore, z, blockIdx.z); lr_z(n_out, z, blockIdx.z); lr_z(n_out2, z, blockIdx.z); lr_z(corr8, z, blockIdx.z); lr_z(counters, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(src0, z, blockIdx.z);
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
        if (tid == 0) { counters[LR_CNT_NVALID] = 0; counters[LR_CNT_NVALID2] = 0; }
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
    hipLaunchKernelGGL(mutual_flag_kernel, dim3(nb, 1, ws->zP), dim3(256), 0, st, n0, idx1, rev, flags, ws->blk_cnt, ws->z);
    // the number of best buddies is wanted even when no list is (GPF's TOTAL_NUM, matching.py:115-116)
    hipLaunchKernelGGL(compact_kernel, dim3(nb, 1, ws->zP), dim3(256), 0, st, n0, flags, ws->blk_cnt, idx1, idx2, (const float *)nullptr,
                       o0, o1, o2, (float *)nullptr, n_out, ws->counters + LR_CNT_NBB, xyz0, xyz1, corr8, ws->counters, ws->z);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// mode "no_filter" (FR.py:53-54): every NN pair survives
__global__ void identity_corr_kernel(int n0, const int32_t *__restrict__ idx1, const int32_t *__restrict__ idx2,
                                     int32_t *o0, int32_t *o1, int32_t *o2, int32_t *n_out, lr_zargs z)
{
    if (z.descs) n0 = z.descs[blockIdx.z].n0;
    lr_z(idx1, z, blockIdx.z); lr_z(idx2, z, blockIdx.z); lr_z(o0, z, blockIdx.z); lr_z(o1, z, blockIdx.z); lr_z(o2, z, blockIdx.z); lr_z(n_out, z, blockIdx.z);
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
    hipLaunchKernelGGL(identity_corr_kernel, dim3(lr_cdiv(n0, 256), 1, ws->zP), dim3(256), 0, st, n0, idx1, idx2, o0, o1, o2, n_out, ws->z);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// ------------------------------------------------------------------ pack survivors for RANSAC
// corr8: point pairs in the pair-interleaved record layout of lr_corr_at (one 64-byte scalar load per two
// correspondences in the scoring loop)
__global__ void pack_corr_kernel(const float *__restrict__ xyz0, const float *__restrict__ xyz1,
                                 const int32_t *__restrict__ i0, const int32_t *__restrict__ i1,
                                 int m_max, const int32_t *__restrict__ m_dev, float *__restrict__ corr8,
                                 int32_t *__restrict__ counters, const int32_t *__restrict__ rank, lr_zargs z)
{
    if (z.descs) { xyz0 = z.descs[blockIdx.z].xyz0; xyz1 = z.descs[blockIdx.z].xyz1; }
    lr_z(i0, z, blockIdx.z); lr_z(i1, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(corr8, z, blockIdx.z); lr_z(counters, z, blockIdx.z); lr_z(rank, z, blockIdx.z);
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < LR_CNT_TOTAL - LR_CNT_COUNT) counters[LR_CNT_COUNT + c] = 0;     // the RANSAC that follows starts from scratch
    if (c == 0) { counters[LR_CNT_NVALID] = 0; counters[LR_CNT_NVALID2] = 0; }
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
    hipLaunchKernelGGL(pack_corr_kernel, dim3(lr_cdiv(m_max > 0 ? m_max : 1, 256), 1, ws->zP), dim3(256), 0, st, xyz0, xyz1, i0, i1, m_max, m_dev, corr8,
                       ws->counters, rank, ws->z);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// ------------------------------------------------------------------ PROSAC order (GC_RANSAC.py:39-43)
// rank[c] = position of pair c when the pairs are sorted by ascending feature distance (= descending match quality,
// FR.py:74-80), ties by c (numpy's argsort leaves ties unspecified; a stable order keeps the result reproducible), NaN
// last.  Bucket sort: LR_PR_BUCKETS (8192) linear buckets over the value range, then the exact rank inside the bucket by comparing with
// its members -- O(M x bucket size) instead of the O(M^2) of a plain counting rank (M can be all N pairs under GPF).
__device__ __forceinline__ float pr_key(float v) { return v == v ? v : __builtin_huge_valf(); }
__device__ __forceinline__ int pr_bucket(float v, float lo, float scale)
{
    if (!(v < __builtin_huge_valf())) return LR_PR_BUCKETS - 1;
    const int b = (int)((v - lo) * scale);
    return min(max(b, 0), LR_PR_BUCKETS - 1);
}

// one block per pair: range of the finite values, bucket histogram in LDS, exclusive offsets -> offs[0..B]; then the members of every bucket
// with their keys, bucket by bucket (members / mkeys; the order inside a bucket is whatever the LDS position counters hand out -- the
// ranks below compare (key, index), so it does not matter).  The scatter used to be a launch of its own with one returning device-scope
// atomic per pair of the list (67 us per 32 x 30k pairs, on every CU); here the counters are LDS words of the block that has the
// histogram anyway, and a one-block-per-pair kernel hides behind the other calls' filter passes.
template <bool FUSED>
__global__ void __launch_bounds__(1024)
prosac_scan_kernel(const float *__restrict__ q, int m_max, const int32_t *__restrict__ m_dev, int32_t *__restrict__ offs, int32_t *__restrict__ fill,
                   int32_t *__restrict__ members, float *__restrict__ mkeys, float *__restrict__ range, lr_zargs z)
{
    lr_z(q, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(offs, z, blockIdx.z); lr_z(fill, z, blockIdx.z); lr_z(members, z, blockIdx.z); lr_z(mkeys, z, blockIdx.z); lr_z(range, z, blockIdx.z);
    __shared__ int s_h[LR_PR_BUCKETS];
    __shared__ float s_lo[16], s_hi[16];
    __shared__ int s_w[16];
    const int m = m_dev ? min(*m_dev, m_max) : m_max;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float lo = __builtin_huge_valf(), hi = -__builtin_huge_valf();
    for (int i = threadIdx.x; i < m; i += 1024) { const float v = pr_key(q[i]); if (v < __builtin_huge_valf()) { lo = fminf(lo, v); hi = fmaxf(hi, v); } }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
    if (lane == 0) { s_lo[wave] = lo; s_hi[wave] = hi; }
    for (int k = threadIdx.x; k < LR_PR_BUCKETS; k += 1024) s_h[k] = 0;
    __syncthreads();
    for (int w = 0; w < 16; ++w) { lo = fminf(lo, s_lo[w]); hi = fmaxf(hi, s_hi[w]); }
    const float scale = hi > lo ? (float)LR_PR_BUCKETS / (hi - lo) : 0.0f;
    if (threadIdx.x == 0) { range[0] = lo; range[1] = scale; }
    for (int i0 = threadIdx.x; i0 < m; i0 += 8 * 1024) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = q[min(i0 + 1024 * k, m - 1)];
#pragma unroll
        for (int k = 0; k < 8; ++k) if (i0 + 1024 * k < m) atomicAdd(&s_h[pr_bucket(pr_key(v[k]), lo, scale)], 1);
    }
    __syncthreads();
    int v[LR_PR_BUCKETS / 1024], sum = 0;
#pragma unroll
    for (int k = 0; k < LR_PR_BUCKETS / 1024; ++k) { v[k] = s_h[threadIdx.x * (LR_PR_BUCKETS / 1024) + k]; sum += v[k]; }
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int w = 0; w < wave; ++w) run += s_w[w];
#pragma unroll
    for (int k = 0; k < LR_PR_BUCKETS / 1024; ++k) {
        const int b = threadIdx.x * (LR_PR_BUCKETS / 1024) + k;
        offs[b] = run;
        if constexpr (FUSED) s_h[b] = run;          // (s_h: from here on the next free position of the bucket)
        else fill[b] = run;                         // (calls of a few pairs: the scatter is a launch of its own, below)
        run += v[k];
    }
    if (threadIdx.x == 1023) offs[LR_PR_BUCKETS] = run;
    if constexpr (!FUSED) return;
    __syncthreads();
    for (int i0 = threadIdx.x; i0 < m; i0 += 8 * 1024) {
        float w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = pr_key(q[min(i0 + 1024 * k, m - 1)]);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (i0 + 1024 * k < m) {
                const int pos = atomicAdd(&s_h[pr_bucket(w[k], lo, scale)], 1);
                members[pos] = i0 + 1024 * k; mkeys[pos] = w[k];          // the key travels with the member: the ranking loop has no dependent loads
            }
    }
}

// the scatter as a launch of its own, for calls of a few pairs (the GPU is idle next to the one block per pair above: a lone 30k pair pays
// 26 us for the fused form, 10 + 5 for scan + this)
__global__ void __launch_bounds__(256)
prosac_scatter_kernel(const float *__restrict__ q, int m_max, const int32_t *__restrict__ m_dev, const float *__restrict__ range,
                      int32_t *__restrict__ fill, int32_t *__restrict__ members, float *__restrict__ mkeys, lr_zargs z)
{
    lr_z(q, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(range, z, blockIdx.z); lr_z(fill, z, blockIdx.z); lr_z(members, z, blockIdx.z); lr_z(mkeys, z, blockIdx.z);
    const int m = m_dev ? min(*m_dev, m_max) : m_max;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= m) return;
    const float v = pr_key(q[c]);
    const int pos = atomicAdd(&fill[pr_bucket(v, range[0], range[1])], 1);
    members[pos] = c; mkeys[pos] = v;
}

// thread = POSITION t of the bucket-ordered list (not pair c: the members of a pair's bucket lie around t, so the loads of a wave are
// neighbours -- indexed by pair they were 2 x 30k scattered sectors per cloud, 95 us per 32 x 30k pairs)
__global__ void __launch_bounds__(256)
prosac_rank_kernel(int m_max, const int32_t *__restrict__ m_dev, const float *__restrict__ range,
                   const int32_t *__restrict__ offs, const int32_t *__restrict__ members, const float *__restrict__ mkeys,
                   int32_t *__restrict__ rank, lr_zargs z)
{
    lr_z(m_dev, z, blockIdx.z); lr_z(range, z, blockIdx.z); lr_z(offs, z, blockIdx.z); lr_z(members, z, blockIdx.z); lr_z(mkeys, z, blockIdx.z);
    lr_z(rank, z, blockIdx.z);
    const int m = m_dev ? min(*m_dev, m_max) : m_max;
    const int t0 = blockIdx.x * 256 + threadIdx.x;
    if (t0 >= m) return;
    const int c = members[t0];
    const float v = mkeys[t0];
    const int b = pr_bucket(v, range[0], range[1]);
    const int e = offs[b + 1];
    int r = offs[b];
    // four members per step, their eight loads independent (the quality distribution decides the bucket sizes: GPF's normalised distance
    // piles a tenth of the pairs up just below 1)
    for (int t = offs[b]; t < e; t += 4) {
        int j[4]; float w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int tt = min(t + u, e - 1); j[u] = members[tt]; w[u] = mkeys[tt]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) r += (t + u < e && (w[u] < v || (w[u] == v && j[u] < c))) ? 1 : 0;
    }
    rank[c] = r;
}

__global__ void ratio_kernel(const float *__restrict__ F0, const float *__restrict__ F1, int dim, int m, const int32_t *__restrict__ m_dev,
                             const int32_t *__restrict__ i0, const int32_t *__restrict__ i1, const int32_t *__restrict__ i2,
                             float *__restrict__ out, uint32_t *__restrict__ mm, const float *__restrict__ xyz0, lr_zargs z);

// quality = feature-distance ratio of the listed pairs (FR.py:77) unless the caller has one already (GPF's
// norm_feat_dist, FR.py:75); then the ranks
int lr_prosac_order(lr_workspace *ws, const float *F0, const float *F1, int dim, const float *quality, int m_max, const int32_t *m_dev,
                    hipStream_t st)
{
    const int nb = lr_cdiv(m_max > 0 ? m_max : 1, 256);
    if (!quality) {
        hipLaunchKernelGGL(ratio_kernel, dim3(nb, 1, ws->zP), dim3(256), 0, st, F0, F1, dim, m_max, m_dev, (const int32_t *)ws->corr_idx0,
                           (const int32_t *)ws->corr_idx1, (const int32_t *)ws->corr_idx2, ws->ratio, (uint32_t *)nullptr, (const float *)nullptr, ws->z);
        quality = ws->ratio;
    }
    // GPF's scratch is free by now: offsets | fill | range in its cell arrays, bucket members in its sort buffer
    int32_t *offs = ws->gpf_cells, *fill = offs + LR_PR_BUCKETS + 8;
    float *range = reinterpret_cast<float *>(fill + LR_PR_BUCKETS + 8);
    float *mkeys = reinterpret_cast<float *>(ws->cell);       // (GPF's cell ids: free by now, like the rest of its scratch)
    const bool fused = ws->zP > 4;      // (full batches: the scatter inside the one-block-per-pair scan, hidden behind the other calls' filter passes)
    if (fused)
        hipLaunchKernelGGL(prosac_scan_kernel<true>, dim3(1, 1, ws->zP), dim3(1024), 0, st, quality, m_max, m_dev, offs, fill, ws->cell_sorted, mkeys, range, ws->z);
    else {
        hipLaunchKernelGGL(prosac_scan_kernel<false>, dim3(1, 1, ws->zP), dim3(1024), 0, st, quality, m_max, m_dev, offs, fill, ws->cell_sorted, mkeys, range, ws->z);
        hipLaunchKernelGGL(prosac_scatter_kernel, dim3(nb, 1, ws->zP), dim3(256), 0, st, quality, m_max, m_dev, (const float *)range, fill, ws->cell_sorted, mkeys, ws->z);
    }
    hipLaunchKernelGGL(prosac_rank_kernel, dim3(nb, 1, ws->zP), dim3(256), 0, st, m_max, m_dev, (const float *)range, (const int32_t *)offs,
                       (const int32_t *)ws->cell_sorted, (const float *)mkeys, ws->prosac_rank, ws->z);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// ------------------------------------------------------------------ ratio (a6)
// order-preserving map float -> uint32 (and back): lets min / max run as integer atomicMax on a zero-initialised slot
__device__ __forceinline__ uint32_t lr_enc(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float lr_dec(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e ^ 0x80000000u) : ~e); }

// ||A - B1|| / (||A - B2|| + 1e-6): direct differences, sequential k, no contraction (== oracle).
// mm (optional, 6 zero-initialised slots): min/max of the ratio and of x, y of the pairs' cloud-0 points, kept as
// atomicMax of enc(-v) / enc(v) (matching.py:118-122,136-141 need them; NaN values are ignored like fminf/fmaxf do).
__global__ void __launch_bounds__(256)
ratio_kernel(const float *__restrict__ F0, const float *__restrict__ F1, int dim, int m, const int32_t *__restrict__ m_dev,
             const int32_t *__restrict__ i0, const int32_t *__restrict__ i1, const int32_t *__restrict__ i2,
             float *__restrict__ out, uint32_t *__restrict__ mm, const float *__restrict__ xyz0, lr_zargs z)
{
    __shared__ float s_r[6][4];
    if (z.descs) { const lr_pair_desc d = z.descs[blockIdx.z]; F0 = d.F0; F1 = d.F1; if (xyz0) xyz0 = d.xyz0; if (!m_dev) m = min(m, d.n0); }
    lr_z(m_dev, z, blockIdx.z); lr_z(i0, z, blockIdx.z); lr_z(i1, z, blockIdx.z); lr_z(i2, z, blockIdx.z); lr_z(out, z, blockIdx.z); lr_z(mm, z, blockIdx.z);
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (m_dev) m = min(m, *m_dev);
    float v[3] = { 0.0f, 0.0f, 0.0f };
    const bool live = c < m;
    if (dim == 32) {
        // FCGF's 32 dimensions (block-uniform branch): EIGHT lanes per pair, lane t holding elements 4 t .. 4 t + 3 of the three rows,
        // so a load instruction of the wave fetches 8 whole 128-byte rows instead of 16 bytes of 64 different ones (the
        // one-thread-per-pair form spent its time in the texture addresser: 110 us per 32 x 30k pairs).  The sums still run over
        // k = 0, 1, 2, ... in order: the running sums are handed from lane t to lane t + 1, four additions per hop.  The block's 256
        // pairs are taken in 8 rounds of 32; lane 7 of a group leaves the round's ratio in LDS, thread i picks up pair i's.
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        __shared__ int s_i[3][256];
        __shared__ float s_ratio[256];
        const int tid = threadIdx.x, g = tid >> 3, t = tid & 7;
        s_i[0][tid] = live ? (i0 ? i0[c] : c) : 0; s_i[1][tid] = live ? i1[c] : 0; s_i[2][tid] = live ? i2[c] : 0;
        __syncthreads();
#pragma unroll 2
        for (int r = 0; r < 8; r += 4) {
            f32x4 av[4], pv[4], qv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int row = (r + u) * 32 + g;
                av[u] = reinterpret_cast<const f32x4 *>(F0 + (size_t)s_i[0][row] * 32)[t];
                pv[u] = reinterpret_cast<const f32x4 *>(F1 + (size_t)s_i[1][row] * 32)[t];
                qv[u] = reinterpret_cast<const f32x4 *>(F1 + (size_t)s_i[2][row] * 32)[t];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float q1[4], q2[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float e1 = av[u][e] - pv[u][e], e2 = av[u][e] - qv[u][e];
                    q1[e] = e1 * e1; q2[e] = e2 * e2;
                }
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int hop = 0; hop < 8; ++hop) {
                    const float in1 = __shfl_up(s1, 1, 8), in2 = __shfl_up(s2, 1, 8);
                    float c1 = t ? in1 : 0.0f, c2 = t ? in2 : 0.0f;
                    c1 = c1 + q1[0]; c2 = c2 + q2[0];
                    c1 = c1 + q1[1]; c2 = c2 + q2[1];
                    c1 = c1 + q1[2]; c2 = c2 + q2[2];
                    c1 = c1 + q1[3]; c2 = c2 + q2[3];
                    if (t == hop) { s1 = c1; s2 = c2; }
                }
                if (t == 7) {
                    float d1 = __builtin_sqrtf(s1), d2 = __builtin_sqrtf(s2);
                    s_ratio[(r + u) * 32 + g] = ((d1) / (d2 + 1e-6f));
                }
            }
        }
        __syncthreads();
        if (live) {
            v[0] = s_ratio[tid];
            out[c] = v[0];
            if (mm) { const int pa = s_i[0][tid]; v[1] = xyz0[3 * pa]; v[2] = xyz0[3 * pa + 1]; }
        }
    } else if (live) {
        const int pa = i0 ? i0[c] : c;
        const float *a = F0 + (size_t)pa * dim;
        const float *b1 = F1 + (size_t)i1[c] * dim;
        const float *b2 = F1 + (size_t)i2[c] * dim;
        float s1 = 0.0f, s2 = 0.0f;
        if ((dim & 3) == 0) {
            // 16-byte loads; the sums still run over k = 0, 1, 2, ... in order (rows of a [n, dim] float array with
            // dim % 4 == 0 are 16-byte aligned whenever the array is, and torch / hipMalloc allocations are)
            typedef float f32x4 __attribute__((ext_vector_type(4)));
            const f32x4 *a4 = reinterpret_cast<const f32x4 *>(a), *p4 = reinterpret_cast<const f32x4 *>(b1), *q4 = reinterpret_cast<const f32x4 *>(b2);
            for (int k = 0; k < dim / 4; ++k) {
                const f32x4 av = a4[k], pv = p4[k], qv = q4[k];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float e1 = av[t] - pv[t], e2 = av[t] - qv[t];
                    float q1 = e1 * e1, q2 = e2 * e2;
                    s1 = s1 + q1;
                    s2 = s2 + q2;
                }
            }
        } else {
            for (int k = 0; k < dim; ++k) {
                float av = a[k];
                float e1 = av - b1[k], e2 = av - b2[k];
                float q1 = e1 * e1, q2 = e2 * e2;
                s1 = s1 + q1;
                s2 = s2 + q2;
            }
        }
        float d1 = __builtin_sqrtf(s1), d2 = __builtin_sqrtf(s2);
        v[0] = ((d1) / (d2 + 1e-6f));
        out[c] = v[0];
        if (mm) { v[1] = xyz0[3 * pa]; v[2] = xyz0[3 * pa + 1]; }
    }
    if (!mm) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float lo = live ? v[k] : __builtin_huge_valf(), hi = live ? v[k] : -__builtin_huge_valf();
        if (!(lo == lo)) { lo = __builtin_huge_valf(); hi = -__builtin_huge_valf(); }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); }
        if (lane == 0) { s_r[2 * k][wave] = lo; s_r[2 * k + 1][wave] = hi; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float r = s_r[k][0];
        for (int w = 1; w < 4; ++w) r = (k & 1) ? fmaxf(r, s_r[k][w]) : fminf(r, s_r[k][w]);
        atomicMax(&mm[k], lr_enc((k & 1) ? r : -r));
    }
}

extern "C" int lr_feat_ratio(const float *F0, const float *F1, int dim, int m, const int32_t *i0, const int32_t *i1,
                             const int32_t *i2, float *out, void *stream)
{
    LR_REQUIRE(F0 && F1 && i1 && i2 && out && dim > 0 && m >= 0, LR_EINVAL, "lr_feat_ratio: bad argument");
    if (m == 0) return LR_OK;
    hipLaunchKernelGGL(ratio_kernel, dim3(lr_cdiv(m, 256)), dim3(256), 0, (hipStream_t)stream, F0, F1, dim, m,
                       (const int32_t *)nullptr, i0, i1, i2, out, (uint32_t *)nullptr, (const float *)nullptr, lr_zargs{ 0, nullptr });
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// ------------------------------------------------------------------ GPF (a7, BB_first=False)
// Step numbers follow matching.py:100-205.
//
// mm (six uint32 behind cell_fill): encoded extrema [0] -min ratio [1] max ratio [2] -min x [3] max x [4] -min y [5] max y
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

// normalised score (matching.py:118-134) and grid cell (matching.py:136-146), all in fp32 as torch does; the cell
// histogram is built per block in LDS and added to the global one with one atomic per (block, occupied cell)
__global__ void __launch_bounds__(256)
gpf_score_cell_kernel(int n0, int G, const uint32_t *__restrict__ mm, const uint8_t *__restrict__ is_bb,
                      const float *__restrict__ xyz0, float *__restrict__ ratio_inout, int32_t *__restrict__ cell,
                      int32_t *__restrict__ cell_count, lr_zargs z, const int32_t *__restrict__ m_dev = nullptr,
                      const int32_t *__restrict__ pidx = nullptr)
{
    __shared__ int s_cnt[LR_GPF_MAX_CELLS];
    if (z.descs) { n0 = z.descs[blockIdx.z].n0; xyz0 = z.descs[blockIdx.z].xyz0; }
    lr_z(mm, z, blockIdx.z); lr_z(is_bb, z, blockIdx.z); lr_z(ratio_inout, z, blockIdx.z); lr_z(cell, z, blockIdx.z); lr_z(cell_count, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(pidx, z, blockIdx.z);
    const int C = G * G;
    for (int k = threadIdx.x; k < C; k += 256) s_cnt[k] = 0;
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (m_dev) n0 = min(n0, *m_dev);
    if (i < n0) {
        const int pi = pidx ? pidx[i] : i;
        const float m = -lr_dec(mm[0]), M = lr_dec(mm[1]);
        const float x0 = -lr_dec(mm[2]), x1 = lr_dec(mm[3]), y0 = -lr_dec(mm[4]), y1 = lr_dec(mm[5]);
        float nfd = ((ratio_inout[i] - m) / (M - m));
        if (is_bb && is_bb[i]) nfd = nfd - 1.0f;          // BB_first=True has no best-buddy shift (matching.py:126)
        ratio_inout[i] = nfd;
        const float denx = (x1 - x0) + 1e-3f, deny = (y1 - y0) + 1e-3f;
        float qx = floorf((float)G * ((xyz0[3 * pi] - x0) / (denx)));
        float qy = floorf((float)G * ((xyz0[3 * pi + 1] - y0) / (deny)));
        int c = (int)qx * G + (int)qy;
        cell[i] = c;
        if (c >= 0 && c < C) atomicAdd(&s_cnt[c], 1);      // a pair with non-finite coordinates has no cell: it is never kept
    }
    __syncthreads();
    for (int k = threadIdx.x; k < C; k += 256) if (s_cnt[k]) atomicAdd(&cell_count[k], s_cnt[k]);
}

// numpy's pairwise_sum (umath loops, float64, contiguous) over v[i] = m[i] < h ? m[i] : h.  The leaf (n <= 128, i.e.
// every grid up to 11 x 11) is inlined into the bisection loop; larger grids take the recursive split through a call.
__device__ __forceinline__ double gpf_pairwise_leaf(const double *m, int n, double h)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += (m[i] < h) ? m[i] : h;
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (m[j] < h) ? m[j] : h;
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = m[i + j];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += (v[j] < h) ? v[j] : h;
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += (m[i] < h) ? m[i] : h;
    return res;
}
__device__ __noinline__ double gpf_pairwise_split(const double *m, int n, double h)
{
    if (n <= 128) return gpf_pairwise_leaf(m, n, h);
    int n2 = n / 2;
    n2 -= n2 % 8;
    return gpf_pairwise_split(m, n2, h) + gpf_pairwise_split(m + n2, n - n2, h);
}
__device__ __forceinline__ double gpf_pairwise_sum(const double *m, int n, double h)
{
    return n <= 128 ? gpf_pairwise_leaf(m, n, h) : gpf_pairwise_split(m, n, h);
}

// water-filling bisection in fp64 exactly as matching.py:154-179, then exclusive cell offsets.  One thread does the
// arithmetic (the sums run in cell order, which fixes their rounding), on a copy of the counts in LDS.
// total_fixed >= 0 selects the BB_first=True form: TOTAL = GPF_max_matches, and nothing is filtered (has_score = 0) when
// the mutual set is already that small (matching.py:109-113)
__global__ void __launch_bounds__(64)
gpf_waterfill_kernel(int G, double factor, const int32_t *__restrict__ counters,
                     const int32_t *__restrict__ cell_count, double *__restrict__ quota,
                     int32_t *__restrict__ cell_off, lr_zargs z, double total_fixed = -1.0,
                     const int32_t *__restrict__ m_dev = nullptr, int32_t *__restrict__ has_score = nullptr)
{
    __shared__ double s_m[LR_GPF_MAX_CELLS];
    lr_z(counters, z, blockIdx.z); lr_z(cell_count, z, blockIdx.z); lr_z(quota, z, blockIdx.z); lr_z(cell_off, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(has_score, z, blockIdx.z);
    __shared__ double s_hr;
    const int C = G * G;
    for (int c = threadIdx.x; c < C; c += 64) s_m[c] = (double)cell_count[c];
    __syncthreads();
    // the bisection: every lane runs the same loop.  per_quad.sum() of matching.py:169 is numpy's pairwise summation over
    // the flattened [G, G] float64 array; its order (eight running sums, fixed combination tree, blocks of at most 128
    // elements split recursively) is part of the result whenever the sum rounds, and it is reproduced exactly: for up to
    // 128 cells lanes 0..7 own the eight running sums and the tree is three shuffles; larger grids fall back to one lane.
    const int lane = threadIdx.x;
    auto total_at = [&](double h) -> double {
        if (C < 8 || C > 128) {
            double t = 0.0;
            if (lane == 0) t = gpf_pairwise_sum(s_m, C, h);
            return __shfl(t, 0);
        }
        const int full = C - (C % 8);
        double r = 0.0;
        if (lane < 8) {
            const double m0 = s_m[lane];
            r = (m0 < h) ? m0 : h;
            for (int i = 8 + lane; i < full; i += 8) { const double m = s_m[i]; r += (m < h) ? m : h; }
        }
        r = r + __shfl_down(r, 1);      // (r0+r1) (r2+r3) (r4+r5) (r6+r7) in lanes 0 2 4 6
        r = r + __shfl_down(r, 2);      // ((r0+r1)+(r2+r3)) in lane 0, ((r4+r5)+(r6+r7)) in lane 4
        r = r + __shfl_down(r, 4);
        if (lane == 0) for (int i = full; i < C; ++i) { const double m = s_m[i]; r += (m < h) ? m : h; }
        return __shfl(r, 0);
    };
    {
        const double TOTAL = total_fixed >= 0.0 ? total_fixed : factor * (double)counters[LR_CNT_NBB];
        bool keep_all = false;
        if (total_fixed >= 0.0) {
            keep_all = TOTAL >= (double)*m_dev;
            if (has_score && lane == 0) *has_score = keep_all ? 0 : 1;
        }
        double hr_v = __builtin_huge_val();      // +inf = keep every pair
        if (!keep_all) {
            double max_h = TOTAL, min_h = 0.0, cur = (max_h + min_h) / 2;
            while (fabs(max_h - min_h) > 2) {
                double t = total_at(cur);
                if (t == TOTAL) break;
                else if (t < TOTAL) min_h = cur;
                else if (t > TOTAL) max_h = cur;
                cur = (max_h + min_h) / 2;
            }
            hr_v = rint(cur);                          // np.round: half to even
        }
        if (lane == 0) s_hr = hr_v;
    }
    __syncthreads();
    // all lanes: quotas and the exclusive cell offsets (a wave-wide scan, 64 cells at a time)
    const double hr = s_hr;
    int carry = 0;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + threadIdx.x;
        const double m = c < C ? s_m[c] : 0.0;
        if (c < C) quota[c] = (m < hr) ? m : hr;
        int inc = (int)m;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if ((int)threadIdx.x >= d) inc += o; }
        if (c < C) cell_off[c] = carry + inc - (int)m;
        carry += __shfl(inc, 63);
    }
    if (threadIdx.x == 0) cell_off[C] = carry;
}

// bucket pair ids by cell (order inside a bucket is irrelevant: ranks below use (score, id)): ranks inside the block
// from LDS atomics, one global atomic per (block, occupied cell) reserves the block's range
__global__ void __launch_bounds__(256)
gpf_bucket_kernel(int n0, int G, const int32_t *__restrict__ cell, const int32_t *__restrict__ cell_off,
                  int32_t *__restrict__ cell_fill, int32_t *__restrict__ bucket, lr_zargs z, const int32_t *__restrict__ m_dev = nullptr)
{
    __shared__ int s_cnt[LR_GPF_MAX_CELLS];
    if (z.descs) n0 = z.descs[blockIdx.z].n0;
    lr_z(cell, z, blockIdx.z); lr_z(cell_off, z, blockIdx.z); lr_z(cell_fill, z, blockIdx.z); lr_z(bucket, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z);
    const int C = G * G;
    for (int k = threadIdx.x; k < C; k += 256) s_cnt[k] = 0;
    __syncthreads();
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (m_dev) n0 = min(n0, *m_dev);
    int c = -1, local = 0;
    if (i < n0) {
        c = cell[i];
        if (c >= 0 && c < C) local = atomicAdd(&s_cnt[c], 1);
        else c = -1;                                       // no cell (non-finite coordinates)
    }
    __syncthreads();
    for (int k = threadIdx.x; k < C; k += 256) if (s_cnt[k]) s_cnt[k] = atomicAdd(&cell_fill[k], s_cnt[k]);     // count -> base
    __syncthreads();
    if (c >= 0) bucket[cell_off[c] + s_cnt[c] + local] = i;
}

// keep[i] = all of the cell if quota == count, else rank of (score, i) inside the cell < quota
// (matching.py:184-195; ties in torch.argsort are unspecified, resolved towards the lower pair id)
__global__ void __launch_bounds__(256)
gpf_select_kernel(int n0, const int32_t *__restrict__ cell, const int32_t *__restrict__ cell_off,
                  const int32_t *__restrict__ cell_count, const double *__restrict__ quota,
                  const int32_t *__restrict__ bucket, const float *__restrict__ score, uint8_t *__restrict__ keep,
                  int C, lr_zargs z, const int32_t *__restrict__ m_dev = nullptr)
{
    if (z.descs) n0 = z.descs[blockIdx.z].n0;
    lr_z(cell, z, blockIdx.z); lr_z(cell_off, z, blockIdx.z); lr_z(cell_count, z, blockIdx.z); lr_z(quota, z, blockIdx.z); lr_z(bucket, z, blockIdx.z); lr_z(score, z, blockIdx.z); lr_z(keep, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z);
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (m_dev) n0 = min(n0, *m_dev);
    if (i >= n0) return;
    const int c = cell[i];
    if (c < 0 || c >= C) { keep[i] = 0; return; }          // non-finite coordinates: outside every cell
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
    // one memset clears the cell counters and the six min/max slots behind them (atomicMax on encoded values, 0 = identity)
    uint32_t *mm = reinterpret_cast<uint32_t *>(cell_fill + LR_GPF_MAX_CELLS);
    LR_TRY_HIP(lr_zero_scratch(ws, cell_count, sizeof(int32_t) * 2 * (LR_GPF_MAX_CELLS + 8), st));
    const dim3 gN(nb, 1, ws->zP), g1(1, 1, ws->zP);
    // ratio over all n0 NN pairs (corres_idx0 == arange), and the extrema of ratio / x / y
    hipLaunchKernelGGL(ratio_kernel, gN, dim3(256), 0, st, F0, F1, dim, n0, (const int32_t *)nullptr,
                       (const int32_t *)nullptr, idx1, idx2, ws->ratio, mm, xyz0, ws->z);
    hipLaunchKernelGGL(gpf_score_cell_kernel, gN, dim3(256), 0, st, n0, G, (const uint32_t *)mm, is_bb, xyz0, ws->ratio,
                       ws->cell, cell_count, ws->z);
    hipLaunchKernelGGL(gpf_waterfill_kernel, g1, dim3(64), 0, st, G, factor, ws->counters, cell_count, quota, cell_off, ws->z);
    hipLaunchKernelGGL(gpf_bucket_kernel, gN, dim3(256), 0, st, n0, G, ws->cell, cell_off, cell_fill, ws->cell_sorted, ws->z);
    hipLaunchKernelGGL(gpf_select_kernel, gN, dim3(256), 0, st, n0, ws->cell, cell_off, cell_count, quota,
                       ws->cell_sorted, ws->ratio, keep, G * G, ws->z);
    hipLaunchKernelGGL(count_flags_kernel, gN, dim3(256), 0, st, n0, (const int32_t *)nullptr, keep, ws->blk_cnt, ws->z);
    hipLaunchKernelGGL(compact_kernel, gN, dim3(256), 0, st, n0, keep, ws->blk_cnt, idx1, idx2, ws->ratio, o0, o1, o2, oscore,
                       n_out, (int32_t *)nullptr, xyz0, xyz1, corr8, ws->counters, ws->z);
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
    uint32_t *mm = reinterpret_cast<uint32_t *>(cell_fill + LR_GPF_MAX_CELLS);
    // (single-pair operator only: the reference's TEASER wrapper is not on the batched path)
    const lr_zargs z1 = { 0, nullptr };
    LR_HIP(hipMemsetAsync(cell_count, 0, sizeof(int32_t) * 2 * (LR_GPF_MAX_CELLS + 8), st));
    hipLaunchKernelGGL(ratio_kernel, dim3(nb), dim3(256), 0, st, F0, F1, dim, n0, mb_dev, b0, b1, b2, ws->ratio, mm, xyz0, z1);
    hipLaunchKernelGGL(gpf_score_cell_kernel, dim3(nb), dim3(256), 0, st, n0, G, (const uint32_t *)mm, (const uint8_t *)nullptr, xyz0, ws->ratio,
                       ws->cell, cell_count, z1, mb_dev, b0);
    hipLaunchKernelGGL(gpf_waterfill_kernel, dim3(1), dim3(64), 0, st, G, 0.0, ws->counters, cell_count, quota, cell_off, z1, max_matches,
                       mb_dev, has_score);
    hipLaunchKernelGGL(gpf_bucket_kernel, dim3(nb), dim3(256), 0, st, n0, G, ws->cell, cell_off, cell_fill, ws->cell_sorted, z1, mb_dev);
    hipLaunchKernelGGL(gpf_select_kernel, dim3(nb), dim3(256), 0, st, n0, ws->cell, cell_off, cell_count, quota,
                       ws->cell_sorted, ws->ratio, keep, G * G, z1, mb_dev);
    hipLaunchKernelGGL(count_flags_kernel, dim3(nb), dim3(256), 0, st, n0, mb_dev, keep, ws->blk_cnt, z1);
    hipLaunchKernelGGL(compact_kernel, dim3(nb), dim3(256), 0, st, n0, keep, ws->blk_cnt, b1, b2, ws->ratio, o0, o1, o2, oscore,
                       n_out, (int32_t *)nullptr, (const float *)nullptr, (const float *)nullptr, (float *)nullptr, ws->counters, z1, mb_dev, b0);
    LR_LAUNCH_CHECK();
    return LR_OK;
}
