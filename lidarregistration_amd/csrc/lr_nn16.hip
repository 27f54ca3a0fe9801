// Fast exact nearest / second-nearest neighbour search: f16 matrix-core filter + exact fp32 verification.
//
// Same contract as lr_nn.hip (reference Experiments/algorithms/matching.py:22-65; arithmetic of oracle/oracle.c):
// the RESULT is bit-identical to the fp32 fma-chain definition.  What changes is how the N0 x N1 candidates are
// pruned.  On gfx950 the fp32-input MFMA runs at the fp32 VALU rate and (measured, profiles/) does not overlap
// with VALU work, so the fp32 kernel pays 1024 + ~800 cycles per 32x32 tile.  Here:
//
//   pass A  f16 MFMA (32x32x16, the real matrix pipe) over every `stride`-th column tile; per query row the
//           2nd smallest (1st for top-1) approximate value u' = n1[j] - 2 dot16(i,j) of that subset: U_i
//   thresh  tau_i = U_i + 2 E_i (+ a sqrt-rounding band), E_i a rigorous bound on |d2_exact - (n0_i + u')|
//   pass B  f16 MFMA over ALL column tiles; a column is a candidate of row i iff u' <= tau_i
//           (2 VALU ops per element: one fma, one compare; candidates are ~1e-4 of the elements)
//   exact   per row, the fp32 fma-chain distance of its few candidates, ordered by (sqrt value, index) --
//           this is exactly torch.min's "first minimal value" order, so no separate tie-break path is needed;
//           rows whose candidate list overflowed (duplicate-heavy inputs) or could not be filled (non-finite
//           f16 conversions) go through the full exact row kernel nn_fix_kernel of lr_nn.hip.
//
// Why the candidate set is a superset of what the exact order needs: let S be the sampled columns and j1, j2 in S the
// two with the smallest u'.  Their exact distances are <= n0_i + U_i + E_i, hence so is the exact 2nd smallest x2 of
// the whole row.  Every j whose sqrt value ties with or beats the 2nd smallest has d2(j) <= x2 (1 + 2^-21), therefore
// u'(j) <= d2(j) - n0_i + E_i <= U_i + 2 E_i + 2^-21 (n0_i + U_i + E_i)  -- the threshold used below.
//
// Error bound (u = 2^-24; n0, n1 squared norms; all terms worst case):
//   exact chain vs real arithmetic      <= (2 g32 + 3u)(n0 + n1),  g32 = 32u/(1-32u)          ~  67 u (n0+n1)
//   f16 input rounding (RN, 2^-11 rel.) <= (2^-10 + 2^-22) 2 sqrt(n0 n1) <= (2^-10 + 2^-22)(n0+n1)
//   f16 underflow (|x| < 2^-14)         <= 2 * 2^-25 (|a|_1 + |b|_1) <= 3.4e-7 (1 + (n0+n1)/2)
//   MFMA fp32 accumulation of exact f16 products, norms, final fma             <= ~162 u (n0+n1) (generous)
//   => E_ij <= 9.91e-4 (n0+n1) + 3.4e-7 ;  used: 1.05e-3 (n0_i + max_j n1_j) + 4e-7.
#include "lr_internal.h"
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define LR_INF __builtin_huge_valf()
#define LR_IMAX 0x7fffffff

// ------------------------------------------------------------------ prep: norms, fp32 de-interleaved copy, f16 copy
// H[row] (64 B) = f16 of { k0..7, k16..23 | k8..15, k24..31 }: lane half h of an MFMA operand reads bytes [32h, 32h+32).
__global__ void __launch_bounds__(256)
nn16_prep_kernel(const float *__restrict__ F, int n, float *__restrict__ Fp, _Float16 *__restrict__ H,
                 float *__restrict__ nrm, uint32_t *__restrict__ max_norm_bits)
{
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.0f;
    if (row < n) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(F + (size_t)row * 32);
        float v[32];
#pragma unroll
        for (int q = 0; q < 8; ++q) { f32x4 t = src[q]; v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
#pragma unroll
        for (int k = 0; k < 32; ++k) acc = __builtin_fmaf(v[k], v[k], acc);
        nrm[row] = acc;
        f32x4 *dst = reinterpret_cast<f32x4 *>(Fp + (size_t)row * 32);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 e = { v[8 * q], v[8 * q + 2], v[8 * q + 4], v[8 * q + 6] };
            f32x4 o = { v[8 * q + 1], v[8 * q + 3], v[8 * q + 5], v[8 * q + 7] };
            dst[q] = e; dst[4 + q] = o;
        }
        f16x8 *hd = reinterpret_cast<f16x8 *>(H + (size_t)row * 32);
        const int kbase[4] = { 0, 16, 8, 24 };
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f16x8 hv;
#pragma unroll
            for (int j = 0; j < 8; ++j) hv[j] = (_Float16)v[kbase[c] + j];
            hd[c] = hv;
        }
    }
    // block max of the norms -> one atomic (norms are >= 0, so their bit patterns order like unsigned ints)
    float m = acc;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(max_norm_bits, __float_as_uint(m));
}

// ------------------------------------------------------------------ pass A / pass B
// One wave = 64 query rows (two 32-row MFMA blocks) x the column tiles of its strip.
//   MODE 0 (pass A): running (u1, u2) per accumulator element over the sampled tiles -> partials [strip][row]
//   MODE 1 (pass B): compare against tau[row]; passing (row, col) are appended to the row's candidate list
template <int MODE>
__global__ void __launch_bounds__(256)
nn16_pass_kernel(const _Float16 *__restrict__ Hq, int na, const _Float16 *__restrict__ Hc, const float *__restrict__ nC, int nb,
                 int tiles_per_strip, int tile_stride, int part_stride,
                 float *__restrict__ pu1, float *__restrict__ pu2,
                 const float *__restrict__ tau, int32_t *__restrict__ cand_cnt, int32_t *__restrict__ cand, int cap)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = (blockIdx.x * 4 + wave) * 64;
    if (row0 >= na) return;
    const int strip = blockIdx.y;
    const int ntiles = (nb + 31) >> 5;
    const int t_begin = strip * tiles_per_strip;
    const int t_end = min(ntiles, t_begin + tiles_per_strip);

    f16x8 a[2][2];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        const int row = min(row0 + 32 * rb + r, na - 1);
        const f16x8 *p = reinterpret_cast<const f16x8 *>(Hq + (size_t)row * 32 + 16 * h);
        a[rb][0] = p[0]; a[rb][1] = p[1];
    }

    float u1[2][16], u2[2][16];      // MODE 0 state; MODE 1: u1 holds tau per element row
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            if (MODE == 0) { u1[rb][g] = LR_INF; u2[rb][g] = LR_INF; }
            else {
                const int row = row0 + 32 * rb + (g & 3) + 8 * (g >> 2) + 4 * h;
                u1[rb][g] = row < na ? tau[row] : -LR_INF;
                u2[rb][g] = 0.0f;
            }
        }

    // software prefetch of the next candidate fragment
    f16x8 bn0, bn1;
    float nbn;
    {
        const int col = t_begin * 32 + r;
        const f16x8 *p = reinterpret_cast<const f16x8 *>(Hc + (size_t)min(col, nb - 1) * 32 + 16 * h);
        bn0 = p[0]; bn1 = p[1];
        const float nv = nC[min(col, nb - 1)];
        nbn = col < nb ? nv : LR_INF;
    }
    for (int t = t_begin; t < t_end; t += tile_stride) {
        const f16x8 b0 = bn0, b1 = bn1;
        const float nbv = nbn;
        const int col = t * 32 + r;
        if (t + tile_stride < t_end) {
            const int coln = col + 32 * tile_stride;
            const f16x8 *p = reinterpret_cast<const f16x8 *>(Hc + (size_t)min(coln, nb - 1) * 32 + 16 * h);
            bn0 = p[0]; bn1 = p[1];
            const float nv = nC[min(coln, nb - 1)];
            nbn = coln < nb ? nv : LR_INF;
        }
        f32x16 acc[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            acc[rb] = f32x16{ 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rb][0], b0, acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rb][1], b1, acc[rb], 0, 0, 0);
        }
        if (MODE == 0) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float u = __builtin_fmaf(-2.0f, acc[rb][g], nbv);
                    u2[rb][g] = __builtin_amdgcn_fmed3f(u1[rb][g], u2[rb][g], u);
                    u1[rb][g] = fminf(u1[rb][g], u);
                }
        } else {
            bool any = false;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const float u = __builtin_fmaf(-2.0f, acc[rb][g], nbv);
                    any |= (u <= u1[rb][g]);
                }
            if (__builtin_amdgcn_ballot_w64(any) != 0ull) {
                // rare: some lane holds a candidate in this tile
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const float u = __builtin_fmaf(-2.0f, acc[rb][g], nbv);
                        if (u <= u1[rb][g]) {
                            const int row = row0 + 32 * rb + (g & 3) + 8 * (g >> 2) + 4 * h;
                            const int slot = atomicAdd(&cand_cnt[row], 1);
                            if (slot < cap) cand[(size_t)row * cap + slot] = col;
                        }
                    }
            }
        }
    }

    if (MODE == 0) {
        // fold the 32 lanes (columns) of each half; values only
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
#pragma unroll
                for (int m = 1; m < 32; m <<= 1) {
                    const float c1 = __shfl_xor(u1[rb][g], m), c2 = __shfl_xor(u2[rb][g], m);
                    const float lo = fminf(u1[rb][g], c1), hi = fmaxf(u1[rb][g], c1);
                    u2[rb][g] = fminf(hi, fminf(u2[rb][g], c2));
                    u1[rb][g] = lo;
                }
            }
        if (r == 0) {
            const size_t base = (size_t)strip * part_stride;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = row0 + 32 * rb + (g & 3) + 8 * (g >> 2) + 4 * h;
                    if (row < na) { pu1[base + row] = u1[rb][g]; pu2[base + row] = u2[rb][g]; }
                }
        }
    }
}

// ------------------------------------------------------------------ thresholds
__global__ void __launch_bounds__(256)
nn16_thresh_kernel(int na, int nstrips, int part_stride, const float *__restrict__ pu1, const float *__restrict__ pu2,
                   const float *__restrict__ nQ, const uint32_t *__restrict__ max_norm_c_bits, int need,
                   float *__restrict__ tau, int32_t *__restrict__ cand_cnt)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= na) return;
    float a1 = pu1[row], a2 = pu2[row];
    for (int s = 1; s < nstrips; ++s) {
        const float c1 = pu1[(size_t)s * part_stride + row], c2 = pu2[(size_t)s * part_stride + row];
        const float lo = fminf(a1, c1), hi = fmaxf(a1, c1);
        a2 = fminf(hi, fminf(a2, c2));
        a1 = lo;
    }
    const float U = need >= 2 ? a2 : a1;
    const float scale = nQ[row] + __uint_as_float(*max_norm_c_bits);
    const float E = 1.05e-3f * scale + 4e-7f;
    // U + 2E + sqrt band 2^-21 (n0 + U + E) + rounding slop of this very expression
    float t = U + 2.0f * E + 3e-6f * scale + 1e-6f * fabsf(U);
    tau[row] = t;            // +inf when fewer than `need` columns were sampled: every column becomes a candidate
    cand_cnt[row] = 0;
}

// ------------------------------------------------------------------ exact verification of the candidates
// One thread per row.  Rows with an overflowing or too-short candidate list are queued for the full exact row kernel.
__global__ void __launch_bounds__(128)
nn16_exact_kernel(const float *__restrict__ Fq, const float *__restrict__ nQ, int na,
                  const float *__restrict__ Fc, const float *__restrict__ nC, int nb,
                  const int32_t *__restrict__ cand_cnt, const int32_t *__restrict__ cand, int cap, int need,
                  int32_t *__restrict__ idx1, int32_t *__restrict__ idx2, float *__restrict__ s1o, float *__restrict__ s2o,
                  int32_t *__restrict__ fix_list, int32_t *__restrict__ counters)
{
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= na) return;
    const int cnt = cand_cnt[row];
    const int want = min(need, nb);
    if (cnt > cap || cnt < want) {
        const int slot = atomicAdd(&counters[LR_CNT_FIX], 1);
        fix_list[slot] = row;
        return;
    }
    float a[32];
    const f32x4 *pa = reinterpret_cast<const f32x4 *>(Fq + (size_t)row * 32);
#pragma unroll
    for (int q = 0; q < 8; ++q) { f32x4 t = pa[q]; a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w; }
    const float nq = nQ[row];
    float b1 = LR_INF, b2 = LR_INF;
    int i1 = LR_IMAX, i2 = LR_IMAX;
    for (int c = 0; c < cnt; ++c) {
        const int j = cand[(size_t)row * cap + c];
        const f32x4 *pb = reinterpret_cast<const f32x4 *>(Fc + (size_t)j * 32);
        float acc = 0.0f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 t = pb[q];
            acc = __builtin_fmaf(a[4 * q], t.x, acc);
            acc = __builtin_fmaf(a[4 * q + 1], t.y, acc);
            acc = __builtin_fmaf(a[4 * q + 2], t.z, acc);
            acc = __builtin_fmaf(a[4 * q + 3], t.w, acc);
        }
        const float tt = nq + nC[j];
        const float d2 = __builtin_fmaf(-2.0f, acc, tt);
        const float s = __builtin_sqrtf(fmaxf(d2, 1e-30f));
        // candidates arrive in arbitrary order: order by (s, j)
        const bool lt1 = s < b1 || (s == b1 && j < i1);
        const bool lt2 = s < b2 || (s == b2 && j < i2);
        if (lt1) { b2 = b1; i2 = i1; b1 = s; i1 = j; }
        else if (lt2) { b2 = s; i2 = j; }
    }
    idx1[row] = i1;
    if (idx2) idx2[row] = i2;
    if (s1o) s1o[row] = b1;
    if (s2o) s2o[row] = b2;
}

// ------------------------------------------------------------------ host side
int lr_nn16_prep(lr_workspace *ws, const float *F, int n, float *Fp, _Float16 *H, float *nrm, uint32_t *max_bits, hipStream_t st)
{
    (void)ws;
    LR_HIP(hipMemsetAsync(max_bits, 0, sizeof(uint32_t), st));
    hipLaunchKernelGGL(nn16_prep_kernel, dim3(lr_cdiv(n, 256)), dim3(256), 0, st, F, n, Fp, H, nrm, max_bits);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

int lr_nn16_run(lr_workspace *ws, const float *Fq, const float *Fpq, const _Float16 *Hq, const float *nQ, int na,
                const float *Fc, const float *Fpc, const _Float16 *Hc, const float *nC, const uint32_t *max_c_bits, int nb,
                int need, int32_t *idx1, int32_t *idx2, float *s1, float *s2, hipStream_t st)
{
    const int ntiles = lr_cdiv(nb, 32);
    const int row_blocks = lr_cdiv(na, 256);
    // pass A samples every `stride`-th tile (any subset gives a valid, if looser, threshold)
    int stride = ntiles / 16;
    if (stride > LR_NN16_STRIDE) stride = LR_NN16_STRIDE;
    if (stride < 1) stride = 1;
    // strips: enough blocks to fill 256 CUs a few times over, at least 8 tiles per strip
    int strips = lr_cdiv(1024, row_blocks);
    int smax = ntiles / (8 * stride);
    if (strips > smax) strips = smax;
    if (strips > LR_NN_MAX_STRIPS) strips = LR_NN_MAX_STRIPS;
    if (strips < 1) strips = 1;
    int tps = lr_cdiv(lr_cdiv(ntiles, strips), stride) * stride;     // tiles per strip, multiple of the stride
    LR_HIP(hipMemsetAsync(ws->counters + LR_CNT_FIX, 0, sizeof(int32_t), st));
    dim3 grid(row_blocks, strips);
    hipLaunchKernelGGL(nn16_pass_kernel<0>, grid, dim3(256), 0, st, Hq, na, Hc, nC, nb, tps, stride, ws->max_n, ws->pb1, ws->pb2,
                       (const float *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, 0);
    hipLaunchKernelGGL(nn16_thresh_kernel, dim3(lr_cdiv(na, 256)), dim3(256), 0, st, na, strips, ws->max_n, ws->pb1, ws->pb2, nQ,
                       max_c_bits, need, ws->tau, ws->cand_cnt);
    if (ws->timing && !ws->ev_pending) { LR_HIP(hipEventRecord(ws->ev[0], st)); }
    hipLaunchKernelGGL(nn16_pass_kernel<1>, grid, dim3(256), 0, st, Hq, na, Hc, nC, nb, tps, 1, ws->max_n, (float *)nullptr,
                       (float *)nullptr, ws->tau, ws->cand_cnt, ws->cand, LR_NN16_CAP);
    if (ws->timing && !ws->ev_pending) { LR_HIP(hipEventRecord(ws->ev[1], st)); ws->ev_pending = 1; }
    hipLaunchKernelGGL(nn16_exact_kernel, dim3(lr_cdiv(na, 128)), dim3(128), 0, st, Fq, nQ, na, Fc, nC, nb, ws->cand_cnt, ws->cand,
                       LR_NN16_CAP, need, idx1, idx2, s1, s2, ws->fix_list, ws->counters);
    LR_LAUNCH_CHECK();
    return lr_nn_fix_rows(ws, Fpq, nQ, Fpc, nC, nb, idx1, idx2, s1, s2, st);
}
