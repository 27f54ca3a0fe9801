// Feature-space nearest / second-nearest neighbour for N x 32 FCGF descriptors on gfx950.
//
// Replaces find_nn / knn_dist (reference Experiments/algorithms/matching.py:22-65): the reference
// materialises 250 x N1 distance chunks through HBM (einsum + clamp + sqrt + min + scatter + min);
// here the N0 x N1 matrix only ever exists as 32x32 MFMA accumulator tiles.
//
// Arithmetic contract (identical to oracle/oracle.c):
//   dot  = fp32 fma chain over k = 0..31            (v_mfma_f32_32x32x2_f32 is exactly that chain)
//   d2   = fma(-2, dot, n0[i] + n1[j]);   s = sqrt(max(d2, 1e-30))
//   order: ascending (s, j) -- first minimal value wins, like torch.min(dim=1).
//
// sqrt is monotone, so the hot loop compares d2 and keeps the three smallest values (two with their
// indices).  Only when sqrt rounds the 2nd and 3rd smallest d2 to the same float can a candidate that
// was never indexed outrank one that was; those rows are re-done exactly by nn_fix_kernel (rare: ~1e-5
// of rows on random descriptors).
#include "lr_internal.h"
#include <math.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define LR_INF __builtin_huge_valf()
#define LR_IMAX 0x7fffffff

// ------------------------------------------------------------------ prep: norms + k de-interleave
// Fp[row] = { F[row][0], F[row][2], ..., F[row][30] | F[row][1], F[row][3], ..., F[row][31] } so that lane
// half h of an MFMA operand loads one contiguous 64-byte run and MFMA m consumes k = (2m, 2m+1).
__global__ void __launch_bounds__(256) nn_prep_kernel(const float *__restrict__ F, int n, float *__restrict__ Fp,
                                                      float *__restrict__ nrm)
{
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    const f32x4 *src = reinterpret_cast<const f32x4 *>(F + (size_t)row * 32);
    float v[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        f32x4 t = src[q];
        v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
    }
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 32; ++k) acc = __builtin_fmaf(v[k], v[k], acc);
    nrm[row] = acc;
    f32x4 *dst = reinterpret_cast<f32x4 *>(Fp + (size_t)row * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 e = { v[8 * q], v[8 * q + 2], v[8 * q + 4], v[8 * q + 6] };
        f32x4 o = { v[8 * q + 1], v[8 * q + 3], v[8 * q + 5], v[8 * q + 7] };
        dst[q] = e;
        dst[4 + q] = o;
    }
}

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ bool lex_lt(float a, int ia, float b, int ib) { return a < b || (a == b && ia < ib); }

// merge two (b1,i1,b2,i2,b3) summaries of disjoint candidate sets under (value, index) order
__device__ __forceinline__ void top3_merge(float &b1, int &i1, float &b2, int &i2, float &b3,
                                           float c1, int j1, float c2, int j2, float c3)
{
    bool cf = lex_lt(c1, j1, b1, i1);
    float x1 = cf ? c1 : b1, x2 = cf ? c2 : b2, x3 = cf ? c3 : b3;
    int xi1 = cf ? j1 : i1, xi2 = cf ? j2 : i2;
    float y1 = cf ? b1 : c1, y2 = cf ? b2 : c2;
    int yi1 = cf ? i1 : j1;
    bool s = lex_lt(y1, yi1, x2, xi2);
    b1 = x1; i1 = xi1;
    b2 = s ? y1 : x2; i2 = s ? yi1 : xi2;
    b3 = s ? fminf(x2, y2) : fminf(x3, y1);
}

// ------------------------------------------------------------------ the distance kernel
// One wave = 32 query rows (MFMA rows) x one column strip; 4 independent waves per block.
// Accumulator layout of v_mfma_f32_32x32x2_f32: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// Each lane therefore keeps, for 16 rows, the running top-3 over the columns congruent to its lane.
__global__ void __launch_bounds__(256, 2)
nn_strip_kernel(const float *__restrict__ Ap, const float *__restrict__ nA, int na,
                const float *__restrict__ Bp, const float *__restrict__ nB, int nb,
                int cols_per_strip, int part_stride,
                float *__restrict__ pb1, float *__restrict__ pb2, float *__restrict__ pb3,
                int32_t *__restrict__ pi1, int32_t *__restrict__ pi2)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = lane & 31, h = lane >> 5;
    const int row0 = (blockIdx.x * 4 + wave) * 32;
    if (row0 >= na) return;
    const int strip = blockIdx.y;
    const int c_begin = strip * cols_per_strip;
    const int c_end = min(nb, c_begin + cols_per_strip);

    // query fragment: 16 k-values of row (row0 + r), half h
    float a[16];
    {
        int row = min(row0 + r, na - 1);
        const f32x4 *p = reinterpret_cast<const f32x4 *>(Ap + (size_t)row * 32 + 16 * h);
#pragma unroll
        for (int q = 0; q < 4; ++q) { f32x4 t = p[q]; a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w; }
    }
    float nq[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) nq[g] = nA[min(row0 + (g & 3) + 8 * (g >> 2) + 4 * h, na - 1)];

    float b1[16], b2[16], b3[16];
    int i1[16], i2[16];
#pragma unroll
    for (int g = 0; g < 16; ++g) { b1[g] = LR_INF; b2[g] = LR_INF; b3[g] = LR_INF; i1[g] = LR_IMAX; i2[g] = LR_IMAX; }

    // Software pipeline over 32-column tiles: the 16 MFMAs of tile t+1 are interleaved, one per accumulator
    // register, with the top-3 update of tile t (11 VALU ops each), so the matrix pipe and the VALU run together
    // even with one wave per SIMD; the candidate fragment of tile t+2 is in flight meanwhile.
#define LR_LOAD_B(dst, ndst, c0_)                                                                             \
    {                                                                                                         \
        int col_ = (c0_) + r;                                                                                 \
        const f32x4 *p_ = reinterpret_cast<const f32x4 *>(Bp + (size_t)min(col_, nb - 1) * 32 + 16 * h);      \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                       \
            f32x4 t_ = p_[q];                                                                                 \
            dst[4 * q] = t_.x; dst[4 * q + 1] = t_.y; dst[4 * q + 2] = t_.z; dst[4 * q + 3] = t_.w;           \
        }                                                                                                     \
        float nv_ = nB[min(col_, nb - 1)];                                                                    \
        ndst = col_ < c_end ? nv_ : LR_INF;                                                                   \
    }
#define LR_SCHED_HINT __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, 11, 0);
#define LR_UPDATE(g, accv, colv, nbv)                                                                         \
    {                                                                                                         \
        float t_ = nq[g] + (nbv);                                                                             \
        float d2_ = fmaxf(__builtin_fmaf(-2.0f, (accv), t_), 1e-30f);                                         \
        float ob1_ = b1[g], ob2_ = b2[g];                                                                     \
        int oi1_ = i1[g], oi2_ = i2[g];                                                                       \
        bool lt1_ = d2_ < ob1_, lt2_ = d2_ < ob2_;                                                            \
        b3[g] = __builtin_amdgcn_fmed3f(ob2_, b3[g], d2_);                                                    \
        b2[g] = __builtin_amdgcn_fmed3f(ob1_, ob2_, d2_);                                                     \
        b1[g] = fminf(ob1_, d2_);                                                                             \
        int k2_ = lt2_ ? (colv) : oi2_;                                                                       \
        i2[g] = lt1_ ? oi1_ : k2_;                                                                            \
        i1[g] = lt1_ ? (colv) : oi1_;                                                                         \
    }
    // one pipeline stage: multiply tile (bt) into accN while folding accC (tile at column colC) into the state
#define LR_STAGE(accN, bt, accC, colC, nbC)                                                                   \
    {                                                                                                         \
        accN = f32x16{ 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };                                      \
        _Pragma("unroll") for (int m = 0; m < 16; ++m) {                                                      \
            accN = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bt[m], accN, 0, 0, 0);                          \
            LR_UPDATE(m, accC[m], colC, nbC)                                                                  \
            LR_SCHED_HINT                                                                                     \
        }                                                                                                     \
    }

    const int ntiles = (c_end - c_begin + 31) / 32;
    float bA[16], bB[16];
    float nbA, nbB;
    f32x16 accA, accB;
    LR_LOAD_B(bA, nbA, c_begin)
    LR_LOAD_B(bB, nbB, c_begin + 32)
    accA = f32x16{ 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
#pragma unroll
    for (int m = 0; m < 16; ++m) accA = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bA[m], accA, 0, 0, 0);
    // invariant at loop top (t even): accA holds tile t (columns colA, norms nbA); bB/nbB hold tile t+1
    int t = 0;
    for (; t + 2 < ntiles; t += 2) {
        const int cA = c_begin + 32 * t;
        const float nA_ = nbA;
        LR_LOAD_B(bA, nbA, cA + 64)                       // tile t+2
        LR_STAGE(accB, bB, accA, cA + r, nA_)             // multiply t+1, fold t
        const float nB_ = nbB;
        LR_LOAD_B(bB, nbB, cA + 96)                       // tile t+3 (clamped loads; masked by +inf norm when past the strip)
        LR_STAGE(accA, bA, accB, cA + 32 + r, nB_)        // multiply t+2, fold t+1
    }
    // tail: tiles t (in accA) and possibly t+1 (fragment in bB)
    {
        const int cA = c_begin + 32 * t;
        if (t + 1 < ntiles) {
            LR_STAGE(accB, bB, accA, cA + r, nbA)
#pragma unroll
            for (int g = 0; g < 16; ++g) LR_UPDATE(g, accB[g], cA + 32 + r, nbB)
        } else {
#pragma unroll
            for (int g = 0; g < 16; ++g) LR_UPDATE(g, accA[g], cA + r, nbA)
        }
    }
#undef LR_STAGE
#undef LR_SCHED_HINT
#undef LR_UPDATE
#undef LR_LOAD_B

    // fold the 32 lanes (columns) of each half together; xor masks < 32 stay inside the half
#pragma unroll
    for (int g = 0; g < 16; ++g) {
#pragma unroll
        for (int m = 1; m < 32; m <<= 1) {
            float c1 = __shfl_xor(b1[g], m), c2 = __shfl_xor(b2[g], m), c3 = __shfl_xor(b3[g], m);
            int j1 = __shfl_xor(i1[g], m), j2 = __shfl_xor(i2[g], m);
            top3_merge(b1[g], i1[g], b2[g], i2[g], b3[g], c1, j1, c2, j2, c3);
        }
    }
    if (r == 0) {
        size_t base = (size_t)strip * part_stride;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            int row = row0 + (g & 3) + 8 * (g >> 2) + 4 * h;
            if (row < na) {
                pb1[base + row] = b1[g]; pb2[base + row] = b2[g]; pb3[base + row] = b3[g];
                pi1[base + row] = i1[g]; pi2[base + row] = i2[g];
            }
        }
    }
}

// ------------------------------------------------------------------ finalize: merge strips, apply sqrt order
__global__ void __launch_bounds__(256)
nn_finalize_kernel(int na, int nstrips, int part_stride,
                   const float *__restrict__ pb1, const float *__restrict__ pb2, const float *__restrict__ pb3,
                   const int32_t *__restrict__ pi1, const int32_t *__restrict__ pi2,
                   int32_t *__restrict__ idx1, int32_t *__restrict__ idx2, float *__restrict__ s1o, float *__restrict__ s2o,
                   int32_t *__restrict__ fix_list, int32_t *__restrict__ counters)
{
    int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= na) return;
    float b1 = pb1[row], b2 = pb2[row], b3 = pb3[row];
    int i1 = pi1[row], i2 = pi2[row];
    for (int s = 1; s < nstrips; ++s) {
        size_t o = (size_t)s * part_stride + row;
        top3_merge(b1, i1, b2, i2, b3, pb1[o], pi1[o], pb2[o], pi2[o], pb3[o]);
    }
    float s1 = __builtin_sqrtf(b1), s2 = __builtin_sqrtf(b2), s3 = __builtin_sqrtf(b3);
    if (s2 == s3) {
        // a third candidate ties with the second after sqrt rounding: resolve exactly
        int slot = atomicAdd(&counters[LR_CNT_FIX], 1);
        fix_list[slot] = row;
        return;
    }
    if (s1 == s2 && i2 < i1) { int t = i1; i1 = i2; i2 = t; }
    idx1[row] = i1;
    if (idx2) idx2[row] = i2;
    if (s1o) s1o[row] = s1;
    if (s2o) s2o[row] = s2;
}

// ------------------------------------------------------------------ exact path for flagged rows
// One wave per flagged row; every candidate's s = sqrt(max(d2,1e-30)) is formed and ordered by (s, j).
// PERM: the operands are the de-interleaved copies of nn_prep_kernel (k = 2m at m, 2m+1 at 16+m); otherwise plain rows.
template <bool PERM>
__global__ void __launch_bounds__(64)
nn_fix_kernel(const float *__restrict__ Ap, const float *__restrict__ nA,
              const float *__restrict__ Bp, const float *__restrict__ nB, int nb,
              const int32_t *__restrict__ fix_list, int32_t *__restrict__ counters,
              int32_t *__restrict__ idx1, int32_t *__restrict__ idx2, float *__restrict__ s1o, float *__restrict__ s2o)
{
    const int nfix = counters[LR_CNT_FIX];
    const int lane = threadIdx.x;
    for (int f = blockIdx.x; f < nfix; f += gridDim.x) {
        const int row = fix_list[f];
        float a[32];
        const f32x4 *pa = reinterpret_cast<const f32x4 *>(Ap + (size_t)row * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) { f32x4 t = pa[q]; a[4 * q] = t.x; a[4 * q + 1] = t.y; a[4 * q + 2] = t.z; a[4 * q + 3] = t.w; }
        const float nq = nA[row];
        float b1 = LR_INF, b2 = LR_INF;
        int i1 = LR_IMAX, i2 = LR_IMAX;
        for (int j = lane; j < nb; j += 64) {
            const f32x4 *pb = reinterpret_cast<const f32x4 *>(Bp + (size_t)j * 32);
            float b[32];
#pragma unroll
            for (int q = 0; q < 8; ++q) { f32x4 t = pb[q]; b[4 * q] = t.x; b[4 * q + 1] = t.y; b[4 * q + 2] = t.z; b[4 * q + 3] = t.w; }
            float acc = 0.0f;
            if (PERM) {
#pragma unroll
                for (int m = 0; m < 16; ++m) {      // natural k order: (2m) sits at m, (2m+1) at 16+m
                    acc = __builtin_fmaf(a[m], b[m], acc);
                    acc = __builtin_fmaf(a[16 + m], b[16 + m], acc);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 32; ++k) acc = __builtin_fmaf(a[k], b[k], acc);
            }
            float t = nq + nB[j];
            float d2 = __builtin_fmaf(-2.0f, acc, t);
            float s = __builtin_sqrtf(fmaxf(d2, 1e-30f));
            if (s < b1) { b2 = b1; i2 = i1; b1 = s; i1 = j; }
            else if (s < b2) { b2 = s; i2 = j; }
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            float c1 = __shfl_xor(b1, m), c2 = __shfl_xor(b2, m);
            int j1 = __shfl_xor(i1, m), j2 = __shfl_xor(i2, m);
            float dummy = LR_INF;
            top3_merge(b1, i1, b2, i2, dummy, c1, j1, c2, j2, LR_INF);
        }
        if (lane == 0) {
            idx1[row] = i1;
            if (idx2) idx2[row] = i2;
            if (s1o) s1o[row] = b1;
            if (s2o) s2o[row] = b2;
        }
    }
    if (blockIdx.x == 0 && lane == 0) atomicAdd(&counters[LR_CNT_FIX_TOTAL], nfix);
}

// ------------------------------------------------------------------ host side
int lr_nn_prep(lr_workspace *ws, const float *F, int n, float *Fp, float *nrm, hipStream_t st)
{
    (void)ws;
    hipLaunchKernelGGL(nn_prep_kernel, dim3(lr_cdiv(n, 256)), dim3(256), 0, st, F, n, Fp, nrm);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

int lr_nn_fix_rows(lr_workspace *ws, bool permuted, const float *Fa, const float *nrma, const float *Fb, const float *nrmb, int nb,
                   int32_t *idx1, int32_t *idx2, float *s1, float *s2, hipStream_t st)
{
    if (permuted)
        hipLaunchKernelGGL(nn_fix_kernel<true>, dim3(256), dim3(64), 0, st, Fa, nrma, Fb, nrmb, nb, ws->fix_list, ws->counters,
                           idx1, idx2, s1, s2);
    else
        hipLaunchKernelGGL(nn_fix_kernel<false>, dim3(256), dim3(64), 0, st, Fa, nrma, Fb, nrmb, nb, ws->fix_list, ws->counters,
                           idx1, idx2, s1, s2);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

static int pick_strips(int na, int nb)
{
    // aim for >= 2 waves per SIMD (2048 waves on 256 CUs) without making strips shorter than 1024 columns
    int row_waves = lr_cdiv(na, 32);
    int s = lr_cdiv(2048, row_waves > 0 ? row_waves : 1);
    int smax = nb / 1024;
    if (s > smax) s = smax;
    if (s > LR_NN_MAX_STRIPS) s = LR_NN_MAX_STRIPS;
    if (s < 1) s = 1;
    return s;
}

int lr_nn_run(lr_workspace *ws, const float *Fa, const float *nrma, int na, const float *Fb, const float *nrmb, int nb,
              int32_t *idx1, int32_t *idx2, float *s1, float *s2, hipStream_t st)
{
    const int nstrips = pick_strips(na, nb);
    int cols = lr_cdiv(lr_cdiv(nb, nstrips), 32) * 32;
    LR_HIP(hipMemsetAsync(ws->counters + LR_CNT_FIX, 0, sizeof(int32_t), st));
    dim3 grid(lr_cdiv(na, 128), nstrips);
    if (ws->timing && !ws->ev_pending) { LR_HIP(hipEventRecord(ws->ev[0], st)); }
    hipLaunchKernelGGL(nn_strip_kernel, grid, dim3(256), 0, st, Fa, nrma, na, Fb, nrmb, nb, cols, ws->max_n,
                       ws->pb1, ws->pb2, ws->pb3, ws->pi1, ws->pi2);
    LR_LAUNCH_CHECK();
    if (ws->timing && !ws->ev_pending) { LR_HIP(hipEventRecord(ws->ev[1], st)); ws->ev_pending = 1; }
    hipLaunchKernelGGL(nn_finalize_kernel, dim3(lr_cdiv(na, 256)), dim3(256), 0, st, na, nstrips, ws->max_n,
                       ws->pb1, ws->pb2, ws->pb3, ws->pi1, ws->pi2, idx1, idx2, s1, s2, ws->fix_list, ws->counters);
    LR_LAUNCH_CHECK();
    LR_LAUNCH_CHECK();
    return lr_nn_fix_rows(ws, true, Fa, nrma, Fb, nrmb, nb, idx1, idx2, s1, s2, st);
}
