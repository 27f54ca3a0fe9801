// Voxel de-duplication of a raw LiDAR cloud on gfx950 -- the step between the reference's cloud cache and the hot path.
//
// Replaces ME.utils.sparse_quantize(xyz / voxel_size, return_index=True) as the reference's loaders call it
// (Experiments/dataloader/generic_balanced_loader.py:62-63, voxel_size 0.3; MinkowskiEngine 0.5.4, Requirements/
// conda_GC_full.yml:106 -- not vendored: parity unpinned, restated in oracle/oracle.py::sparse_quantize): coordinates are
// floored to integer cells and one point per occupied cell is kept -- the FIRST one in input order -- with the kept indices
// returned in ascending order, so that xyz[sel] keeps the scan order of the cloud.
//
// Structure (HBM-bound integer work, three small launches): open-addressing hash table over the packed cell key
// (3 x 21 bits), 2..4 slots per point; every point claims its cell's slot with a 64-bit compare-and-swap and lowers the
// slot's point index with atomicMin (so the result does not depend on the order the threads run in); a point is kept iff it
// is its cell's minimum; ordered compaction = per-block counts + prefix + scatter.
#include "lr_internal.h"
#include <math.h>

#define LR_VX_EMPTY 0xffffffffffffffffull
#define LR_VX_BIAS (1 << 20)          // cells are stored biased: |cell| < 2^20 per axis

__device__ __forceinline__ unsigned long long vx_key(const double *__restrict__ c, int i, bool &ok)
{
    const double fx = floor(c[3 * (size_t)i]), fy = floor(c[3 * (size_t)i + 1]), fz = floor(c[3 * (size_t)i + 2]);
    ok = fabs(fx) < (double)LR_VX_BIAS && fabs(fy) < (double)LR_VX_BIAS && fabs(fz) < (double)LR_VX_BIAS;     // false for NaN / inf too
    const unsigned long long x = (unsigned long long)((long long)fx + LR_VX_BIAS), y = (unsigned long long)((long long)fy + LR_VX_BIAS),
                             z = (unsigned long long)((long long)fz + LR_VX_BIAS);
    return ok ? (x << 42) | (y << 21) | z : 0ull;
}

__device__ __forceinline__ unsigned vx_hash(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned)k;
}

__global__ void __launch_bounds__(256)
voxel_insert_kernel(const double *__restrict__ coords, int n, unsigned long long *__restrict__ keys, int32_t *__restrict__ first,
                    unsigned cap_mask, int32_t *__restrict__ slot_of)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool ok;
    const unsigned long long k = vx_key(coords, i, ok);
    if (!ok) { slot_of[i] = -1; return; }           // outside the representable grid / non-finite: dropped
    unsigned s = vx_hash(k) & cap_mask;
    for (;;) {
        const unsigned long long prev = atomicCAS(&keys[s], LR_VX_EMPTY, k);
        if (prev == LR_VX_EMPTY || prev == k) break;
        s = (s + 1) & cap_mask;
    }
    atomicMin(&first[s], i);
    slot_of[i] = (int32_t)s;
}

__global__ void __launch_bounds__(256)
voxel_flag_kernel(int n, const int32_t *__restrict__ first, const int32_t *__restrict__ slot_of, uint8_t *__restrict__ keep,
                  int32_t *__restrict__ blk_cnt)
{
    __shared__ int s_w[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool k = false;
    if (i < n) { const int s = slot_of[i]; k = s >= 0 && first[s] == i; keep[i] = k ? 1 : 0; }
    const unsigned long long bal = __ballot(k);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ void __launch_bounds__(256)
voxel_compact_kernel(const double *__restrict__ coords, int n, const uint8_t *__restrict__ keep, const int32_t *__restrict__ blk_cnt,
                     int32_t *__restrict__ sel, int32_t *__restrict__ cells, int32_t *__restrict__ n_sel)
{
    __shared__ int s_w[4], s_p[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int c = 0;
    for (int b = tid; b < (int)blockIdx.x; b += 256) c += blk_cnt[b];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m);
    const int i = blockIdx.x * 256 + tid;
    const bool k = i < n && keep[i] != 0;
    const unsigned long long bal = __ballot(k);
    if (lane == 0) { s_p[wave] = c; s_w[wave] = __popcll(bal); }
    __syncthreads();
    const int prefix = s_p[0] + s_p[1] + s_p[2] + s_p[3];
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_w[w];
    const int slot = prefix + woff + __popcll(bal & ((1ull << lane) - 1ull));
    if (k) {
        sel[slot] = i;
        if (cells) {
#pragma unroll
            for (int a = 0; a < 3; ++a) cells[3 * (size_t)slot + a] = (int32_t)floor(coords[3 * (size_t)i + a]);
        }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) *n_sel = prefix + s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

static size_t vx_capacity(int n)
{
    size_t c = 1024;
    while (c < 2 * (size_t)(n > 0 ? n : 1)) c <<= 1;
    return c;
}

extern "C" size_t lr_voxel_dedup_scratch_bytes(int n)
{
    const size_t cap = vx_capacity(n), nn = (size_t)(n > 0 ? n : 1);
    return cap * 8 + cap * 4 + nn * 4 + ((nn + 255) & ~size_t(255)) + (nn / 256 + 2) * 4 + 1024;
}

extern "C" int lr_voxel_dedup(const double *coords, int n, int32_t *sel, int32_t *n_sel, int32_t *cells, void *scratch,
                              size_t scratch_bytes, void *stream)
{
    LR_REQUIRE(n >= 0, LR_EINVAL, "lr_voxel_dedup: negative point count");
    LR_REQUIRE(n_sel && (n == 0 || (coords && sel && scratch)), LR_EINVAL, "lr_voxel_dedup: null pointer");
    LR_REQUIRE(scratch_bytes >= lr_voxel_dedup_scratch_bytes(n), LR_ESIZE, "lr_voxel_dedup: scratch too small (lr_voxel_dedup_scratch_bytes)");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { LR_HIP(hipMemsetAsync(n_sel, 0, sizeof(int32_t), st)); return LR_OK; }
    const size_t cap = vx_capacity(n);
    char *p = reinterpret_cast<char *>(scratch);
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(p); p += cap * 8;
    int32_t *first = reinterpret_cast<int32_t *>(p); p += cap * 4;
    int32_t *slot_of = reinterpret_cast<int32_t *>(p); p += (size_t)n * 4;
    uint8_t *keep = reinterpret_cast<uint8_t *>(p); p += ((size_t)n + 255) & ~size_t(255);
    int32_t *blk_cnt = reinterpret_cast<int32_t *>(p);
    LR_HIP(hipMemsetAsync(keys, 0xff, cap * 8, st));
    LR_HIP(hipMemsetAsync(first, 0x7f, cap * 4, st));
    const int nb = lr_cdiv(n, 256);
    hipLaunchKernelGGL(voxel_insert_kernel, dim3(nb), dim3(256), 0, st, coords, n, keys, first, (unsigned)(cap - 1), slot_of);
    hipLaunchKernelGGL(voxel_flag_kernel, dim3(nb), dim3(256), 0, st, n, (const int32_t *)first, (const int32_t *)slot_of, keep, blk_cnt);
    hipLaunchKernelGGL(voxel_compact_kernel, dim3(nb), dim3(256), 0, st, coords, n, (const uint8_t *)keep, (const int32_t *)blk_cnt, sel, cells, n_sel);
    LR_LAUNCH_CHECK();
    return LR_OK;
}
