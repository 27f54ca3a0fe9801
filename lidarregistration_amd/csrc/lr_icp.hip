// Point-to-point ICP refinement on gfx950 -- the stage right after the RANSAC path in the reference's harness.
//
// Replaces o3d.pipelines.registration.registration_icp(src, tgt, 0.6, init, TransformationEstimationPointToPoint())
// (reference Experiments/test.py:183-189; FCGF_FAST/net/RANSAC.py:105-112).  Open3D 0.13.0 semantics (third-party, not
// vendored -> parity unpinned, restated in oracle/oracle.c::orc_icp): correspondences = nearest target point of every
// transformed source point within max_dist (strict), update = least-squares rigid fit of those pairs, at most 30
// updates, stop when fitness and inlier RMSE both change by less than 1e-6.
//
// Structure: the target cloud is bucketed once per pair into a hashed uniform grid (cell = max_dist, counting sort).
// Each iteration is ONE launch: every thread finds the exact nearest target of its source point in the 27 surrounding
// cells (fp64), the block reduces the fp64 moments of its pairs, and the last block to finish (agent-scope release /
// acquire around a ticket counter) sums the partials in fixed order, solves Kabsch, composes the transform and decides
// convergence.  Later launches of a converged pair return at their first instruction.
#include "lr_internal.h"
#include "lr_kabsch.h"
#include <math.h>

#define LR_ICP_NB 32768          // hash buckets (power of two)

__device__ __forceinline__ uint32_t icp_hash(int ix, int iy, int iz)
{
    uint32_t h = (uint32_t)ix * 73856093u ^ (uint32_t)iy * 19349663u ^ (uint32_t)iz * 83492791u;
    h ^= h >> 15;
    return h & (LR_ICP_NB - 1);
}

__global__ void __launch_bounds__(256)
icp_hist_kernel(const float *__restrict__ xyz, int n, double inv_cell, int32_t *__restrict__ bucket_of, int32_t *__restrict__ hist, lr_zargs z)
{
    if (z.descs) { xyz = z.descs[blockIdx.z].xyz1; n = z.descs[blockIdx.z].n1; }
    lr_z(bucket_of, z, blockIdx.z); lr_z(hist, z, blockIdx.z);
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int ix = (int)floor((double)xyz[3 * j] * inv_cell), iy = (int)floor((double)xyz[3 * j + 1] * inv_cell),
              iz = (int)floor((double)xyz[3 * j + 2] * inv_cell);
    const int b = (int)icp_hash(ix, iy, iz);
    bucket_of[j] = b;
    atomicAdd(&hist[b], 1);
}

// exclusive scan of the LR_ICP_NB bucket counts (one block; each thread owns LR_ICP_NB/1024 consecutive buckets)
__global__ void __launch_bounds__(1024)
icp_scan_kernel(const int32_t *__restrict__ hist, int32_t *__restrict__ start, lr_zargs z)
{
    lr_z(hist, z, blockIdx.z); lr_z(start, z, blockIdx.z);
    __shared__ int s_w[16];
    constexpr int PER = LR_ICP_NB / 1024;
    int local[PER];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { local[k] = sum; sum += hist[threadIdx.x * PER + k]; }
    // block-wide exclusive scan of `sum`
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_w[w];
    const int excl = woff + incl - sum;
#pragma unroll
    for (int k = 0; k < PER; ++k) start[threadIdx.x * PER + k] = excl + local[k];
    if (threadIdx.x == 1023) start[LR_ICP_NB] = excl + sum;
}

__global__ void __launch_bounds__(256)
icp_scatter_kernel(int n, const int32_t *__restrict__ bucket_of, const int32_t *__restrict__ start, int32_t *__restrict__ fill,
                   const float *__restrict__ tgt, float4 *__restrict__ pts, lr_zargs z)
{
    if (z.descs) { tgt = z.descs[blockIdx.z].xyz1; n = z.descs[blockIdx.z].n1; }
    lr_z(bucket_of, z, blockIdx.z); lr_z(start, z, blockIdx.z); lr_z(fill, z, blockIdx.z); lr_z(pts, z, blockIdx.z);
    // the bucket holds the points themselves, { x, y, z, index bits }: the search loop then reads one contiguous 16-byte
    // record per candidate instead of chasing index -> coordinates
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const int b = bucket_of[j];
    pts[start[b] + atomicAdd(&fill[b], 1)] = make_float4(tgt[3 * j], tgt[3 * j + 1], tgt[3 * j + 2], __int_as_float(j));
}

// state block (doubles): [0..15] current T, [16] previous fitness, [17] previous rmse, [18] iteration index k,
// [19] done flag, [20] fitness of the last evaluation, [21] its rmse, [22] its correspondence count, [23] ticket (as int)
#define LR_ICP_STATE 32

__global__ void icp_init_kernel(const double *__restrict__ T_init, const lr_ransac_result *__restrict__ gate, double *__restrict__ state, lr_zargs z)
{
    lr_z(T_init, z, blockIdx.z); lr_z(gate, z, blockIdx.z); lr_z(state, z, blockIdx.z);
    const int k = threadIdx.x;
    if (k < 16) state[k] = T_init[k];
    if (k >= 16 && k < LR_ICP_STATE) state[k] = 0.0;
    if (k == 19 && gate && gate->best_h < 0) state[19] = 1.0;      // no model to refine
}

__global__ void __launch_bounds__(256)
icp_iter_kernel(const float *__restrict__ src, int n0, const float *__restrict__ tgt, const int32_t *__restrict__ start,
                const float4 *__restrict__ pts, double inv_cell, double max_d2, int max_iter, double rel_fit, double rel_rmse,
                double *__restrict__ state, double *__restrict__ partial, lr_zargs z)
{
    if (z.descs) { const lr_pair_desc d = z.descs[blockIdx.z]; src = d.xyz0; n0 = d.n0; tgt = d.xyz1; }
    lr_z(start, z, blockIdx.z); lr_z(pts, z, blockIdx.z); lr_z(state, z, blockIdx.z); lr_z(partial, z, blockIdx.z);
    __shared__ double sm[4][18];
    __shared__ int s_last;
    __shared__ double mom[18];
    if (state[19] != 0.0) return;                                  // converged earlier
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = state[k];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double v[18];
#pragma unroll
    for (int k = 0; k < 18; ++k) v[k] = 0.0;
    if (i < n0) {
        const double px = src[3 * i], py = src[3 * i + 1], pz = src[3 * i + 2];
        double p[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) p[a] = ((T[4 * a] * px + T[4 * a + 1] * py) + T[4 * a + 2] * pz) + T[4 * a + 3];
        const int cx = (int)floor(p[0] * inv_cell), cy = (int)floor(p[1] * inv_cell), cz = (int)floor(p[2] * inv_cell);
        double best = max_d2;
        int bj = -1;
        // bucket ranges of the 27 cells first (independent loads), then the buckets' records
        int cs[27], ce[27];
#pragma unroll
        for (int c = 0; c < 27; ++c) {
            const int b = (int)icp_hash(cx + (c % 3) - 1, cy + ((c / 3) % 3) - 1, cz + (c / 9) - 1);
            cs[c] = start[b]; ce[c] = start[b + 1];
        }
#pragma unroll
        for (int c = 0; c < 27; ++c)
            for (int t = cs[c]; t < ce[c]; ++t) {
                const float4 r = pts[t];
                const int j = __float_as_int(r.w);
                const double qx = (double)r.x - p[0], qy = (double)r.y - p[1], qz = (double)r.z - p[2];
                const double d2 = (qx * qx + qy * qy) + qz * qz;
                // strictly inside the radius; nearest wins, ties to the lower target index (hash collisions
                // can visit a point twice, which changes nothing)
                if (d2 < best || (d2 == best && bj >= 0 && j < bj)) { best = d2; bj = j; }
            }
        if (bj >= 0) {
            const double q[3] = { (double)tgt[3 * bj], (double)tgt[3 * bj + 1], (double)tgt[3 * bj + 2] };
            v[0] = 1.0;
#pragma unroll
            for (int a = 0; a < 3; ++a) { v[1 + a] = p[a]; v[4 + a] = q[a]; }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) v[7 + 3 * a + b] = p[a] * q[b];
            v[16] = best;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 17; ++k) {
        double s = v[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        if (lane == 0) sm[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 17)
        partial[(size_t)blockIdx.x * 18 + threadIdx.x] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
    // ---- last block done: agent-scope release of the partials, ticket, acquire, reduce + solve
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int *ticket = reinterpret_cast<int *>(&state[23]);
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == (int)gridDim.x - 1);
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    if (!s_last) return;
    {
        // block partials summed in block order (reproducible); staged through LDS 64 blocks at a time so that the 17 summing
        // lanes do not walk a chain of dependent global loads
        __shared__ double stage[64 * 18];
        double acc = 0.0;
        for (unsigned b0 = 0; b0 < gridDim.x; b0 += 64) {
            const int nbk = min(64u, gridDim.x - b0);
            for (int t = threadIdx.x; t < nbk * 18; t += 256) stage[t] = partial[(size_t)b0 * 18 + t];
            __syncthreads();
            if (threadIdx.x < 17)
                for (int b = 0; b < nbk; ++b) acc += stage[b * 18 + threadIdx.x];
            __syncthreads();
        }
        if (threadIdx.x < 17) mom[threadIdx.x] = acc;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const double n = mom[0];
    const double fitness = n / (double)n0;
    const double rmse = n > 0.0 ? sqrt(mom[16] / n) : 0.0;
    const int k = (int)state[18];
    bool done = false;
    if (k > 0 && fabs(state[16] - fitness) < rel_fit && fabs(state[17] - rmse) < rel_rmse) done = true;   // ICPConvergenceCriteria
    if (k >= max_iter || n < 3.0) done = true;
    state[20] = fitness; state[21] = rmse; state[22] = n;
    if (!done) {
        double cp[3], cq[3], H[3][3], U[16];
        for (int a = 0; a < 3; ++a) { cp[a] = mom[1 + a] / n; cq[a] = mom[4 + a] / n; }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) H[a][b] = mom[7 + 3 * a + b] - (n * cp[a]) * cq[b];
        lr_rt_from_cov(H, cp, cq, U);
        // T <- U * T
        double Tn[12];
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) Tn[4 * a + b] = (U[4 * a] * T[b] + U[4 * a + 1] * T[4 + b]) + U[4 * a + 2] * T[8 + b];
            Tn[4 * a + 3] = ((U[4 * a] * T[3] + U[4 * a + 1] * T[7]) + U[4 * a + 2] * T[11]) + U[4 * a + 3];
        }
        for (int q = 0; q < 12; ++q) state[q] = Tn[q];
        state[16] = fitness; state[17] = rmse; state[18] = (double)(k + 1);
    } else {
        state[19] = 1.0;
    }
}

__global__ void icp_result_kernel(const double *__restrict__ state, double *__restrict__ T_out, lr_icp_result *__restrict__ res, lr_zargs z)
{
    lr_z(state, z, blockIdx.z); lr_z(T_out, z, blockIdx.z); lr_z(res, z, blockIdx.z);
    const int k = threadIdx.x;
    if (k < 12) T_out[k] = state[k];
    if (k >= 12 && k < 16) T_out[k] = k == 15 ? 1.0 : 0.0;
    if (k == 0 && res) {
        res->fitness = state[20]; res->inlier_rmse = state[21]; res->n_corr = (int32_t)state[22]; res->iterations = (int32_t)state[18];
    }
}

int lr_icp_run(lr_workspace *ws, const float *xyz0, int n0, const float *xyz1, int n1, const double *T_init,
               const lr_ransac_result *gate, double max_dist, int max_iter, double rel_fit, double rel_rmse,
               double *T_out, lr_icp_result *res, hipStream_t st)
{
    LR_REQUIRE(max_dist > 0.0 && max_iter >= 0 && max_iter <= 1000, LR_EINVAL, "lr_icp: bad max_dist / max_iter");
    const double inv_cell = 1.0 / max_dist;
    int32_t *hist = ws->icp_ints, *fill = hist + LR_ICP_NB + 8, *start = fill + LR_ICP_NB + 8;
    LR_TRY_HIP(lr_zero_scratch(ws, hist, sizeof(int32_t) * 2 * (LR_ICP_NB + 8), st));
    const int P = ws->zP;
    hipLaunchKernelGGL(icp_hist_kernel, dim3(lr_cdiv(n1, 256), 1, P), dim3(256), 0, st, xyz1, n1, inv_cell, ws->icp_bucket, hist, ws->z);
    hipLaunchKernelGGL(icp_scan_kernel, dim3(1, 1, P), dim3(1024), 0, st, hist, start, ws->z);
    hipLaunchKernelGGL(icp_scatter_kernel, dim3(lr_cdiv(n1, 256), 1, P), dim3(256), 0, st, n1, ws->icp_bucket, start, fill, xyz1, reinterpret_cast<float4 *>(ws->icp_pts), ws->z);
    hipLaunchKernelGGL(icp_init_kernel, dim3(1, 1, P), dim3(64), 0, st, T_init, gate, ws->icp_state, ws->z);
    const int nb = lr_cdiv(n0, 256);
    for (int k = 0; k <= max_iter; ++k)
        hipLaunchKernelGGL(icp_iter_kernel, dim3(nb, 1, P), dim3(256), 0, st, xyz0, n0, xyz1, start, reinterpret_cast<const float4 *>(ws->icp_pts), inv_cell,
                           max_dist * max_dist, max_iter, rel_fit, rel_rmse, ws->icp_state, ws->icp_part, ws->z);
    hipLaunchKernelGGL(icp_result_kernel, dim3(1, 1, P), dim3(64), 0, st, ws->icp_state, T_out, res, ws->z);
    LR_LAUNCH_CHECK();
    return LR_OK;
}
