// RANSAC hypothesis generation + inlier scoring + Kabsch fit + LS refit on gfx950.
//
// Replaces the third-party loops behind Experiments/algorithms/FR.py:122-139 (Open3D
// registration_ransac_based_on_correspondence) and GC_RANSAC.py:46-49 (pygcransac.findRigidTransform,
// native driver GC-RANSAC/src/pygcransac/src/gcransac_python.cpp:404-624 with the authors' edge-length
// pre-check preemption_edge_length.h:71-128), plus the LS refit FR.py:99-111.
//
// Structure (all counts stay on device, no host round trip):
//   gen     one thread per hypothesis id h: Philox sample -> ELC (fp64) -> minimal-sample Kabsch (fp64, Horn
//           quaternion + Jacobi) -> fp32 model appended to a dense list
//   score   one LANE per surviving hypothesis; the correspondence stream is wave-uniform, so it arrives
//           through scalar loads (s_load_dwordx8) and every VALU op is useful work: ~21 ops per
//           (hypothesis, correspondence), no cross-lane reduction, no LDS traffic.  Inlier count and the
//           fixed-point squared error are both integers, so chunk partials combine with integer atomics
//           and the result does not depend on scheduling
//   select  more inliers, then lower error, then lower hypothesis id (ties on the count are the norm when
//           the inlier noise is far below the threshold, so the error is accumulated for everyone)
//   refit   fp64 moments of the inliers over the ORIGINAL NN pairs -> Kabsch
//
// Arithmetic is spelled out op by op and must stay identical to oracle/oracle.c (build: -ffp-contract=off).
#include "lr_internal.h"
#include <math.h>
#define LR_INF_F __builtin_huge_valf()

// ------------------------------------------------------------------ Philox4x32-10
__device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
        uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

#include "lr_kabsch.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// unweighted Kabsch on NS (3 or 4) sample points held in registers
template <int NS>
__device__ __forceinline__ void kabsch_sample(const double P[NS][3], const double Q[NS][3], double T[16])
{
    double cp[3] = { 0, 0, 0 }, cq[3] = { 0, 0, 0 }, W = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        W = W + 1.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) { cp[a] = cp[a] + 1.0 * P[i][a]; cq[a] = cq[a] + 1.0 * Q[i][a]; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { cp[a] = cp[a] / W; cq[a] = cq[a] / W; }
    double H[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double pc[3], qc[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { pc[a] = P[i][a] - cp[a]; qc[a] = Q[i][a] - cq[a]; }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) H[a][b] = H[a][b] + (1.0 * pc[a]) * qc[b];
    }
    lr_rt_from_cov(H, cp, cq, T);
}

// sample of hypothesis h and its edge-length pre-check; false when the pre-check rejects it
// PROSAC growth function (Chum & Matas 2005, as tabulated by USAC / GC-RANSAC's prosac_sampler.h):
//   T_n = T_N prod_{i<ns} (n-i)/(M-i),  G[ns] = 1,  G[n+1] = G[n] + ceil(T_{n+1} - T_n)
// Draw k uses the first n_k = min(M, ns + #{n in [ns, M) : G[n] <= k}) correspondences.  T_n is evaluated in closed
// form (same operation order as oracle/oracle.c) so that the table is a parallel prefix sum: one block.
__device__ __forceinline__ double prosac_Tn(int n, int M, int ns, double TN)
{
    double t = TN;
    for (int i = 0; i < ns; ++i) t = t * (double)(n - i) / (double)(M - i);
    return t;
}

__global__ void __launch_bounds__(1024)
prosac_growth_kernel(int m_max, const int32_t *__restrict__ m_dev, int ns, int TN, int32_t *__restrict__ G, lr_zargs z)
{
    lr_z(m_dev, z, blockIdx.z); lr_z(G, z, blockIdx.z);
    __shared__ long long s_w[16];
    __shared__ long long s_carry;
    const int M = m_dev ? min(*m_dev, m_max) : m_max;
    if (threadIdx.x == 0) s_carry = 1;       // G[ns] = 1
    __syncthreads();
    // elements n = ns .. M-1 carry d_n = ceil(T_{n+1} - T_n) >= 1; G[n] = 1 + sum_{j<n} d_j, written for n = ns .. M.
    // Four consecutive elements per thread (five evaluations of T instead of eight, a quarter of the block-wide scans: the kernel is one
    // block per pair and sits in a single pair's critical path -- 30 us with one element per thread)
    constexpr int E = 4;
    for (int base = ns; base < M; base += E * 1024) {
        const int n0 = base + E * (int)threadIdx.x;
        long long d[E], tot = 0;
        double t_prev = n0 < M ? prosac_Tn(n0, M, ns, (double)TN) : 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            d[e] = 0;
            if (n0 + e < M) {
                const double t_next = prosac_Tn(n0 + e + 1, M, ns, (double)TN);
                d[e] = (long long)ceil(t_next - t_prev);
                if (d[e] < 1) d[e] = 1;
                t_prev = t_next;
            }
            tot += d[e];
        }
        long long inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const long long v = __shfl_up(inc, o); if ((int)(threadIdx.x & 63) >= o) inc += v; }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = inc;
        __syncthreads();
        long long pre = s_carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) pre += s_w[w];
        long long g = pre + inc - tot;                            // G[n0]
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (n0 + e < M) {
                G[n0 + e] = (int32_t)(g < 0x3fffffffLL ? g : 0x3fffffffLL);
                g += d[e];
                if (n0 + e == M - 1) G[M] = (int32_t)(g < 0x3fffffffLL ? g : 0x3fffffffLL);
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = pre + inc;
        __syncthreads();
    }
}

template <int NS>
__device__ __forceinline__ bool hypothesis_sample(const float *__restrict__ corr8, int m, uint64_t seed, uint64_t h,
                                                  int use_elc, double P[NS][3], double Q[NS][3],
                                                  const int32_t *__restrict__ G, int TN, int unique = 0)
{
    uint32_t c[4] = { (uint32_t)h, (uint32_t)(h >> 32), 0u, 0u };
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    // PROSAC: draw k = h + 1 takes NS-1 indices from the first n-1 correspondences and the n-th one
    int n_top = 0;
    if (G && h < (uint64_t)TN && m > NS) {
        const int k = (int)h + 1;
        int lo = NS, hi = m;                       // first n in [NS, m) with G[n] > k
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (G[mid] <= k) lo = mid + 1; else hi = mid; }
        n_top = min(m, lo);                        // = NS + #{n : G[n] <= k}
    }
    uint32_t sidx[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        uint32_t s = __umulhi(c[k], (uint32_t)m);
        if (n_top) s = k < NS - 1 ? __umulhi(c[k], (uint32_t)(n_top - 1)) : (uint32_t)(n_top - 1);
        sidx[k] = s;
        // the correspondence's six floats sit at every second place of the first 48 bytes of its pair's record (lr_corr_at): three 16-byte
        // gathers and a selection by parity instead of six 4-byte gathers -- a gather costs the memory pipe a cache line per lane whatever
        // its width, and the gathers are what this kernel waits for
        const f32x4 *rec = reinterpret_cast<const f32x4 *>(corr8 + (size_t)(s >> 1) * 16);
        const f32x4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
        const bool odd = (s & 1u) != 0u;
        P[k][0] = (double)(odd ? r0.y : r0.x); P[k][1] = (double)(odd ? r0.w : r0.z); P[k][2] = (double)(odd ? r1.y : r1.x);
        Q[k][0] = (double)(odd ? r1.w : r1.z); Q[k][1] = (double)(odd ? r2.y : r2.x); Q[k][2] = (double)(odd ? r2.w : r2.z);
    }
    bool ok = true;
    if (unique) {       // GC-RANSAC's samplers draw distinct indices: a repeated index rejects the draw
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int j = i + 1; j < NS; ++j) if (sidx[i] == sidx[j]) ok = false;
    }
    if (!use_elc) return ok;
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int j = i + 1; j < NS; ++j) {
            double sx = P[j][0] - P[i][0], sy = P[j][1] - P[i][1], sz = P[j][2] - P[i][2];
            double tx = Q[j][0] - Q[i][0], ty = Q[j][1] - Q[i][1], tz = Q[j][2] - Q[i][2];
            // the contract compares the LENGTHS (oracle.c: ds < dt * 0.9 || dt < ds * 0.9, sqrt and products rounded in fp64).  The
            // roundings move either side by < 4 ulp, so away from equality the squares decide the same way, and the six fp64 square
            // roots per draw -- a third of this kernel -- are only taken inside a relative band of 1e-13 around it (or when a square is
            // zero or not finite: every comparison below is then false)
            const double a2 = (sx * sx + sy * sy) + sz * sz, b2 = (tx * tx + ty * ty) + tz * tz;
            const double band = 1e-13 * (a2 + b2);
            if (fabs(a2 - 0.81 * b2) > band && fabs(b2 - 0.81 * a2) > band) {
                if (a2 < 0.81 * b2 || b2 < 0.81 * a2) ok = false;
            } else {
                const double ds = sqrt(a2), dt = sqrt(b2);
                if (ds < dt * 0.9 || dt < ds * 0.9) ok = false;
            }
        }
    return ok;
}

// ------------------------------------------------------------------ pilot-ordered scoring: shared state (see "score" below)
#define LR_SC_HEAD 256
#define LR_SC_NB 128
#define LR_SC_W 0.0625f
#define LR_SC_MIN_M 2048
#define LR_SC_MIN_V 128
#define LR_SC_MIN_PAIRS 4       // calls of at most this many pairs score every model over every correspondence (no ordering passes)
// what the head and order passes leave for the main pass (one per pair arena)
struct lr_score_info {
    unsigned long long pilot_key;     // max over the models of (head inliers << 32 | ~slot): the pilot
    int32_t prune;                    // 1: corr8s / perm / glen describe this batch
    int32_t n_sorted;                 // records in the sorted copy (M - head)
    uint32_t box[6];                  // bounding box of the source points (lr_ford: order-preserving bit patterns; min x y z, max x y z)
    int32_t bad;                      // a source coordinate is not finite: no pruning
    int32_t pad;
    int32_t hist[LR_SC_NB];           // records per residual bucket
    int32_t fill[LR_SC_NB];           // next free position of every bucket while the sorted copy is written
    int32_t offs[LR_SC_NB + 1];       // start of every residual bucket in the sorted copy (offs[b + 1]: its end)
};
// float <-> unsigned with the same order (atomicMin / atomicMax on floats of either sign)
__device__ __forceinline__ uint32_t lr_ford(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float lr_ford_inv(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }
static_assert(sizeof(lr_score_info) <= LR_SC_INFO_BYTES, "lr_score_info does not fit its scratch block");

__device__ __forceinline__ void lr_score_info_reset(lr_score_info *info)      // by one block of >= LR_SC_NB threads
{
    const int t = threadIdx.x;
    if (t < LR_SC_NB) info->hist[t] = 0;
    if (t < 3) { info->box[t] = 0xffffffffu; info->box[3 + t] = 0u; }
    if (t == 0) { info->pilot_key = 0ull; info->bad = 0; info->prune = 0; }
}

// ------------------------------------------------------------------ gen
// 256 hypothesis ids per block.  Phase 1: every thread draws its sample and runs the pre-check (cheap, ~93 % fail on
// 3-point samples at a 40 % inlier ratio).  Phase 2: the survivors are compacted through LDS so that the expensive fp64
// Kabsch runs on densely packed lanes of one wave instead of a few stragglers in every wave.
template <int NS>
__global__ void __launch_bounds__(256)
ransac_gen_kernel(const float *__restrict__ corr8, int m_max, const int32_t *__restrict__ m_dev, lr_ransac_params p,
                  int h_begin, int h_end, int32_t *__restrict__ model_h,
                  uint32_t *__restrict__ score_cnt, unsigned long long *__restrict__ score_ssq,
                  int32_t *__restrict__ counters, const int32_t *__restrict__ G, int TN,
                  lr_score_info *__restrict__ info, lr_zargs z)
{
    // ids that passed the pre-check -> dense list model_h[0 .. NVALID) (appended by whole groups: one device atomic per block and
    // FIT_AT ids); ransac_fit_kernel estimates their models.  The fit (fp64 Kabsch, a dependent chain of ~14 us for a wave) used to
    // run here, on the 18 of 256 ids of a group that pass the edge-length check: a quarter of ONE wave per block for the full length
    // of the fit, 1.25 such waves per SIMD -- 61 of the kernel's 111 us per 32-pair launch were that wait.
    constexpr int FIT_AT = 192;
    __shared__ int s_pass[FIT_AT + 256];
    __shared__ int s_np, s_base;
    lr_z(info, z, blockIdx.z);
    if (blockIdx.x == 0) lr_score_info_reset(info);       // the scoring passes of this batch elect their pilot model afresh
    lr_z(corr8, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(model_h, z, blockIdx.z); lr_z(score_cnt, z, blockIdx.z); lr_z(score_ssq, z, blockIdx.z); lr_z(counters, z, blockIdx.z); lr_z(G, z, blockIdx.z);
    const int m = m_dev ? min(*m_dev, m_max) : m_max;
    if (m <= 0 || reinterpret_cast<const lr_ransac_state *>(counters + LR_CNT_COUNT)->done) return;
    if (threadIdx.x == 0) s_np = 0;
    __syncthreads();
    // (a capped grid strides over the 256-id groups: the launches of the batches behind an early exit then cost a few hundred
    // idle blocks instead of tens of thousands)
    for (int blk = blockIdx.x; blk * 256 < h_end - h_begin; blk += gridDim.x) {
        const int h = h_begin + blk * 256 + threadIdx.x;
        {
            double P[NS][3], Q[NS][3];
            if (h < h_end && hypothesis_sample<NS>(corr8, m, p.seed, (uint64_t)h, p.use_elc == 1, P, Q, G, TN, p.sampler != 0)) s_pass[atomicAdd(&s_np, 1)] = h;
        }
        __syncthreads();
        const int np = s_np;
        const bool last = (blk + (int)gridDim.x) * 256 >= h_end - h_begin;
        if (threadIdx.x == 0 && np > 0 && (np >= FIT_AT || last)) s_base = atomicAdd(&counters[LR_CNT_NVALID], np);
        __syncthreads();                             // (every thread has read np before the next group's ids are appended)
        if (np < FIT_AT && !last) continue;          // (block-uniform: np and last are the same for every thread)
        for (int e = threadIdx.x; e < np; e += 256) {
            const int slot = s_base + e;
            model_h[slot] = s_pass[e];
            score_cnt[slot] = 0u;
            score_ssq[slot] = 0ull;
        }
        if (!last) {
            __syncthreads();
            if (threadIdx.x == 0) s_np = 0;
            __syncthreads();
        }
    }
}

// the models of the ids ransac_gen_kernel listed: a thread per list slot re-draws its sample (same Philox counter) and fits it
template <int NS>
__global__ void __launch_bounds__(64)
ransac_fit_kernel(const float *__restrict__ corr8, int m_max, const int32_t *__restrict__ m_dev, lr_ransac_params p,
                  float *__restrict__ models, double *__restrict__ models64, const int32_t *__restrict__ model_h,
                  const int32_t *__restrict__ counters, const int32_t *__restrict__ G, int TN, int model_stride, lr_zargs z)
{
    lr_z(corr8, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(models, z, blockIdx.z); lr_z(models64, z, blockIdx.z); lr_z(model_h, z, blockIdx.z); lr_z(counters, z, blockIdx.z); lr_z(G, z, blockIdx.z);
    const int m = m_dev ? min(*m_dev, m_max) : m_max;
    if (m <= 0 || reinterpret_cast<const lr_ransac_state *>(counters + LR_CNT_COUNT)->done) return;
    const int V = counters[LR_CNT_NVALID];
    // (a capped grid strides over the list: with the pre-check the list is a fraction of the batch, and tens of thousands of blocks
    // that only find that out cost more than the fits)
    for (int slot = blockIdx.x * 64 + threadIdx.x; slot < V; slot += gridDim.x * 64) {
        const int hh = model_h[slot];
        double P[NS][3], Q[NS][3], T[16];
        hypothesis_sample<NS>(corr8, m, p.seed, (uint64_t)hh, 0, P, Q, G, TN);
        kabsch_sample<NS>(P, Q, T);
        // fp32 models component-major ([12][model_stride]: the scoring kernel's 64 lanes read 64 consecutive floats per
        // component), fp64 models row-major (only the winner is ever read back)
#pragma unroll
        for (int k = 0; k < 12; ++k) { models[(size_t)k * model_stride + slot] = (float)T[k]; models64[(size_t)slot * 12 + k] = T[k]; }
    }
}


// ------------------------------------------------------------------ SPRT pre-verification (--fast_rejection SPRT)
// See oracle/oracle.c (sprt_test) for the test and its sources.  One LANE per model, the first LR_SPRT_HORIZON
// correspondences in list order as a wave-uniform stream (scalar loads); a wave leaves as soon as all its models are
// decided.  Survivors are appended to a second dense model list (their order does not matter: the winner is chosen by
// score and hypothesis id); the consistent / verified point counts of the rejected models feed the next batch's design.
#define LR_SPRT_HORIZON 256
#define LR_SPRT_EPS0 0.1
#define LR_SPRT_DELTA0 0.01
__device__ __forceinline__ double lr_sprt_threshold(double eps, double delta)
{
    const double C = (1.0 - delta) * lr_det_log((1.0 - delta) / (1.0 - eps)) + delta * lr_det_log(delta / eps);
    const double K = (200.0 * C) / 1.0 + 1.0;
    double A = K;
    for (int i = 0; i < 10; ++i) A = K + lr_det_log(A);
    return A;
}

__global__ void __launch_bounds__(256)
ransac_sprt_kernel(const float *__restrict__ corr8, int m_max, const int32_t *__restrict__ m_dev, float thr2,
                   const float *__restrict__ models, const double *__restrict__ models64, const int32_t *__restrict__ model_h,
                   float *__restrict__ models2, double *__restrict__ models64_2, int32_t *__restrict__ model_h2,
                   uint32_t *__restrict__ score_cnt, unsigned long long *__restrict__ score_ssq, int32_t *__restrict__ counters,
                   int model_stride, lr_zargs z)
{
    lr_z(corr8, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(models, z, blockIdx.z); lr_z(models64, z, blockIdx.z); lr_z(model_h, z, blockIdx.z);
    lr_z(models2, z, blockIdx.z); lr_z(models64_2, z, blockIdx.z); lr_z(model_h2, z, blockIdx.z); lr_z(score_cnt, z, blockIdx.z);
    lr_z(score_ssq, z, blockIdx.z); lr_z(counters, z, blockIdx.z);
    lr_ransac_state *state = reinterpret_cast<lr_ransac_state *>(counters + LR_CNT_COUNT);
    const int m = m_dev ? min(*m_dev, m_max) : m_max;
    const int V = counters[LR_CNT_NVALID];
    if (V <= 0 || m <= 0 || state->done) return;
    const int lane = threadIdx.x & 63;
    const int group = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    if (group * 64 >= V) return;
    const double eps = state->sprt_eps > 0.0 ? state->sprt_eps : LR_SPRT_EPS0, delta = state->sprt_delta > 0.0 ? state->sprt_delta : LR_SPRT_DELTA0;
    const double A = lr_sprt_threshold(eps, delta), fin = delta / eps, fout = (1.0 - delta) / (1.0 - eps);
    const int slot = group * 64 + lane;
    const bool active = slot < V;
    const float *mp = models + (active ? slot : 0);
    const size_t ms = (size_t)model_stride;
    const float r00 = mp[0], r01 = mp[ms], r02 = mp[2 * ms], tx = mp[3 * ms];
    const float r10 = mp[4 * ms], r11 = mp[5 * ms], r12 = mp[6 * ms], ty = mp[7 * ms];
    const float r20 = mp[8 * ms], r21 = mp[9 * ms], r22 = mp[10 * ms], tz = mp[11 * ms];
    const int n = min(m, LR_SPRT_HORIZON);
    double lambda = 1.0;
    int inl = 0, k_rej = 0;
    bool rejected = !active;
    for (int i = 0; i < n; ++i) {
        if (__builtin_amdgcn_ballot_w64(!rejected) == 0ull) break;
        const float px = corr8[lr_corr_at(i, 0)], py = corr8[lr_corr_at(i, 1)], pz = corr8[lr_corr_at(i, 2)];
        const float x = __builtin_fmaf(r00, px, __builtin_fmaf(r01, py, __builtin_fmaf(r02, pz, tx)));
        const float y = __builtin_fmaf(r10, px, __builtin_fmaf(r11, py, __builtin_fmaf(r12, pz, ty)));
        const float zz = __builtin_fmaf(r20, px, __builtin_fmaf(r21, py, __builtin_fmaf(r22, pz, tz)));
        const float dx = x - corr8[lr_corr_at(i, 3)], dy = y - corr8[lr_corr_at(i, 4)], dz = zz - corr8[lr_corr_at(i, 5)];
        const float d2 = __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
        if (!rejected) {
            if (d2 < thr2) { inl += 1; lambda = lambda * fin; } else lambda = lambda * fout;
            if (lambda > A) { rejected = true; k_rej = i + 1; }
        }
    }
    const bool keep = active && !rejected;
    // survivors -> the second list
    const unsigned long long kb = __builtin_amdgcn_ballot_w64(keep);
    int base = 0;
    if (lane == 0 && kb) base = atomicAdd(&counters[LR_CNT_NVALID2], (int)__builtin_popcountll(kb));
    base = __shfl(base, 0);
    if (keep) {
        const int dst = base + (int)__builtin_popcountll(kb & ((1ull << lane) - 1ull));
        models2[dst] = r00; models2[ms + dst] = r01; models2[2 * ms + dst] = r02; models2[3 * ms + dst] = tx;
        models2[4 * ms + dst] = r10; models2[5 * ms + dst] = r11; models2[6 * ms + dst] = r12; models2[7 * ms + dst] = ty;
        models2[8 * ms + dst] = r20; models2[9 * ms + dst] = r21; models2[10 * ms + dst] = r22; models2[11 * ms + dst] = tz;
#pragma unroll
        for (int k = 0; k < 12; ++k) models64_2[(size_t)dst * 12 + k] = models64[(size_t)slot * 12 + k];
        model_h2[dst] = model_h[slot];
        score_cnt[dst] = 0u; score_ssq[dst] = 0ull;
    }
    // rejected models: consistent / verified points (integer sums: order independent)
    unsigned long long ri = (active && rejected) ? (unsigned long long)inl : 0ull, rp = (active && rejected) ? (unsigned long long)k_rej : 0ull;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { ri += __shfl_xor(ri, o); rp += __shfl_xor(rp, o); }
    if (lane == 0 && rp) { atomicAdd(&state->rej_inl, ri); atomicAdd(&state->rej_pts, rp); }
}

// ------------------------------------------------------------------ score
// Work items = (group of 64 hypotheses) x (chunk of correspondences); blocks stride over them.
//
// Pilot-ordered scoring (exact; the oracle scores everything and gets the same integers).  Most of the models that pass the
// pre-check are GOOD models -- all-inlier samples: 83 % of the survivors on the benchmark pair -- and they agree with each other
// to within the sensor noise, so they share their inlier set: scoring each of them over all M correspondences re-discovers the
// same 60 % of outliers thousands of times.  For two models v, * and a correspondence (p, q):
//     |T_v p - q| >= |T_* p - q| - |T_v p - T_* p|  >=  r_*(p, q) - eps_v,     eps_v = |R_v - R_*|_F rho + |(R_v - R_*) c0 + t_v - t_*|
// (c0, rho: centre and radius of a ball around the source points), so (p, q) can only be an inlier of v when r_* < thr + eps_v.
//   head    every model over the first LR_SC_HEAD correspondences in list order; the model with the most inliers there is the
//           pilot * (ransac_score_kernel<1>)
//   order   residuals r_* of the other correspondences -> LR_SC_NB buckets of LR_SC_W metres -> a copy of the records sorted
//           by bucket; per model the last bucket it has to look at, and the models sorted by that bucket, longest first
//           (ransac_order_kernel: one block per pair)
//   main    a group of 64 models of similar reach scans its prefix of the sorted records (ransac_score_kernel<0>)
// Counts and error sums are integer sums over the inliers, so neither the order of the records nor the skipped outliers change
// them.  A pilot that is a bad model only costs the saving (eps is large for everyone), never the result.  Rounding: r_*, eps
// and the scoring arithmetic are fp32 evaluations of quantities of the size of the coordinates; `slack` (1 cm + 4e-6 of that
// size) is two orders of magnitude above their rounding errors.  Not worth its set-up below LR_SC_MIN_M correspondences or
// LR_SC_MIN_V models: then the head is empty and the main pass scans the list as it lies.
#ifndef LR_SCORE_CHUNK
#define LR_SCORE_CHUNK 256
#endif
__device__ __forceinline__ int lr_sc_head(int m, int V) { return (m >= LR_SC_MIN_M && V >= LR_SC_MIN_V) ? LR_SC_HEAD : 0; }

// one lane's model over the records [begin, end) of a stream (begin even; a trailing odd correspondence is taken alone)
struct lr_model12 { float r00, r01, r02, tx, r10, r11, r12, ty, r20, r21, r22, tz; };
__device__ __forceinline__ lr_model12 lr_load_model(const float *__restrict__ mp, size_t ms)
{
    lr_model12 M;
    M.r00 = mp[0]; M.r01 = mp[ms]; M.r02 = mp[2 * ms]; M.tx = mp[3 * ms];
    M.r10 = mp[4 * ms]; M.r11 = mp[5 * ms]; M.r12 = mp[6 * ms]; M.ty = mp[7 * ms];
    M.r20 = mp[8 * ms]; M.r21 = mp[9 * ms]; M.r22 = mp[10 * ms]; M.tz = mp[11 * ms];
    return M;
}
__device__ __forceinline__ float lr_model_d2(const lr_model12 &M, const float *__restrict__ corr8, int i)
{
    const float px = corr8[lr_corr_at(i, 0)], py = corr8[lr_corr_at(i, 1)], pz = corr8[lr_corr_at(i, 2)];
    const float x = __builtin_fmaf(M.r00, px, __builtin_fmaf(M.r01, py, __builtin_fmaf(M.r02, pz, M.tx)));
    const float y = __builtin_fmaf(M.r10, px, __builtin_fmaf(M.r11, py, __builtin_fmaf(M.r12, pz, M.ty)));
    const float z = __builtin_fmaf(M.r20, px, __builtin_fmaf(M.r21, py, __builtin_fmaf(M.r22, pz, M.tz)));
    const float dx = x - corr8[lr_corr_at(i, 3)], dy = y - corr8[lr_corr_at(i, 4)], dz = z - corr8[lr_corr_at(i, 5)];
    return __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
}
__device__ __forceinline__ void lr_score_stream(const float *__restrict__ corr8, int begin, int end, int sub, float thr2,
                                                const lr_model12 &M, uint32_t &cnt, unsigned long long &ssq)
{
    // two correspondences per packed instruction: the record of a pair is wave-uniform (scalar load) and its halves feed
    // packed fp32 instructions directly; every component keeps the fma order of the arithmetic contract.
    // The error sum runs in 32 bits over sub-blocks short enough not to overflow (sub * thr2 * 2^20 < 2^32).
    // Scalar loads return out of order, so every wait for one drains all of them: the stream is read TWO records (four
    // correspondences) per wait, the loads of the next two issued before the arithmetic of these two -- one wait per ~50 vector
    // instructions with a full iteration of lookahead (a record per wait had the next record's loads and their wait ~25 instructions apart).
    const f32x2 R00 = { M.r00, M.r00 }, R01 = { M.r01, M.r01 }, R02 = { M.r02, M.r02 }, TX = { M.tx, M.tx };
    const f32x2 R10 = { M.r10, M.r10 }, R11 = { M.r11, M.r11 }, R12 = { M.r12, M.r12 }, TY = { M.ty, M.ty };
    const f32x2 R20 = { M.r20, M.r20 }, R21 = { M.r21, M.r21 }, R22 = { M.r22, M.r22 }, TZ = { M.tz, M.tz };
    const f32x2 SC = { 1048576.0f, 1048576.0f };
    uint32_t q32 = 0;
    auto pair_of = [&](const f32x2 px, const f32x2 py, const f32x2 pz, const f32x2 qx, const f32x2 qy, const f32x2 qz) {
        const f32x2 x = __builtin_elementwise_fma(R00, px, __builtin_elementwise_fma(R01, py, __builtin_elementwise_fma(R02, pz, TX)));
        const f32x2 y = __builtin_elementwise_fma(R10, px, __builtin_elementwise_fma(R11, py, __builtin_elementwise_fma(R12, pz, TY)));
        const f32x2 z = __builtin_elementwise_fma(R20, px, __builtin_elementwise_fma(R21, py, __builtin_elementwise_fma(R22, pz, TZ)));
        const f32x2 dx = x - qx, dy = y - qy, dz = z - qz;
        const f32x2 d2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));
        const f32x2 fx = d2 * SC;
        const bool in0 = d2.x < thr2, in1 = d2.y < thr2;
        cnt += (in0 ? 1u : 0u) + (in1 ? 1u : 0u);
        q32 += (in0 ? (uint32_t)fx.x : 0u) + (in1 ? (uint32_t)fx.y : 0u);
    };
    const int pend = end & ~1;
    for (int b0 = begin; b0 < pend; b0 += sub) {
        const int b1 = min(pend, b0 + sub);
        q32 = 0;
        int i = b0;
        const f32x2 *rec = reinterpret_cast<const f32x2 *>(corr8);
        if (i + 4 <= b1) {
            f32x2 c0[6], c1[6], n0[6], n1[6];
            auto load2 = [&](int at, f32x2 (&r0)[6], f32x2 (&r1)[6]) {
#pragma unroll
                for (int k = 0; k < 6; ++k) { r0[k] = rec[(size_t)(at >> 1) * 8 + k]; r1[k] = rec[(size_t)(at >> 1) * 8 + 8 + k]; }
                __builtin_amdgcn_sched_barrier(0);       // (the loads stay ahead of the arithmetic that follows: left alone the scheduler sinks them to their use)
            };
            load2(i, c0, c1);
            // ping-pong between two register sets (no copies); the last load of a sub-block re-reads its own records: no read past the stream
            while (i + 4 <= b1) {
                // (wait for the records in hand BEFORE the next loads are issued: scalar loads return out of order, a wait placed after
                // them -- where the compiler would put it, at the first use -- drains the prefetch as well)
                __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0)
                load2(i + 8 <= b1 ? i + 4 : i, n0, n1);
                pair_of(c0[0], c0[1], c0[2], c0[3], c0[4], c0[5]);
                pair_of(c1[0], c1[1], c1[2], c1[3], c1[4], c1[5]);
                i += 4;
                if (i + 4 > b1) break;
                __builtin_amdgcn_s_waitcnt(0xC07F);
                load2(i + 8 <= b1 ? i + 4 : i, c0, c1);
                pair_of(n0[0], n0[1], n0[2], n0[3], n0[4], n0[5]);
                pair_of(n1[0], n1[1], n1[2], n1[3], n1[4], n1[5]);
                i += 4;
            }
        }
        if (i < b1) {        // one record left in the sub-block
            const f32x2 *q = rec + (size_t)(i >> 1) * 8;
            pair_of(q[0], q[1], q[2], q[3], q[4], q[5]);
        }
        ssq += q32;
    }
    if (pend < end && begin < end) {      // (a chunk past the end -- begin > end -- owns nothing, not even the odd last correspondence)
        const float d2 = lr_model_d2(M, corr8, pend);
        if (d2 < thr2) { cnt += 1u; ssq += (uint32_t)(d2 * 1048576.0f); }
    }
}

template <int HEAD>
__global__ void __launch_bounds__(256)
ransac_score_kernel(const float *__restrict__ corr8, const float *__restrict__ corr8s, int m_max, const int32_t *__restrict__ m_dev, float thr2,
                    const float *__restrict__ models, const float *__restrict__ models_s, uint32_t *__restrict__ score_cnt,
                    unsigned long long *__restrict__ score_ssq, const int32_t *__restrict__ counters, lr_score_info *__restrict__ info,
                    const int32_t *__restrict__ perm, const int32_t *__restrict__ glen, int sub, int model_stride, int vslot,
                    int gx, int total, int allow_prune, lr_zargs z)
{
    // 1-D XCD-aware grid -> (block of the pair, pair): the blocks of one pair run on one XCD, whose L2 then serves the pair's
    // record stream (0.5 MB, read by every wave) and its models
    int logical;
    if (!lr_xcd_block(total, logical)) return;
    const int bxi = logical % gx, pair = logical / gx;
    lr_z(corr8, z, pair); lr_z(corr8s, z, pair); lr_z(m_dev, z, pair); lr_z(models, z, pair); lr_z(models_s, z, pair); lr_z(score_cnt, z, pair); lr_z(score_ssq, z, pair);
    lr_z(counters, z, pair); lr_z(info, z, pair); lr_z(perm, z, pair); lr_z(glen, z, pair);
    const int m = m_dev ? min(*m_dev, m_max) : m_max;
    const int V = counters[vslot];           // LR_CNT_NVALID, or LR_CNT_NVALID2 behind the SPRT pre-verification
    const int hb = (V + 63) >> 6;
    if (hb == 0 || m <= 0 || reinterpret_cast<const lr_ransac_state *>(counters + LR_CNT_COUNT)->done) return;
    const int K0 = allow_prune ? lr_sc_head(m, V) : 0;     // (the host leaves the ordering passes out for short batches: same decision here)
    const int lane = threadIdx.x & 63;
    const size_t ms = (size_t)model_stride;
    // a block is four independent waves (one-wave blocks cap the CU at half its wave slots); wave w of the pair's gx blocks takes
    // the work items w, w + W, ...
    // (the wave index is wave-uniform, which the compiler cannot see through threadIdx: without the readfirstlane the record
    // loads become per-lane global loads with VALU address arithmetic instead of scalar loads)
    const int W = gx * 4, w0 = bxi * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (HEAD) {
        if (K0 == 0) return;
        // work item = (group of 64 models, quarter of the head): the head is short, so a group per wave would leave three waves in four
        // idle; the pilot is elected on the first quarter's counts alone (any model is a valid pilot, see above)
        constexpr int HQ = 4, QL = LR_SC_HEAD / HQ;
        for (int w = w0; w < hb * HQ; w += W) {
            const int g = w % hb, qi = w / hb;
            const int slot = g * 64 + lane;
            const bool active = slot < V;
            const lr_model12 M = lr_load_model(models + (active ? slot : 0), ms);
            uint32_t cnt = 0; unsigned long long ssq = 0;
            lr_score_stream(corr8, qi * QL, (qi + 1) * QL, sub, thr2, M, cnt, ssq);
            if (active && cnt) { atomicAdd(&score_cnt[slot], cnt); atomicAdd(&score_ssq[slot], ssq); }
            if (qi == 0) {
                unsigned long long key = active ? ((unsigned long long)cnt << 32) | (unsigned long long)(0xffffffffu - (uint32_t)slot) : 0ull;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) { const unsigned long long k2 = __shfl_xor(key, o); key = k2 > key ? k2 : key; }
                if (lane == 0) atomicMax(&info->pilot_key, key);
            }
        }
        return;
    }
    const bool prune = K0 > 0;
    const float *__restrict__ stream = prune ? corr8s : corr8;
    const int L = prune ? m - K0 : m;
    int chunks = W / hb;
    const int cmax = L / LR_SCORE_CHUNK > 0 ? L / LR_SCORE_CHUNK : 1;   // at least LR_SCORE_CHUNK correspondences per work item
    if (chunks > cmax) chunks = cmax;
    if (chunks < 1) chunks = 1;
    const int per = ((L + chunks - 1) / chunks + 1) & ~1;      // even: chunks start on a pair boundary
    for (int w = w0; w < hb * chunks; w += W) {
        const int g = w % hb, c = w / hb;
        const int len = prune ? min(glen[g], (L + 1) & ~1) : L;      // records this group has to look at
        const int begin = c * per, end = min(len, begin + per);
        if (begin >= end) continue;
        const int spos = g * 64 + lane;        // position in the model order of this batch's scoring (reach-sorted, or the list as it lies)
        const bool active = spos < V;
        const lr_model12 M = lr_load_model((prune ? models_s : models) + (active ? spos : 0), ms);
        uint32_t cnt = 0;
        unsigned long long ssq = 0;
        lr_score_stream(stream, begin, end, sub, thr2, M, cnt, ssq);
        if (active && cnt) {
            const int slot = prune ? perm[spos] : spos;
            atomicAdd(&score_cnt[slot], cnt); atomicAdd(&score_ssq[slot], ssq);
        }
    }
}

// order pass, part 1 (a thread per correspondence): residual under the pilot model -> bucket; bucket histogram and the bounding box of
// the source points through one atomic per block and value
__global__ void __launch_bounds__(256)
ransac_resid_kernel(const float *__restrict__ corr8, int m_max, const int32_t *__restrict__ m_dev, const float *__restrict__ models,
                    const int32_t *__restrict__ counters, lr_score_info *__restrict__ info, uint8_t *__restrict__ cb, int model_stride, int vslot, lr_zargs z)
{
    __shared__ int s_h[LR_SC_NB];
    __shared__ float s_red[4][6];
    __shared__ int s_bad;
    lr_z(corr8, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(models, z, blockIdx.z); lr_z(counters, z, blockIdx.z); lr_z(info, z, blockIdx.z); lr_z(cb, z, blockIdx.z);
    const int m = m_dev ? min(*m_dev, m_max) : m_max;
    const int V = counters[vslot];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = blockIdx.x * 256 + tid;
    if (V <= 0 || blockIdx.x * 256 >= m || reinterpret_cast<const lr_ransac_state *>(counters + LR_CNT_COUNT)->done) return;
    const int K0 = lr_sc_head(m, V);
    if (K0 == 0) return;
    if (tid < LR_SC_NB) s_h[tid] = 0;
    if (tid == 0) s_bad = 0;
    __syncthreads();
    int ps = (int)(0xffffffffu - (uint32_t)info->pilot_key);
    ps = min(max(ps, 0), V - 1);
    const lr_model12 P = lr_load_model(models + ps, (size_t)model_stride);
    float lo[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, hi[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    if (c < m) {
        bool bad = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) { const float v = corr8[lr_corr_at(c, a)]; bad |= !(fabsf(v) < 3.0e38f); lo[a] = v; hi[a] = v; }
        if (bad) s_bad = 1;
        if (c >= K0) {
            constexpr float INVW = 1.0f / LR_SC_W, RMAX = LR_SC_NB * LR_SC_W;
            const float r = sqrtf(lr_model_d2(P, corr8, c));
            int b = LR_SC_NB - 1;
            if (r < RMAX) b = min((int)(r * INVW), LR_SC_NB - 1);      // (NaN, inf: last bucket)
            cb[c] = (uint8_t)b;
            atomicAdd(&s_h[b], 1);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], o)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o)); }
    if (lane == 0) { for (int a = 0; a < 3; ++a) { s_red[wave][a] = lo[a]; s_red[wave][3 + a] = hi[a]; } }
    __syncthreads();
    if (tid < LR_SC_NB && s_h[tid]) atomicAdd(&info->hist[tid], s_h[tid]);
    if (tid < 6) {
        float v = s_red[0][tid];
        for (int w = 1; w < 4; ++w) v = tid < 3 ? fminf(v, s_red[w][tid]) : fmaxf(v, s_red[w][tid]);
        if (tid < 3) atomicMin(&info->box[tid], lr_ford(v)); else atomicMax(&info->box[tid], lr_ford(v));
    }
    if (tid == 0 && s_bad) info->bad = 1;
}

// order pass, part 2 (one block per pair): bucket offsets; reach of every model = the last bucket it has to look at; models sorted by
// reach, longest first; records every group of 64 sorted models scans
__global__ void __launch_bounds__(1024)
ransac_order_kernel(int m_max, const int32_t *__restrict__ m_dev, float thr2, const float *__restrict__ models, int32_t *__restrict__ counters,
                    lr_score_info *__restrict__ info, int32_t *__restrict__ perm, int32_t *__restrict__ glen, uint8_t *__restrict__ mb,
                    float *__restrict__ models_s, int model_stride, int vslot, lr_zargs z)
{
    __shared__ int s_h[LR_SC_NB], s_off[LR_SC_NB + 1], s_fill[LR_SC_NB];
    __shared__ unsigned long long s_ev[16];
    lr_z(m_dev, z, blockIdx.z); lr_z(models, z, blockIdx.z); lr_z(counters, z, blockIdx.z); lr_z(models_s, z, blockIdx.z);
    lr_z(info, z, blockIdx.z); lr_z(perm, z, blockIdx.z); lr_z(glen, z, blockIdx.z); lr_z(mb, z, blockIdx.z);
    lr_ransac_state *state = reinterpret_cast<lr_ransac_state *>(counters + LR_CNT_COUNT);
    const int m = m_dev ? min(*m_dev, m_max) : m_max;
    const int V = counters[vslot];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (V <= 0 || m <= 0 || state->done) return;
    const int K0 = lr_sc_head(m, V);
    if (K0 == 0) return;      // the main pass scans the list as it lies (prune stays 0)
    const size_t ms = (size_t)model_stride;
    int ps = (int)(0xffffffffu - (uint32_t)info->pilot_key);
    ps = min(max(ps, 0), V - 1);
    const lr_model12 P = lr_load_model(models + ps, ms);
    // exclusive scan of the bucket histogram: wave 0, two buckets per lane
    if (tid < 64) {
        const int h0 = info->hist[2 * tid], h1 = info->hist[2 * tid + 1];
        int inc = h0 + h1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (tid >= o) inc += v; }
        const int ex = inc - (h0 + h1);
        s_off[2 * tid] = ex; s_off[2 * tid + 1] = ex + h0;
        info->offs[2 * tid] = ex; info->offs[2 * tid + 1] = ex + h0;
        info->fill[2 * tid] = ex; info->fill[2 * tid + 1] = ex + h0;
        if (tid == 63) { s_off[LR_SC_NB] = inc; info->offs[LR_SC_NB] = inc; }
    }
    if (tid < LR_SC_NB) s_h[tid] = 0;
    __syncthreads();
    // ball around the source points: centre and half diagonal of their bounding box
    const float b0 = lr_ford_inv(info->box[0]), b1 = lr_ford_inv(info->box[1]), b2 = lr_ford_inv(info->box[2]);
    const float b3 = lr_ford_inv(info->box[3]), b4 = lr_ford_inv(info->box[4]), b5 = lr_ford_inv(info->box[5]);
    const float cx = 0.5f * (b0 + b3), cy = 0.5f * (b1 + b4), cz = 0.5f * (b2 + b5);
    const float ex = b3 - b0, ey = b4 - b1, ez = b5 - b2;
    float rho = 0.5f * sqrtf(ex * ex + ey * ey + ez * ez) * 1.0001f + 1e-6f;
    if (info->bad) rho = LR_INF_F;                       // a non-finite coordinate: every model keeps the whole list (eps below is inf or NaN)
    const float amax = fmaxf(fmaxf(fmaxf(fabsf(b0), fabsf(b3)), fmaxf(fabsf(b1), fabsf(b4))), fmaxf(fabsf(b2), fabsf(b5)));
    constexpr float INVW = 1.0f / LR_SC_W, RMAX = LR_SC_NB * LR_SC_W;
    const float thr = sqrtf(thr2);
    const float pt1 = fabsf(P.tx) + fabsf(P.ty) + fabsf(P.tz);
    for (int v = tid; v < V; v += 1024) {
        const lr_model12 M = lr_load_model(models + v, ms);
        const float d00 = M.r00 - P.r00, d01 = M.r01 - P.r01, d02 = M.r02 - P.r02, d10 = M.r10 - P.r10, d11 = M.r11 - P.r11, d12 = M.r12 - P.r12,
                    d20 = M.r20 - P.r20, d21 = M.r21 - P.r21, d22 = M.r22 - P.r22;
        const float F = sqrtf(d00 * d00 + d01 * d01 + d02 * d02 + d10 * d10 + d11 * d11 + d12 * d12 + d20 * d20 + d21 * d21 + d22 * d22);
        const float ux = d00 * cx + d01 * cy + d02 * cz + (M.tx - P.tx), uy = d10 * cx + d11 * cy + d12 * cz + (M.ty - P.ty),
                    uz = d20 * cx + d21 * cy + d22 * cz + (M.tz - P.tz);
        const float eps = (F * rho + sqrtf(ux * ux + uy * uy + uz * uz)) * 1.001f;
        const float slack = 0.01f + 4e-6f * (3.0f * amax + pt1 + fabsf(M.tx) + fabsf(M.ty) + fabsf(M.tz));
        const float cut = thr + eps + slack;
        int bv = LR_SC_NB - 1;
        if (cut < RMAX) bv = min((int)(cut * INVW), LR_SC_NB - 1);      // (NaN, inf: everything)
        mb[v] = (uint8_t)bv;
        atomicAdd(&s_h[LR_SC_NB - 1 - bv], 1);      // longest reach first
    }
    __syncthreads();
    if (tid < 64) {
        const int h0 = s_h[2 * tid], h1 = s_h[2 * tid + 1];
        int inc = h0 + h1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (tid >= o) inc += v; }
        s_fill[2 * tid] = inc - (h0 + h1); s_fill[2 * tid + 1] = inc - h1;
    }
    __syncthreads();
    // the models in that order: a copy (the main pass then reads a group's 64 models with coalesced loads) and the way back to their slots
    for (int v = tid; v < V; v += 1024) {
        const int pos = atomicAdd(&s_fill[LR_SC_NB - 1 - (int)mb[v]], 1);
        perm[pos] = v;
#pragma unroll
        for (int k = 0; k < 12; ++k) models_s[(size_t)k * ms + pos] = models[(size_t)k * ms + v];
    }
    __syncthreads();
    const int L = m - K0;
    const int hb = (V + 63) >> 6;
    unsigned long long ev = 0;
    for (int g = tid; g < hb; g += 1024) {
        const int len = (s_off[(int)mb[perm[g * 64]] + 1] + 1) & ~1;       // the group's first model reaches furthest
        glen[g] = len;
        ev += (unsigned long long)min(len, L) * (unsigned long long)min(64, V - g * 64);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) ev += __shfl_xor(ev, o);
    if (lane == 0) s_ev[wave] = ev;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; ++w) ev += s_ev[w];
        state->evals += ev + (unsigned long long)V * (unsigned long long)K0;
        state->evals_full += (unsigned long long)V * (unsigned long long)m;
        info->prune = 1; info->n_sorted = L;
    }
}

// order pass, part 3 (a thread per correspondence): the records behind the head copied in bucket order (within a bucket in any order:
// the sums are integers).  A block reserves its share of every bucket with one atomic, its threads take positions inside that.
__global__ void __launch_bounds__(256)
ransac_scatter_kernel(const float *__restrict__ corr8, float *__restrict__ corr8s, int m_max, const int32_t *__restrict__ m_dev,
                      const int32_t *__restrict__ counters, lr_score_info *__restrict__ info, const uint8_t *__restrict__ cb, int vslot, lr_zargs z)
{
    __shared__ int s_h[LR_SC_NB], s_base[LR_SC_NB];
    lr_z(corr8, z, blockIdx.z); lr_z(corr8s, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(counters, z, blockIdx.z); lr_z(info, z, blockIdx.z); lr_z(cb, z, blockIdx.z);
    const int m = m_dev ? min(*m_dev, m_max) : m_max;
    const int V = counters[vslot];
    const int tid = threadIdx.x;
    const int c = blockIdx.x * 256 + tid;
    if (V <= 0 || blockIdx.x * 256 >= m || reinterpret_cast<const lr_ransac_state *>(counters + LR_CNT_COUNT)->done) return;
    const int K0 = lr_sc_head(m, V);
    if (K0 == 0) return;
    if (tid < LR_SC_NB) s_h[tid] = 0;
    __syncthreads();
    const bool live = c >= K0 && c < m;
    const int b = live ? (int)cb[c] : 0;
    int rank = 0;
    if (live) rank = atomicAdd(&s_h[b], 1);
    __syncthreads();
    if (tid < LR_SC_NB && s_h[tid]) s_base[tid] = atomicAdd(&info->fill[tid], s_h[tid]);
    __syncthreads();
    if (live) {
        const int pos = s_base[b] + rank;
#pragma unroll
        for (int k = 0; k < 6; ++k) corr8s[lr_corr_at(pos, k)] = corr8[lr_corr_at(c, k)];
    }
    const int L = m - K0;
    if (blockIdx.x == 0 && tid == 0 && (L & 1)) {      // an odd list ends with a record that is nobody's inlier
#pragma unroll
        for (int k = 0; k < 6; ++k) corr8s[lr_corr_at(L, k)] = k < 3 ? 0.0f : 3.0e38f;
    }
}

// ------------------------------------------------------------------ select
// best = more inliers, then lower fixed-point error, then lower hypothesis id; its fp64 model was kept by gen
__device__ __forceinline__ bool better(uint32_t c, unsigned long long q, int h, uint32_t bc, unsigned long long bq, int bh,
                                       uint32_t msac_T)
{
    if (msac_T == 0u) return c > bc || (c == bc && (q < bq || (q == bq && h < bh)));
    // MSAC: larger sum over inliers of (thr2 - d^2) in the fixed point of the error sum; a model without inliers never wins
    if (c == 0u) return false;
    if (bc == 0u) return true;
    const long long k = (long long)c * (long long)msac_T - (long long)q, bk = (long long)bc * (long long)msac_T - (long long)bq;
    return k > bk || (k == bk && h < bh);
}

// re-design of the SPRT for the next batch: eps follows the best model (its inlier count `nc` of `mm` correspondences), delta the
// models rejected so far
__device__ __forceinline__ void lr_sprt_redesign(lr_ransac_state *state, uint32_t nc, int mm)
{
    double eps = state->sprt_eps > 0.0 ? state->sprt_eps : LR_SPRT_EPS0, delta = state->sprt_delta > 0.0 ? state->sprt_delta : LR_SPRT_DELTA0;
    if (nc > 0) { const double e = (double)nc / (double)mm; if (e > eps && e < 1.0) eps = e; }
    if (state->rej_pts > 0) {
        const double d = (double)state->rej_inl / (double)state->rej_pts;
        if (d > 0.0 && d < 0.9 * eps && fabs(d - delta) > 0.05 * delta) delta = d;
    }
    if (!(delta < 0.9 * eps)) delta = 0.9 * eps * 0.5;
    state->sprt_eps = eps; state->sprt_delta = delta;
}

// best of this batch -> merged into the running state; confidence test; outputs rewritten from the state every batch
__global__ void __launch_bounds__(1024)
ransac_final_kernel(const uint32_t *__restrict__ score_cnt, const unsigned long long *__restrict__ score_ssq,
                    const int32_t *__restrict__ model_h, const double *__restrict__ models64,
                    int32_t *__restrict__ counters, int m_max, const int32_t *__restrict__ m_dev, lr_ransac_params p, int h_end,
                    double *__restrict__ T_out, lr_ransac_result *__restrict__ res, int vslot, int32_t *__restrict__ lo_ctl_words, lr_zargs z)
{
    __shared__ unsigned long long s_q[16];
    // the control block of the local optimisation's helper blocks starts from zero at every launch that may use it (before any way out
    // of this kernel: the launch that follows runs whatever happens here)
    if (lo_ctl_words) {
        lr_z(lo_ctl_words, z, blockIdx.z);
        for (int k = threadIdx.x; k < LR_LO_CTL_BYTES / 4; k += 1024) lo_ctl_words[k] = 0;
    }
    lr_z(score_cnt, z, blockIdx.z); lr_z(score_ssq, z, blockIdx.z); lr_z(model_h, z, blockIdx.z); lr_z(models64, z, blockIdx.z); lr_z(counters, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(T_out, z, blockIdx.z); lr_z(res, z, blockIdx.z);
    __shared__ uint32_t s_c[16];
    __shared__ int s_h[16], s_s[16];
    lr_ransac_state *state = reinterpret_cast<lr_ransac_state *>(counters + LR_CNT_COUNT);
    if (state->done) return;
    const uint32_t msac_T = p.scoring == 1 ? (uint32_t)(p.thr2 * 1048576.0f) : 0u;
    const int V = counters[vslot];
    uint32_t bc = 0; unsigned long long bq = ~0ull; int bh = 0x7fffffff, bs = -1;
    for (int s = threadIdx.x; s < V; s += 1024) {
        uint32_t c = score_cnt[s];
        if (c == 0) continue;
        unsigned long long q = score_ssq[s];
        int h = model_h[s];
        if (better(c, q, h, bc, bq, bh, msac_T)) { bc = c; bq = q; bh = h; bs = s; }
    }
#pragma unroll
    for (int mk = 32; mk >= 1; mk >>= 1) {
        uint32_t oc = (uint32_t)__shfl_xor((int)bc, mk);
        unsigned long long oq = __shfl_xor(bq, mk);
        int oh = __shfl_xor(bh, mk), os = __shfl_xor(bs, mk);
        if (better(oc, oq, oh, bc, bq, bh, msac_T)) { bc = oc; bq = oq; bh = oh; bs = os; }
    }
    if ((threadIdx.x & 63) == 0) { s_c[threadIdx.x >> 6] = bc; s_q[threadIdx.x >> 6] = bq; s_h[threadIdx.x >> 6] = bh; s_s[threadIdx.x >> 6] = bs; }
    __syncthreads();
    if (threadIdx.x >= 16) return;
    for (int w = 1; w < 16; ++w)
        if (better(s_c[w], s_q[w], s_h[w], bc, bq, bh, msac_T)) { bc = s_c[w]; bq = s_q[w]; bh = s_h[w]; bs = s_s[w]; }
    // all 16 lanes hold the batch winner; merge it into the state (lane k moves T[k])
    const int k = threadIdx.x;
    const uint32_t oc = state->cnt; const unsigned long long oq = state->ssq; const int oh = state->h;
    const bool take = bc > 0 && bs >= 0 && (oc == 0 || better(bc, bq, bh, oc, oq, oh, msac_T));
    double v = (k % 5 == 0) ? 1.0 : 0.0;
    if (k < 12) {
        if (take) { v = models64[(size_t)bs * 12 + k]; state->T[k] = v; }
        else if (oc > 0) v = state->T[k];
    }
    T_out[k] = v;
    if (k == 0) {
        const uint32_t nc = take ? bc : oc; const unsigned long long nq = take ? bq : oq; const int nh = take ? bh : oh;
        state->cnt = nc; state->ssq = nq; state->h = nh;
        state->n_valid += V; state->n_ids = h_end;
        counters[LR_CNT_NVALID] = 0;                       // the next batch appends from slot 0
        counters[LR_CNT_NVALID2] = 0;
        // with local optimisation a new best model is optimised first (ransac_lo_kernel, next on the stream); the exit test and the
        // re-design of the SPRT run there, on the optimised model
        const bool to_lo = take && p.local_opt == 1 && state->lo_calls < p.lo_max_calls;
        if (to_lo) state->lo_pending = 1;
        if (p.use_elc == 2 && !to_lo) lr_sprt_redesign(state, nc, m_dev ? min(*m_dev, m_max) : m_max);
        if (!to_lo && p.confidence > 0.0f && p.confidence < 1.0f && nc > 0) {
            // exit rule of Open3D's RANSAC / GC-RANSAC at batch granularity: stop once h_end >= log(1-conf)/log(1-(inl/M)^n)
            const int m = m_dev ? min(*m_dev, m_max) : m_max;
            const double f = (double)nc / (double)m;
            double fn = f;
            for (int q = 1; q < p.sample_size; ++q) fn = fn * f;
            const double kk = lr_det_log(1.0 - (double)p.confidence) / lr_det_log(1.0 - fn);
            if ((double)h_end >= kk && h_end >= p.min_iters) state->done = 1;
        }
        lr_ransac_result r;
        r.best_h = nc > 0 ? nh : -1; r.best_count = nc; r.pad0 = (uint32_t)state->lo_timeouts; r.best_ssq = nc > 0 ? nq : 0;
        r.n_valid = state->n_valid; r.n_ids = h_end;
        *res = r;
    }
}


// ------------------------------------------------------------------ local optimisation (GC-RANSAC, --GC_LO) + final polish
// One block of 1024 threads per pair; see oracle/oracle.c (lo_optimise / lo_polish) for the algorithm and its sources.
//   mode 0 (after a batch whose winner became the best model): <= LO_ROUNDS rounds of { inlier list of the model; LO_TRIALS
//          least-squares fits on LO_SAMPLE inliers each (all of them when there are no more: one fit); every fit scored over
//          ALL correspondences by 1024 / trials threads; the best replaces the model if strictly better, else stop }, then the
//          confidence test on the optimised model
//   mode 1 (once, after the last batch): iterated least squares over all inliers
// All sums that decide anything are integers (order independent); the fp64 moments of the all-inlier fits run as 1024 strided
// partials + a fixed halving tree, the order oracle.c reproduces, so the models agree bit for bit.
#define LO_ROUNDS 10
#define LO_TRIALS 20
#define LO_SAMPLE 21
#define LO_POLISH 10
#define LO_THREADS 1024

#define LO_MAXIT 32              // records (two correspondences) per thread and pass of the list builder
struct lo_shared {
    double T[LO_TRIALS][12];         // candidate models of the round (fp64)
    __attribute__((aligned(16))) float Rt[LO_TRIALS][12];         // ... rounded for the scoring arithmetic
    float pts[LO_TRIALS][LO_SAMPLE][6];     // the sampled correspondences (fetched by one lane each, summed by one thread per trial)
    unsigned cnt[LO_TRIALS];
    unsigned long long ssq[LO_TRIALS];
    double red[4][LO_THREADS];       // block reduction of the fp64 moments, four components at a time
    double mom[16];
    double curT[12];                 // model under optimisation
    unsigned long long curq;
    unsigned curc;
    int wcnt[LO_MAXIT][LO_THREADS / 64];    // inliers per (step, wave) of a list-builder pass, then their offsets in the list
    int wsum[16];
    int nI, flag;
    // near-inlier bound of a round (see lo_build_list): bounding box of the source points, how many correspondences lie within
    // thr + LO_NEAR_CAP of the model under optimisation, whether the round's trial models may be scored over those alone
    float box[6];
    int box_state;                   // 0: not computed yet; 1: valid; -1: a non-finite coordinate (no bound)
    int nNear, near_ok;
    double eps[LO_TRIALS];
    double baseT[12];                // the model near8 was copied around
};

__device__ __forceinline__ float lo_d2(const float *Rt, float px, float py, float pz, float qx, float qy, float qz)
{
    const float x = __builtin_fmaf(Rt[0], px, __builtin_fmaf(Rt[1], py, __builtin_fmaf(Rt[2], pz, Rt[3])));
    const float y = __builtin_fmaf(Rt[4], px, __builtin_fmaf(Rt[5], py, __builtin_fmaf(Rt[6], pz, Rt[7])));
    const float z = __builtin_fmaf(Rt[8], px, __builtin_fmaf(Rt[9], py, __builtin_fmaf(Rt[10], pz, Rt[11])));
    const float dx = x - qx, dy = y - qy, dz = z - qz;
    return __builtin_fmaf(dx, dx, __builtin_fmaf(dy, dy, dz * dz));
}

// inliers of sh.curT over the m correspondences, in index order -> list[0 .. sh.nI).
// A thread tests the two correspondences of one 64-byte record per step (three 16-byte loads, the packed arithmetic of the
// scoring kernel: same fma order per component).  A pass covers LO_MAXIT * LO_THREADS records: every thread runs its LO_MAXIT steps
// back to back (independent loads, no barrier in between), keeping the outcomes as a bit mask; the per-(step, wave) counts go to
// LDS, one block-wide exclusive scan turns them into list offsets, and the threads write their inliers.  Four barriers per pass.
//
// NEAR (OUT = 1; the rounds of mode 0): the trial models of a round are least-squares fits over subsets of the current model's inliers, i.e.
// close to it.  For two models t, b and a correspondence (p, q):  |T_t p - q| >= |T_b p - q| - |T_t p - T_b p| >= r_b - eps_t,
//     eps_t = |R_t - R_b|_F rho + |(R_t - R_b) c + t_t - t_b|          (c, rho: centre and half diagonal of the source points' bounding box)
// so (p, q) can be an inlier of t only if r_b < thr + eps_t -- the bound of the pilot-ordered scoring (ransac_order_kernel) with the model
// the optimisation started from as the pilot.  With OUT = 1 the pass copies the correspondences with r_b < (1 + LO_NEAR_CAP) thr, IN INDEX
// ORDER, to `near8` (corr8's record layout) instead of listing the inliers, and reduces the bounding box.  Everything a round does then
// runs over near8 alone -- about a quarter of the list on the balanced-set surrogates -- as long as every model involved stays within
// eps + slack <= LO_NEAR_CAP thr of the base model (checked in fp64 after the fits; otherwise the round scores everything and the next
// one starts from a fresh copy): the inlier list (the same call with OUT = 0 over near8: the same correspondences in the same order, so
// the samples of the trials are the oracle's), the 21-point fits, and the scoring of the 20 trials (113 -> ~30 us per round).  Counts and
// error sums are integer sums over inliers, which all lie in near8: the same integers as a scan of everything (tests/test_gpu_gc.py,
// tools/soak_gc.py against the oracle, which scores all).
#define LO_NEAR_CAP 0.5f
template <int OUT>
__device__ void lo_build_list(lo_shared &sh, const float *__restrict__ corr8, int m, float thr2, int32_t *__restrict__ list, float *__restrict__ near8 = nullptr)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool want_box = OUT == 1 && sh.box_state == 0;
    float bmn[3] = { 3.0e38f, 3.0e38f, 3.0e38f }, bmx[3] = { -3.0e38f, -3.0e38f, -3.0e38f };
    bool bad = false;
    const f32x2 R00 = { (float)sh.curT[0], (float)sh.curT[0] }, R01 = { (float)sh.curT[1], (float)sh.curT[1] }, R02 = { (float)sh.curT[2], (float)sh.curT[2] },
                TX = { (float)sh.curT[3], (float)sh.curT[3] };
    const f32x2 R10 = { (float)sh.curT[4], (float)sh.curT[4] }, R11 = { (float)sh.curT[5], (float)sh.curT[5] }, R12 = { (float)sh.curT[6], (float)sh.curT[6] },
                TY = { (float)sh.curT[7], (float)sh.curT[7] };
    const f32x2 R20 = { (float)sh.curT[8], (float)sh.curT[8] }, R21 = { (float)sh.curT[9], (float)sh.curT[9] }, R22 = { (float)sh.curT[10], (float)sh.curT[10] },
                TZ = { (float)sh.curT[11], (float)sh.curT[11] };
    const int nrec = (m + 1) >> 1;
    const unsigned long long lt = (1ull << lane) - 1ull;
    int total = 0;
    for (int sbase = 0; sbase < nrec; sbase += LO_MAXIT * LO_THREADS) {
        const int nit = min(LO_MAXIT, (nrec - sbase + LO_THREADS - 1) / LO_THREADS);
        unsigned long long bits = 0ull;          // bit 2 it + b: correspondence b of the record of step it
#pragma unroll 2
        for (int it = 0; it < nit; ++it) {
            const int r = sbase + it * LO_THREADS + tid;
            const f32x4 *rec = reinterpret_cast<const f32x4 *>(corr8) + (size_t)min(r, nrec - 1) * 4;
            const f32x4 A = rec[0], B = rec[1], C = rec[2];
            const f32x2 px = { A.x, A.y }, py = { A.z, A.w }, pz = { B.x, B.y }, qx = { B.z, B.w }, qy = { C.x, C.y }, qz = { C.z, C.w };
            const f32x2 x = __builtin_elementwise_fma(R00, px, __builtin_elementwise_fma(R01, py, __builtin_elementwise_fma(R02, pz, TX)));
            const f32x2 y = __builtin_elementwise_fma(R10, px, __builtin_elementwise_fma(R11, py, __builtin_elementwise_fma(R12, pz, TY)));
            const f32x2 z = __builtin_elementwise_fma(R20, px, __builtin_elementwise_fma(R21, py, __builtin_elementwise_fma(R22, pz, TZ)));
            const f32x2 dx = x - qx, dy = y - qy, dz = z - qz;
            const f32x2 d2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));
            const bool in0 = r < nrec && d2.x < thr2, in1 = r < nrec && 2 * r + 1 < m && d2.y < thr2;
            bits |= (unsigned long long)((in0 ? 1u : 0u) | (in1 ? 2u : 0u)) << (2 * it);
            if (want_box && r < nrec) {
                bmn[0] = fminf(bmn[0], A.x); bmx[0] = fmaxf(bmx[0], A.x); bmn[1] = fminf(bmn[1], A.z); bmx[1] = fmaxf(bmx[1], A.z);
                bmn[2] = fminf(bmn[2], B.x); bmx[2] = fmaxf(bmx[2], B.x);
                const float s0 = A.x + A.z + B.x;
                bad = bad || !(s0 - s0 == 0.0f);
                if (2 * r + 1 < m) {
                    bmn[0] = fminf(bmn[0], A.y); bmx[0] = fmaxf(bmx[0], A.y); bmn[1] = fminf(bmn[1], A.w); bmx[1] = fmaxf(bmx[1], A.w);
                    bmn[2] = fminf(bmn[2], B.y); bmx[2] = fmaxf(bmx[2], B.y);
                    const float s1 = A.y + A.w + B.y;
                    bad = bad || !(s1 - s1 == 0.0f);
                }
            }
        }
        for (int it = 0; it < nit; ++it) {
            const unsigned long long b0 = __ballot((bits >> (2 * it)) & 1ull), b1 = __ballot((bits >> (2 * it + 1)) & 1ull);
            if (lane == 0) sh.wcnt[it][wave] = __popcll(b0) + __popcll(b1);
        }
        __syncthreads();
        // exclusive scan over the entries e = step * 16 + wave (the order of the correspondence indices), one entry per thread
        int *flat = &sh.wcnt[0][0];
        const int ne = nit * (LO_THREADS / 64);
        const int v = tid < ne ? flat[tid] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc += o; }
        if (lane == 63) sh.wsum[wave] = inc;
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < LO_THREADS / 64; ++w) { const int x = sh.wsum[w]; all += x; if (w < wave) before += x; }
        if (tid < ne) flat[tid] = total + before + inc - v;
        __syncthreads();
        for (int it = 0; it < nit; ++it) {
            const bool in0 = (bits >> (2 * it)) & 1ull, in1 = (bits >> (2 * it + 1)) & 1ull;
            const unsigned long long b0 = __ballot(in0), b1 = __ballot(in1);
            const int pos = sh.wcnt[it][wave] + (int)__popcll(b0 & lt) + (int)__popcll(b1 & lt);
            const int r = sbase + it * LO_THREADS + tid;
            if constexpr (OUT == 0) {
                if (in0) list[pos] = 2 * r;
                if (in1) list[pos + (in0 ? 1 : 0)] = 2 * r + 1;
            } else if (in0 || in1) {
                // (the record again: it is in L2; keeping LO_MAXIT records in registers across the scan is not an option)
                const f32x4 *rec = reinterpret_cast<const f32x4 *>(corr8) + (size_t)r * 4;
                const f32x4 A = rec[0], B = rec[1], C = rec[2];
                if (in0) { const float w0[6] = { A.x, A.z, B.x, B.z, C.x, C.z };
#pragma unroll
                           for (int k = 0; k < 6; ++k) near8[lr_corr_at(pos, k)] = w0[k]; }
                if (in1) { const float w1[6] = { A.y, A.w, B.y, B.w, C.y, C.w }; const int p1 = pos + (in0 ? 1 : 0);
#pragma unroll
                           for (int k = 0; k < 6; ++k) near8[lr_corr_at(p1, k)] = w1[k]; }
            }
        }
        total += all;
        __syncthreads();          // (the next pass rewrites wcnt / wsum)
    }
    if (tid == 0) { if (OUT == 0) sh.nI = total; else sh.nNear = total; }
    if (want_box) {
        // bounding box of the source points (and "all finite"): wave shuffles, then one thread per component over the waves
        float *scr = reinterpret_cast<float *>(&sh.red[0][0]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { bmn[k] = fminf(bmn[k], __shfl_xor(bmn[k], o)); bmx[k] = fmaxf(bmx[k], __shfl_xor(bmx[k], o)); }
        }
        const bool wbad = __ballot(bad) != 0ull;
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k) { scr[wave * 8 + k] = bmn[k]; scr[wave * 8 + 3 + k] = bmx[k]; }
            scr[wave * 8 + 6] = wbad ? 1.0f : 0.0f;
        }
        __syncthreads();
        if (tid < 7) {
            float v = scr[tid];
            for (int w = 1; w < LO_THREADS / 64; ++w) v = tid < 3 ? fminf(v, scr[w * 8 + tid]) : fmaxf(v, scr[w * 8 + tid]);
            if (tid < 6) sh.box[tid] = v; else sh.box_state = v > 0.0f ? -1 : 1;
        }
    }
    __syncthreads();
}

// least-squares fit over the listed correspondences -> sh.T[0] / sh.Rt[0]; false when fewer than 3 points
__device__ bool lo_fit_all(lo_shared &sh, const float *__restrict__ corr8, const int32_t *__restrict__ list, int n)
{
    const int tid = threadIdx.x;
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = 0.0;
    for (int e = tid; e < n; e += LO_THREADS) {
        const int i = list[e];
        const double p[3] = { (double)corr8[lr_corr_at(i, 0)], (double)corr8[lr_corr_at(i, 1)], (double)corr8[lr_corr_at(i, 2)] };
        const double q[3] = { (double)corr8[lr_corr_at(i, 3)], (double)corr8[lr_corr_at(i, 4)], (double)corr8[lr_corr_at(i, 5)] };
        v[0] += 1.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) { v[1 + a] += p[a]; v[4 + a] += q[a]; }
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) v[7 + 3 * a + b] += p[a] * q[b];
    }
    // halving tree red[t] += red[t + s], s = 512 ... 1 (the order oracle.c reproduces): through LDS while the partners sit in
    // different waves, by shuffles inside wave 0 for s <= 32 (the same additions, no barriers)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) sh.red[k][tid] = v[4 * g + k];
        for (int sft = LO_THREADS / 2; sft >= 64; sft >>= 1) {
            __syncthreads();
            if (tid < sft) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sh.red[k][tid] += sh.red[k][tid + sft];
            }
        }
        __syncthreads();
        if (tid < 64) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double x = sh.red[k][tid];
#pragma unroll
                for (int sft = 32; sft >= 1; sft >>= 1) x = x + __shfl_down(x, sft);
                if (tid == 0) sh.mom[4 * g + k] = x;
            }
        }
    }
    __syncthreads();
    const bool ok = sh.mom[0] >= 3.0;
    if (ok && tid == 0) {
        const double nn = sh.mom[0];
        double cp[3], cq[3], H[3][3], T[16];
        for (int a = 0; a < 3; ++a) { cp[a] = sh.mom[1 + a] / nn; cq[a] = sh.mom[4 + a] / nn; }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) H[a][b] = sh.mom[7 + 3 * a + b] - (nn * cp[a]) * cq[b];
        lr_rt_from_cov(H, cp, cq, T);
        for (int k = 0; k < 12; ++k) { sh.T[0][k] = T[k]; sh.Rt[0][k] = (float)T[k]; }
    }
    __syncthreads();
    return ok;
}

// score sh.Rt[0 .. ntrial) over all m correspondences -> sh.cnt / sh.ssq (integer atomics in LDS): three forms.
// One model (the polish, or a round with at most LO_SAMPLE inliers): one thread per 64-byte record (two correspondences), 32-bit sums
// per thread (the caller guarantees that a thread's share of the correspondences cannot overflow them), one wave reduction and one
// LDS atomic per wave at the end.
__device__ void lo_score_one(lo_shared &sh, const float *__restrict__ corr8, int m, float thr2)
{
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid < LO_TRIALS) { sh.cnt[tid] = 0u; sh.ssq[tid] = 0ull; }
    __syncthreads();
    const float *Rt = sh.Rt[0];
    const f32x2 R00 = { Rt[0], Rt[0] }, R01 = { Rt[1], Rt[1] }, R02 = { Rt[2], Rt[2] }, TX = { Rt[3], Rt[3] };
    const f32x2 R10 = { Rt[4], Rt[4] }, R11 = { Rt[5], Rt[5] }, R12 = { Rt[6], Rt[6] }, TY = { Rt[7], Rt[7] };
    const f32x2 R20 = { Rt[8], Rt[8] }, R21 = { Rt[9], Rt[9] }, R22 = { Rt[10], Rt[10] }, TZ = { Rt[11], Rt[11] };
    const int nrec = (m + 1) >> 1;
    uint32_t c = 0u, q = 0u;
    // the record walk of lo_build_list (three 16-byte loads per 64-byte record, both correspondences in one packed chain; the scalar
    // form -- 24 strided 4-byte loads per step -- kept the texture addresser busy for 76 us per call at 30k correspondences)
#pragma unroll 4
    for (int r = tid; r < nrec; r += LO_THREADS) {
        const f32x4 *rec = reinterpret_cast<const f32x4 *>(corr8) + (size_t)r * 4;
        const f32x4 A = rec[0], B = rec[1], C = rec[2];
        const f32x2 px = { A.x, A.y }, py = { A.z, A.w }, pz = { B.x, B.y }, qx = { B.z, B.w }, qy = { C.x, C.y }, qz = { C.z, C.w };
        const f32x2 x = __builtin_elementwise_fma(R00, px, __builtin_elementwise_fma(R01, py, __builtin_elementwise_fma(R02, pz, TX)));
        const f32x2 y = __builtin_elementwise_fma(R10, px, __builtin_elementwise_fma(R11, py, __builtin_elementwise_fma(R12, pz, TY)));
        const f32x2 z = __builtin_elementwise_fma(R20, px, __builtin_elementwise_fma(R21, py, __builtin_elementwise_fma(R22, pz, TZ)));
        const f32x2 dx = x - qx, dy = y - qy, dz = z - qz;
        const f32x2 d2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));
        const bool in0 = d2.x < thr2, in1 = 2 * r + 1 < m && d2.y < thr2;
        c += (in0 ? 1u : 0u) + (in1 ? 1u : 0u);
        q += (in0 ? (uint32_t)(d2.x * 1048576.0f) : 0u) + (in1 ? (uint32_t)(d2.y * 1048576.0f) : 0u);
    }
    unsigned long long qq = q;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { c += (uint32_t)__shfl_xor((int)c, o); qq += __shfl_xor(qq, o); }
    if (lane == 0 && c) { atomicAdd(&sh.cnt[0], c); atomicAdd(&sh.ssq[0], qq); }
    __syncthreads();
}

// all LO_TRIALS models of a round: the arithmetic of ransac_score_kernel (one LANE per model, two correspondences per packed
// step), with three correspondence streams side by side in a wave -- lanes [20 s, 20 s + 20) apply the 20 models to the records
// 3 w + s, 3 w + s + 48, ... of wave w -- so 60 of the 64 lanes work and a wave's three 64-byte records come in as three
// broadcast global_load_dwordx4 per lane.  12 coefficient + 12 record registers, sums per lane, 48 LDS atomics per model at the end.
__device__ void lo_score_lanes_range(lo_shared &sh, const float *__restrict__ corr8, int m, float thr2, int rec_begin, int rec_end)
{
    // (adds to sh.cnt / sh.ssq: the caller zeroes them; records [rec_begin, rec_end))
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NS = 64 / LO_TRIALS;                 // streams per wave (3)
    constexpr int STRIDE = NS * (LO_THREADS / 64);     // records between two steps of a stream (48)
    const int t = lane % LO_TRIALS, st = lane / LO_TRIALS;
    const bool act = st < NS;
    const f32x4 *rp = reinterpret_cast<const f32x4 *>(sh.Rt[t]);
    const f32x4 r0 = rp[0], r1 = rp[1], r2 = rp[2];
    const f32x2 R00 = { r0.x, r0.x }, R01 = { r0.y, r0.y }, R02 = { r0.z, r0.z }, TX = { r0.w, r0.w };
    const f32x2 R10 = { r1.x, r1.x }, R11 = { r1.y, r1.y }, R12 = { r1.z, r1.z }, TY = { r1.w, r1.w };
    const f32x2 R20 = { r2.x, r2.x }, R21 = { r2.y, r2.y }, R22 = { r2.z, r2.z }, TZ = { r2.w, r2.w };
    const f32x2 SC = { 1048576.0f, 1048576.0f };
    uint32_t c = 0u;
    unsigned long long q = 0ull;
    const int first = rec_begin + wave * NS + (act ? st : 0);
    const int span = rec_end - rec_begin - wave * NS;
    const int steps = span > 0 ? (span + STRIDE - 1) / STRIDE : 0;        // (wave-uniform: the steps of the wave's first stream)
#pragma unroll 2
    for (int k = 0; k < steps; ++k) {
        const int r = first + k * STRIDE;
        const bool live = act && r < rec_end;
        const f32x4 *rec = reinterpret_cast<const f32x4 *>(corr8) + (size_t)(live ? r : 0) * 4;
        const f32x4 A = rec[0], B = rec[1], C = rec[2];
        const f32x2 px = { A.x, A.y }, py = { A.z, A.w }, pz = { B.x, B.y }, qx = { B.z, B.w }, qy = { C.x, C.y }, qz = { C.z, C.w };
        const f32x2 x = __builtin_elementwise_fma(R00, px, __builtin_elementwise_fma(R01, py, __builtin_elementwise_fma(R02, pz, TX)));
        const f32x2 y = __builtin_elementwise_fma(R10, px, __builtin_elementwise_fma(R11, py, __builtin_elementwise_fma(R12, pz, TY)));
        const f32x2 z = __builtin_elementwise_fma(R20, px, __builtin_elementwise_fma(R21, py, __builtin_elementwise_fma(R22, pz, TZ)));
        const f32x2 dx = x - qx, dy = y - qy, dz = z - qz;
        const f32x2 d2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));
        const f32x2 fx = d2 * SC;
        const bool in0 = live && d2.x < thr2, in1 = live && 2 * r + 1 < m && d2.y < thr2;
        c += (in0 ? 1u : 0u) + (in1 ? 1u : 0u);
        q += (unsigned long long)((in0 ? (uint32_t)fx.x : 0u)) + (unsigned long long)((in1 ? (uint32_t)fx.y : 0u));
    }
    if (act && c) { atomicAdd(&sh.cnt[t], c); atomicAdd(&sh.ssq[t], q); }
    __syncthreads();
}
__device__ void lo_score_lanes(lo_shared &sh, const float *__restrict__ corr8, int m, float thr2)
{
    if (threadIdx.x < LO_TRIALS) { sh.cnt[threadIdx.x] = 0u; sh.ssq[threadIdx.x] = 0ull; }
    __syncthreads();
    lo_score_lanes_range(sh, corr8, m, thr2, 0, (m + 1) >> 1);
}

// ---- helper blocks (single-pair and small-batch calls: the GPU is idle while one block per pair optimises its model).
// The scoring of a round's 20 models over all correspondences is 2/3 of a round and embarrassingly parallel.  The master block (block 0
// of the pair) PUBLISHES it as a job -- the 20 rounded models, the chunk count -- in a control block in device memory; every block of
// the pair's group, the master included, claims chunks of LO_CHUNK_REC records from the job's counter, scores them with
// lo_score_lanes_range, adds its integer sums to the job's totals and reports the chunks it did.  The master waits only for chunks that
// were CLAIMED, i.e. for blocks that are running: a helper that was never scheduled (other kernels hold the CUs) costs nothing, and no
// block ever waits for a block that is not running -- there is no grid barrier.  Every job has its own slot (counters are never reused
// within a launch, so a slow helper cannot touch a later job); the slots are zeroed by ransac_final_kernel, which precedes every launch.
// Release / acquire follow MI355X_MICROARCH.md ("inter-workgroup visibility"): stores, vmcnt(0), barrier, agent-scope release, flag;
// agent-scope acquire by one lane, vmcnt(0), barrier, plain loads.  Spins are bounded by the wall clock; a master that gives up
// recomputes the job alone.
#define LO_CHUNK_REC 384
#define LO_JOBS 16
struct lr_lo_job { int32_t next_chunk, done_chunks, nchunks, pad0;
                   float Rt[LO_TRIALS][12]; unsigned cnt[LO_TRIALS]; unsigned long long ssq[LO_TRIALS]; };
struct lr_lo_ctl { int32_t phase, pad[3]; lr_lo_job job[LO_JOBS]; };      // phase: 0 nothing yet; k > 0: job k - 1 is published; -1: the master is done
static_assert(sizeof(lr_lo_ctl) <= LR_LO_CTL_BYTES, "lr_lo_ctl does not fit its scratch block");

// this block's share of job `jb` (models already in sh.Rt): claims chunks until none is left, then adds its sums to the job's totals
__device__ void lo_job_work(lo_shared &sh, lr_lo_job *jb, const float *__restrict__ corr8, int m, float thr2)
{
    const int tid = threadIdx.x;
    const int nrec = (m + 1) >> 1;
    const int nchunks = (nrec + LO_CHUNK_REC - 1) / LO_CHUNK_REC;
    if (tid < LO_TRIALS) { sh.cnt[tid] = 0u; sh.ssq[tid] = 0ull; }
    __syncthreads();
    int mine = 0;
    for (;;) {
        if (tid == 0) sh.flag = __hip_atomic_fetch_add(&jb->next_chunk, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int c = sh.flag;
        __syncthreads();
        if (c >= nchunks) break;
        lo_score_lanes_range(sh, corr8, m, thr2, c * LO_CHUNK_REC, min(nrec, (c + 1) * LO_CHUNK_REC));
        mine += 1;
    }
    if (mine > 0) {
        if (tid < LO_TRIALS && sh.cnt[tid]) {
            __hip_atomic_fetch_add(&jb->cnt[tid], sh.cnt[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&jb->ssq[tid], sh.ssq[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(&jb->done_chunks, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
}

// master: score sh.Rt[0 .. LO_TRIALS) over all correspondences with whoever helps -> sh.cnt / sh.ssq
__device__ void lo_score_shared(lo_shared &sh, lr_lo_ctl *ctl, int job, const float *__restrict__ corr8, int m, float thr2, int32_t *timeouts)
{
    const int tid = threadIdx.x;
    lr_lo_job *jb = &ctl->job[job];
    const int nrec = (m + 1) >> 1;
    const int nchunks = (nrec + LO_CHUNK_REC - 1) / LO_CHUNK_REC;
    for (int k = tid; k < LO_TRIALS * 12; k += LO_THREADS) jb->Rt[k / 12][k % 12] = sh.Rt[k / 12][k % 12];
    if (tid == 0) jb->nchunks = nchunks;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(&ctl->phase, job + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    lo_job_work(sh, jb, corr8, m, thr2);
    // wait for the chunks other blocks claimed (they are running: this ends), bounded by the wall clock
    if (tid == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int ok = 1;
        while (__hip_atomic_load(&jb->done_chunks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nchunks) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) { ok = 0; break; }      // 0.2 s at 100 MHz
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        sh.flag = ok;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int ok = sh.flag;
    __syncthreads();
    if (!ok) {          // (never observed; the sums of the job are abandoned, the event is counted: lr_ransac_result.pad0)
        if (tid == 0 && timeouts) __hip_atomic_fetch_add(timeouts, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lo_score_lanes(sh, corr8, m, thr2);
        return;
    }
    if (tid < LO_TRIALS) {
        sh.cnt[tid] = __hip_atomic_load(&jb->cnt[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh.ssq[tid] = __hip_atomic_load(&jb->ssq[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
}

// helper block: serves the jobs the master publishes until it says it is done (or nothing happens for 0.2 s)
__device__ void lo_helper_loop(lo_shared &sh, lr_lo_ctl *ctl, const float *__restrict__ corr8, int m, float thr2, int32_t *timeouts)
{
    const int tid = threadIdx.x;
    int seen = 0;
    for (;;) {
        if (tid == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            int p;
            while ((p = __hip_atomic_load(&ctl->phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == seen) {
                __builtin_amdgcn_s_sleep(8);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) { p = -1; __hip_atomic_fetch_add(timeouts, 0x10000, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            sh.flag = p;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int p = sh.flag;
        __syncthreads();
        if (p <= 0) return;                       // -1: done (0 cannot be observed as a change)
        seen = p;
        lr_lo_job *jb = &ctl->job[p - 1];
        for (int k = tid; k < LO_TRIALS * 12; k += LO_THREADS) sh.Rt[k / 12][k % 12] = jb->Rt[k / 12][k % 12];
        __syncthreads();
        lo_job_work(sh, jb, corr8, m, thr2);      // (always the full list: the near-inlier copy only exists in launches without helper blocks)
    }
}

// any number of models and any threshold (64-bit sums): LO_THREADS / ntrial threads per model, each striding over the correspondences
__device__ void lo_score_wide(lo_shared &sh, const float *__restrict__ corr8, int m, float thr2, int ntrial)
{
    const int tid = threadIdx.x;
    if (tid < LO_TRIALS) { sh.cnt[tid] = 0u; sh.ssq[tid] = 0ull; }
    __syncthreads();
    const int per = LO_THREADS / ntrial;           // threads per model
    const int t = tid / per, l = tid % per;
    if (t < ntrial) {
        float Rt[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) Rt[k] = sh.Rt[t][k];
        unsigned c = 0; unsigned long long q = 0;
        for (int i = l; i < m; i += per) {
            const float d2 = lo_d2(Rt, corr8[lr_corr_at(i, 0)], corr8[lr_corr_at(i, 1)], corr8[lr_corr_at(i, 2)], corr8[lr_corr_at(i, 3)],
                                   corr8[lr_corr_at(i, 4)], corr8[lr_corr_at(i, 5)]);
            if (d2 < thr2) { c += 1u; q += (unsigned long long)(uint32_t)(d2 * 1048576.0f); }
        }
        if (c) { atomicAdd(&sh.cnt[t], c); atomicAdd(&sh.ssq[t], q); }
    }
    __syncthreads();
}

__device__ __forceinline__ void lo_score(lo_shared &sh, const float *__restrict__ corr8, int m, float thr2, int ntrial, lr_lo_ctl *ctl = nullptr, int job = -1,
                                         int32_t *timeouts = nullptr)
{
    // 32-bit per-thread error sums hold when (correspondences per thread) * thr2 * 2^20 < 2^32
    if (ntrial > 1) {      // (lanes of trials >= ntrial score stale models nobody reads)
        if (ctl && job >= 0 && job < LO_JOBS) lo_score_shared(sh, ctl, job, corr8, m, thr2, timeouts);
        else lo_score_lanes(sh, corr8, m, thr2);
        return;
    }
    const bool narrow = ((double)(m / LO_THREADS + 2)) * (double)thr2 * 1048576.0 < 4.0e9;      // (a thread's correspondences x the largest term)
    if (!narrow) lo_score_wide(sh, corr8, m, thr2, ntrial);
    else lo_score_one(sh, corr8, m, thr2);
}

#ifdef LR_LO_PROBE      // development build (tools/r4_loprobe.sh): 10 ns ticks the master block spends per phase, summed over all launches
__device__ unsigned long long g_lo_probe[16];
extern "C" __attribute__((visibility("default"))) int lr_debug_lo_probe(unsigned long long *out, int reset)
{
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lo_probe), sizeof(unsigned long long) * 16);
    if (reset) { unsigned long long z[16] = { 0 }; hipMemcpyToSymbol(HIP_SYMBOL(g_lo_probe), z, sizeof(z)); }
    return 0;
}
#define LO_TICK(slot) do { if (threadIdx.x == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memrealtime(); atomicAdd(&g_lo_probe[slot], now_ - tk_); tk_ = now_; } } while (0)
#define LO_COUNT(slot) do { if (threadIdx.x == 0) atomicAdd(&g_lo_probe[slot], 1ull); } while (0)
#else
#define LO_TICK(slot) do { } while (0)
#define LO_COUNT(slot) do { } while (0)
#endif

__global__ void __launch_bounds__(LO_THREADS)
ransac_lo_kernel(const float *__restrict__ corr8, int m_max, const int32_t *__restrict__ m_dev, lr_ransac_params p, int h_end, int mode,
                 int32_t *__restrict__ counters, int32_t *__restrict__ list, double *__restrict__ T_out, lr_ransac_result *__restrict__ res,
                 lr_lo_ctl *__restrict__ ctl, float *__restrict__ near8, lr_zargs z)
{
    lr_z(corr8, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(counters, z, blockIdx.z); lr_z(list, z, blockIdx.z); lr_z(T_out, z, blockIdx.z);
    lr_z(res, z, blockIdx.z); lr_z(ctl, z, blockIdx.z); lr_z(near8, z, blockIdx.z);
    __shared__ lo_shared sh;
    lr_ransac_state *state = reinterpret_cast<lr_ransac_state *>(counters + LR_CNT_COUNT);
    const int m = m_dev ? min(*m_dev, m_max) : m_max;
    const int tid = threadIdx.x;
    const bool helpers = gridDim.x > 1 && mode == 0;              // blocks 1.. of the pair's group serve the master's scoring jobs
    if (blockIdx.x > 0) {
        if (helpers && m > 0) lo_helper_loop(sh, ctl, corr8, m, p.thr2, &state->lo_timeouts);
        return;
    }
    // (every way out of the master tells the helpers: they must not wait for jobs that never come)
    auto release_helpers = [&]() { if (helpers && tid == 0) __hip_atomic_store(&ctl->phase, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    if (state->cnt == 0 || m <= 0) { release_helpers(); return; }                         // no model
    if (mode == 0 && !state->lo_pending) { release_helpers(); return; }                  // best model unchanged by the batch just merged
#ifdef LR_LO_PROBE
    unsigned long long tk_ = __builtin_amdgcn_s_memrealtime();
    const unsigned long long tk0_ = tk_;
#endif
    LO_COUNT(mode == 0 ? 9 : 10);
    const uint32_t msac_T = p.scoring == 1 ? (uint32_t)(p.thr2 * 1048576.0f) : 0u;
    if (tid < 12) sh.curT[tid] = state->T[tid];
    if (tid == 0) { sh.curc = state->cnt; sh.curq = state->ssq; }
    const int call = state->lo_calls;
    __syncthreads();
    if (mode == 0) {
        // near8: the correspondences within (1 + LO_NEAR_CAP) thr of the base model, in index order (lo_build_list<1>); while every
        // model involved stays within the cap of the base, a round runs over near8 alone
        const float thr = sqrtf(p.thr2);
        const float thr_near = thr * (1.0f + LO_NEAR_CAP), thr2_near = thr_near * thr_near * 1.000001f;
        // (short lists: nothing to gain; calls with helper blocks -- single pairs -- neither: their scoring is already spread over 16 CUs,
        // the copy would cost more than it saves: 156 -> 178 us per pair measured)
        bool near_on = near8 != nullptr && m >= 4096 && !helpers;
        if (tid == 0) { sh.box_state = 0; sh.near_ok = 0; sh.nNear = 0; }
        __syncthreads();
        bool have_near = false;
        for (int round = 0; round < p.lo_rounds; ++round) {
            LO_COUNT(8);
            const float *cc = corr8; int mm = m;          // what this round's list, fits (and, eps permitting, scoring) run over
            if (near_on) {
                if (!have_near) {
                    if (tid < 12) sh.baseT[tid] = sh.curT[tid];
                    lo_build_list<1>(sh, corr8, m, thr2_near, nullptr, near8);
                    have_near = sh.box_state == 1 && sh.nNear > 0;
                    // (a non-finite source coordinate -- no bounding box, no bound -- or an empty copy: no later round of this call can use one
                    // either; without this every remaining round re-ran the full scan + copy for nothing)
                    if (!have_near) near_on = false;
                    LO_COUNT(13);
                }
                if (have_near) { cc = near8; mm = sh.nNear; }
            }
            lo_build_list<0>(sh, cc, mm, p.thr2, list);
            LO_TICK(0);
            const int nI = sh.nI;
            if (nI <= p.sample_size) break;
            const int ntrial = nI > LO_SAMPLE ? p.lo_trials : 1;
            if (nI > LO_SAMPLE) {
                // LO_SAMPLE distinct positions per trial (word stream keyed by seed, call, round, trial; a word that repeats a position is
                // skipped): 32 lanes per trial, lane j keeps the j-th accepted position, so "already drawn?" is one ballot over the
                // trial's half of the wave instead of a loop over LDS.  Lane j then fetches its correspondence (two dependent loads,
                // all LO_TRIALS * LO_SAMPLE of them in flight at once).
                {
                    const int t = tid >> 5, j = tid & 31, half = (tid >> 5) & 1;
                    if (t < ntrial) {
                        int mine = -1, got = 0;
                        const uint64_t key = p.seed ^ 0x4c4f43414c4f5054ull;
                        for (int blk = 0; blk < 32 && got < LO_SAMPLE; ++blk) {
                            const uint64_t ctr = ((uint64_t)call << 40) | ((uint64_t)round << 32) | ((uint64_t)t << 8) | (uint64_t)blk;
                            uint32_t w[4] = { (uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u };
                            philox4x32_10(w, (uint32_t)key, (uint32_t)(key >> 32));
                            for (int k = 0; k < 4 && got < LO_SAMPLE; ++k) {
                                const int c = (int)__umulhi(w[k], (uint32_t)nI);
                                const bool dup = (uint32_t)(__ballot(mine == c) >> (32 * half)) != 0u;
                                if (!dup) { if (j == got) mine = c; ++got; }
                            }
                        }
                        for (int c = 0; got < LO_SAMPLE; ++c) {
                            const bool dup = (uint32_t)(__ballot(mine == c) >> (32 * half)) != 0u;
                            if (!dup) { if (j == got) mine = c; ++got; }
                        }
                        if (j < LO_SAMPLE) {
                            const int i = list[mine];
#pragma unroll
                            for (int a = 0; a < 6; ++a) sh.pts[t][j][a] = cc[lr_corr_at(i, a)];
                        }
                    }
                }
                __syncthreads();
                // ... and summed by one thread per trial in the order of kabsch_points_kernel / orc_kabsch_points
                if (tid < ntrial) {
                    double cp[3] = { 0, 0, 0 }, cq[3] = { 0, 0, 0 }, W = 0.0;
#pragma unroll 1
                    for (int k = 0; k < LO_SAMPLE; ++k) {      // (rolled: unrolled, the 126 values of both loops are kept live and spill)
                        W = W + 1.0;
                        for (int a = 0; a < 3; ++a) { cp[a] = cp[a] + 1.0 * (double)sh.pts[tid][k][a]; cq[a] = cq[a] + 1.0 * (double)sh.pts[tid][k][3 + a]; }
                    }
                    for (int a = 0; a < 3; ++a) { cp[a] = cp[a] / W; cq[a] = cq[a] / W; }
                    double H[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
#pragma unroll 1
                    for (int k = 0; k < LO_SAMPLE; ++k) {
                        double pc[3], qc[3];
                        for (int a = 0; a < 3; ++a) { pc[a] = (double)sh.pts[tid][k][a] - cp[a]; qc[a] = (double)sh.pts[tid][k][3 + a] - cq[a]; }
                        for (int a = 0; a < 3; ++a)
                            for (int b = 0; b < 3; ++b) H[a][b] = H[a][b] + (1.0 * pc[a]) * qc[b];
                    }
                    double T[16];
                    lr_rt_from_cov(H, cp, cq, T);
                    for (int k = 0; k < 12; ++k) { sh.T[tid][k] = T[k]; sh.Rt[tid][k] = (float)T[k]; }
                    // how far this trial can move a source point away from where the model under optimisation puts it (fp64; see lo_build_list)
                    double e = 1.0e300;
                    if (have_near) {
                        const double cx = 0.5 * ((double)sh.box[0] + (double)sh.box[3]), cy = 0.5 * ((double)sh.box[1] + (double)sh.box[4]), cz = 0.5 * ((double)sh.box[2] + (double)sh.box[5]);
                        const double ex = (double)sh.box[3] - (double)sh.box[0], ey = (double)sh.box[4] - (double)sh.box[1], ez = (double)sh.box[5] - (double)sh.box[2];
                        const double rho = 0.5 * sqrt(ex * ex + ey * ey + ez * ez) * 1.0001 + 1e-6;
                        double d[12], F2 = 0.0;
                        for (int k = 0; k < 12; ++k) d[k] = T[k] - sh.baseT[k];
                        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) F2 += d[4 * a + b] * d[4 * a + b];
                        const double ux = d[0] * cx + d[1] * cy + d[2] * cz + d[3], uy = d[4] * cx + d[5] * cy + d[6] * cz + d[7], uz = d[8] * cx + d[9] * cy + d[10] * cz + d[11];
                        const double amax = fmax(fmax(fmax(fabs((double)sh.box[0]), fabs((double)sh.box[3])), fmax(fabs((double)sh.box[1]), fabs((double)sh.box[4]))),
                                                 fmax(fabs((double)sh.box[2]), fabs((double)sh.box[5])));
                        // (the slack of the pilot-ordered scoring: 1 cm + 4e-6 of the coordinate magnitude, two orders above the fp32 rounding of both residuals)
                        const double slack = 0.01 + 4e-6 * (3.0 * amax + fabs(T[3]) + fabs(T[7]) + fabs(T[11]) + fabs(sh.baseT[3]) + fabs(sh.baseT[7]) + fabs(sh.baseT[11]));
                        e = (sqrt(F2) * rho + sqrt(ux * ux + uy * uy + uz * uz)) * 1.001 + slack;
                    }
                    sh.eps[tid] = e;          // (NaN from a degenerate fit fails the comparison below: the round scores everything)
                }
                __syncthreads();
                if (tid == 0) {
                    bool ok = have_near;
                    for (int t = 0; t < ntrial && ok; ++t) ok = sh.eps[t] <= (double)(LO_NEAR_CAP * thr);
                    sh.near_ok = ok ? 1 : 0;
                }
                __syncthreads();
            } else {
                if (tid == 0) { sh.near_ok = 0; sh.eps[0] = 1.0e300; }      // (one fit over all inliers: not bounded, scored over everything)
                if (!lo_fit_all(sh, cc, list, nI)) break;
            }
            LO_TICK(1);
            // (near_ok implies a launch without helper blocks: the near-inlier copy is never handed to a job)
            if (sh.near_ok) { LO_COUNT(12); lo_score(sh, near8, sh.nNear, p.thr2, ntrial); }
            else lo_score(sh, corr8, m, p.thr2, ntrial, helpers ? ctl : nullptr, round, &state->lo_timeouts);
            LO_TICK(2);
            if (tid == 0) {
                int bt = -1; unsigned bc = 0; unsigned long long bq = 0;
                for (int t = 0; t < ntrial; ++t) {
                    if (sh.cnt[t] == 0u) continue;
                    if (bt < 0 || better(sh.cnt[t], sh.ssq[t], t, bc, bq, bt, msac_T)) { bt = t; bc = sh.cnt[t]; bq = sh.ssq[t]; }
                }
                // strictly better than the model under optimisation
                sh.flag = (bt >= 0 && better(bc, bq, 1, sh.curc, sh.curq, 0, msac_T)) ? bt : -1;
                if (sh.flag >= 0) { sh.curc = bc; sh.curq = bq; for (int k = 0; k < 12; ++k) sh.curT[k] = sh.T[bt][k]; }
            }
            __syncthreads();
            LO_TICK(3);
            if (sh.flag < 0) break;
            // the next round may keep running over near8 only if the model it optimises lies within the cap of the base model
            // (its inliers are then all in near8); else it starts from a fresh copy around that model
            have_near = have_near && sh.eps[sh.flag] <= (double)(LO_NEAR_CAP * thr);
        }
    } else {
        for (int it = 0; it < LO_POLISH; ++it) {
            LO_COUNT(11);
            lo_build_list<false>(sh, corr8, m, p.thr2, list);
            LO_TICK(4);
            const int nI = sh.nI;
            if (nI <= p.sample_size || !lo_fit_all(sh, corr8, list, nI)) break;
            LO_TICK(5);
            lo_score(sh, corr8, m, p.thr2, 1);
            LO_TICK(6);
            if (tid == 0) {
                // a fit that loses inliers is discarded; an equal count is kept and ends the iteration; more: again
                const unsigned tc = sh.cnt[0];
                sh.flag = tc < sh.curc ? -1 : (tc == sh.curc ? 0 : 1);
                if (sh.flag >= 0) { sh.curc = tc; sh.curq = sh.ssq[0]; for (int k = 0; k < 12; ++k) sh.curT[k] = sh.T[0][k]; }
            }
            __syncthreads();
            if (sh.flag <= 0) break;
        }
    }
    __syncthreads();
    release_helpers();
    // back to the running state; outputs rewritten from it
    if (tid < 16) {
        double v = (tid % 5 == 0) ? 1.0 : 0.0;
        if (tid < 12) { v = sh.curT[tid]; state->T[tid] = v; }
        T_out[tid] = v;
    }
    if (tid == 0) {
        state->cnt = sh.curc; state->ssq = sh.curq;
        if (mode == 0) {
            state->lo_pending = 0; state->lo_calls = call + 1;
            if (p.use_elc == 2) lr_sprt_redesign(state, sh.curc, m);
            if (p.confidence > 0.0f && p.confidence < 1.0f) {
                const double f = (double)sh.curc / (double)m;
                double fn = f;
                for (int q = 1; q < p.sample_size; ++q) fn = fn * f;
                const double kk = lr_det_log(1.0 - (double)p.confidence) / lr_det_log(1.0 - fn);
                if ((double)h_end >= kk && h_end >= p.min_iters) state->done = 1;
            }
        }
        lr_ransac_result r;
        r.best_h = state->h; r.best_count = sh.curc; r.pad0 = (uint32_t)state->lo_timeouts; r.best_ssq = sh.curq;
        r.n_valid = state->n_valid; r.n_ids = state->n_ids;
        *res = r;
    }
#ifdef LR_LO_PROBE
    if (tid == 0) atomicAdd(&g_lo_probe[7], __builtin_amdgcn_s_memrealtime() - tk0_);
#endif
}

// ------------------------------------------------------------------ inlier mask (what findRigidTransform returns next to the pose)
__global__ void __launch_bounds__(256)
inlier_mask_kernel(const float *__restrict__ src, const float *__restrict__ tgt, const int32_t *__restrict__ i0, const int32_t *__restrict__ i1,
                   int m_max, const int32_t *__restrict__ m_dev, const double *__restrict__ T, float thr2, uint8_t *__restrict__ mask,
                   int32_t *__restrict__ n_inliers)
{
    const int m = m_dev ? min(*m_dev, m_max) : m_max;
    const int c = blockIdx.x * 256 + threadIdx.x;
    bool in = false;
    if (c < m) {
        float Rt[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) Rt[k] = (float)T[k];
        const int a = i0 ? i0[c] : c, b = i1 ? i1[c] : c;
        in = lo_d2(Rt, src[3 * a], src[3 * a + 1], src[3 * a + 2], tgt[3 * b], tgt[3 * b + 1], tgt[3 * b + 2]) < thr2;
        mask[c] = in ? 1 : 0;
    }
    if (n_inliers) {
        const unsigned long long bal = __ballot(in);
        if ((threadIdx.x & 63) == 0 && bal) atomicAdd(n_inliers, (int)__popcll(bal));
    }
}

int lr_inlier_mask_run(const float *src, const float *tgt, const int32_t *i0, const int32_t *i1, int m_max, const int32_t *m_dev,
                       const double *T, float thr2, uint8_t *mask, int32_t *n_inliers, hipStream_t st)
{
    if (n_inliers) LR_HIP(hipMemsetAsync(n_inliers, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(inlier_mask_kernel, dim3(lr_cdiv(m_max > 0 ? m_max : 1, 256)), dim3(256), 0, st, src, tgt, i0, i1, m_max, m_dev, T, thr2, mask, n_inliers);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

int lr_ransac_run(lr_workspace *ws, const float *corr8, int m_max, const int32_t *m_dev, const lr_ransac_params *p_in,
                  double *T_out, lr_ransac_result *res, hipStream_t st)
{
    LR_REQUIRE(p_in->scoring >= 0 && p_in->scoring <= 2, LR_EINVAL, "lr_ransac: scoring must be 0 (count, then error), 1 (MSAC) or 2 (MSAC at GC-RANSAC's truncated threshold)");
    LR_REQUIRE(p_in->lo_rounds >= 0 && p_in->lo_trials >= 0 && p_in->lo_trials <= 20 && p_in->lo_max_calls >= 0 && p_in->min_iters >= 0, LR_EINVAL,
               "lr_ransac: lo_rounds, lo_trials (<= 20), lo_max_calls and min_iters must be >= 0 (0 = default)");
    // the parameters as the kernels use them: defaults resolved, the truncated threshold applied (same text as oracle.c eff_params)
    lr_ransac_params pe = *p_in;
    if (pe.scoring == 2) { pe.thr2 = pe.thr2 * 2.25f; pe.scoring = 1; }
    if (pe.lo_rounds <= 0) pe.lo_rounds = 10;
    if (pe.lo_trials <= 0) pe.lo_trials = 20;
    if (pe.lo_max_calls <= 0) pe.lo_max_calls = pe.use_elc ? 20 : 50;
    if (pe.min_iters <= 0) pe.min_iters = pe.use_elc ? 20 : 50;
    const lr_ransac_params *p = &pe;
    LR_REQUIRE(p->sample_size == 3 || p->sample_size == 4, LR_EINVAL, "lr_ransac: sample_size must be 3 or 4");
    LR_REQUIRE(p->iters >= 0 && p->iters <= ws->max_iters, LR_ESIZE, "lr_ransac: iters exceeds the workspace");
    LR_REQUIRE(m_max >= 0 && m_max <= ws->max_n0, LR_ESIZE, "lr_ransac: m exceeds the workspace");
    LR_REQUIRE(p->thr2 > 0.0f && p->thr2 < 2048.0f, LR_EINVAL, "lr_ransac: thr2 must be in (0, 2048)");
    int sub = (int)(4095.0 / ((double)p->thr2 * 1.0000001 + 1e-6));     // sub * thr2 * 2^20 < 2^32
    if (sub > 4096) sub = 4096;
    sub &= ~1;                  // the scoring loop takes correspondences two at a time
    if (sub < 2) sub = 2;       // 2 * thr2 * 2^20 < 2^32 for every admissible thr2 (< 2048)
    const bool use_exit = p->confidence > 0.0f && p->confidence < 1.0f;
    // batch lengths: the given one, constant; by default 1024, 8192, 65536, ... -- growing eightfold, so that the exit test is fine-grained
    // where an easy pair stops (the reference tests after every iteration) while a long run still takes few batches (the launches
    // of the batches after the exit cost ~5 us per kernel, ~40 us per batch: measured on the full-list run): 3 batches for 50k ids, 4 for the CLI's default 500k, 5 for 1M; an easy pair scores 1024 ids instead of the 8192 of round 2
    const bool geometric = use_exit && p->batch <= 0;
    long long B = use_exit ? (p->batch > 0 ? p->batch : 1024) : (p->iters > 0 ? p->iters : 1);
    LR_REQUIRE(p->use_elc >= 0 && p->use_elc <= 2, LR_EINVAL, "lr_ransac: use_elc must be 0 (no pre-verification), 1 (edge-length check) or 2 (SPRT)");
    LR_REQUIRE(p->sampler >= 0 && p->sampler <= 2, LR_EINVAL, "lr_ransac: sampler must be 0 (uniform), 1 (PROSAC) or 2 (uniform, unique indices)");
    LR_REQUIRE(p->local_opt >= 0 && p->local_opt <= 2, LR_EINVAL, "lr_ransac: local_opt must be 0, 1 or 2");
    LR_REQUIRE(p->prosac_growth >= 0, LR_EINVAL, "lr_ransac: prosac_growth must be >= 0");
    const int TN = p->prosac_growth > 0 ? p->prosac_growth : 100000;
    const int32_t *G = nullptr;
    // blocks per pair of the local-optimisation launches: with one or a few pairs in the call the GPU is idle next to the one block
    // that optimises a pair's model, so helper blocks take shares of its scoring jobs (lo_score_shared); with a full batch the calls of the
    // other streams fill the GPU (2 / 4 / 8 helper groups per pair of a 32-pair call: +1 ... -3 % over the list runs, round 4)
    const int lo_groups = (p->local_opt == 1 && m_max >= 4 * LO_CHUNK_REC) ? (ws->zP <= 2 ? 16 : ws->zP <= 4 ? 8 : 1) : 1;
    if (ws->timing && ws->ev_pending == 1) { LR_HIP(hipEventRecord(ws->ev[2], st)); }
    if (p->sampler == 1) {
        hipLaunchKernelGGL(prosac_growth_kernel, dim3(1, 1, ws->zP), dim3(1024), 0, st, m_max, m_dev, p->sample_size, TN, ws->prosac_G, ws->z);
        G = ws->prosac_G;
    }
    for (long long h0l = 0; h0l < (p->iters > 0 ? p->iters : 1); h0l += B, B = geometric ? 8 * B : B) {
        const int h0 = (int)h0l;
        const int h1 = h0l + B < p->iters ? (int)(h0l + B) : p->iters;
        const int gb_all = lr_cdiv(h1 - h0 > 0 ? h1 - h0 : 1, 256);      // 256-id groups of the batch
        const int gb = gb_all < 1024 ? gb_all : 1024;                       // blocks per pair (they stride over the groups)
        lr_score_info *info = reinterpret_cast<lr_score_info *>(ws->sc_info);
        // fit: a wave per 64 list slots, at most 4096 waves in the launch (four per SIMD; they stride over longer lists)
        const int fb_all = lr_cdiv(h1 - h0 > 0 ? h1 - h0 : 1, 64), fb_cap = 4096 / ws->zP > 16 ? 4096 / ws->zP : 16;
        const int fb = fb_all < fb_cap ? fb_all : fb_cap;
        if (p->sample_size == 3) {
            hipLaunchKernelGGL(ransac_gen_kernel<3>, dim3(gb, 1, ws->zP), dim3(256), 0, st, corr8, m_max, m_dev, *p, h0, h1, ws->model_h, ws->score_cnt, ws->score_ssq,
                               ws->counters, G, TN, info, ws->z);
            hipLaunchKernelGGL(ransac_fit_kernel<3>, dim3(fb, 1, ws->zP), dim3(64), 0, st, corr8, m_max, m_dev, *p, ws->models, ws->models64, (const int32_t *)ws->model_h,
                               (const int32_t *)ws->counters, G, TN, ws->max_iters, ws->z);
        } else {
            hipLaunchKernelGGL(ransac_gen_kernel<4>, dim3(gb, 1, ws->zP), dim3(256), 0, st, corr8, m_max, m_dev, *p, h0, h1, ws->model_h, ws->score_cnt, ws->score_ssq,
                               ws->counters, G, TN, info, ws->z);
            hipLaunchKernelGGL(ransac_fit_kernel<4>, dim3(fb, 1, ws->zP), dim3(64), 0, st, corr8, m_max, m_dev, *p, ws->models, ws->models64, (const int32_t *)ws->model_h,
                               (const int32_t *)ws->counters, G, TN, ws->max_iters, ws->z);
        }
        const bool sprt = p->use_elc == 2;
        if (sprt)       // every estimated model is pre-verified; the survivors form a second dense list that is scored in full
            hipLaunchKernelGGL(ransac_sprt_kernel, dim3(gb_all, 1, ws->zP), dim3(256), 0, st, corr8, m_max, m_dev, p->thr2,
                               (const float *)ws->models, (const double *)ws->models64, (const int32_t *)ws->model_h, ws->models2, ws->models64_2, ws->model_h2,
                               ws->score_cnt, ws->score_ssq, ws->counters, ws->max_iters, ws->z);
        const int vslot = sprt ? LR_CNT_NVALID2 : LR_CNT_NVALID;
        {
            // pilot-ordered scoring: head (every model over the first records) -> order (one block per pair) -> main.
            // (main: LR_SCORE_BLOCKS / 4 blocks for EVERY pair of a batched call: 64 per pair -- the GPU filled exactly once -- measured
            // 32 % slower, the few large work items of a pair do not balance)
            const float *mdl = sprt ? (const float *)ws->models2 : (const float *)ws->models;
            // (the device decides from the live counts; a batch of up to 1 024 ids is scored in full -- its four ordering launches cost
            // more than they save, and with the pre-check fewer than LR_SC_MIN_V of its ids survive anyway)
            // Nor with a few pairs in the call: the GPU is not full, the main pass over everything costs a lone 30k pair 37 us where the
            // pruned pass costs 39 us AFTER 39 us of ordering launches (FR() 296 -> 254 us; one call of 2 / 4 / 8 / 16 pairs:
            // 0.50 -> 0.46, 0.69 -> 0.65, 1.00 -> 1.03, 1.67 -> 1.76 ms).
            const bool may_prune = m_max >= LR_SC_MIN_M && h1 - h0 >= 2048 && ws->zP > LR_SC_MIN_PAIRS;
            if (may_prune) {
                const int hgx = 64, htotal = hgx * ws->zP;
                hipLaunchKernelGGL(ransac_score_kernel<1>, dim3(8 * lr_cdiv(htotal, 8)), dim3(256), 0, st, corr8, (const float *)ws->corr8s, m_max, m_dev, p->thr2,
                                   mdl, (const float *)ws->models_s, ws->score_cnt, ws->score_ssq, ws->counters, info, (const int32_t *)ws->sc_perm, (const int32_t *)ws->sc_glen, sub,
                                   ws->max_iters, vslot, hgx, htotal, 1, ws->z);
                const dim3 cgrid(lr_cdiv(m_max, 256), 1, ws->zP);
                hipLaunchKernelGGL(ransac_resid_kernel, cgrid, dim3(256), 0, st, corr8, m_max, m_dev, mdl, (const int32_t *)ws->counters, info, ws->sc_cb,
                                   ws->max_iters, vslot, ws->z);
                hipLaunchKernelGGL(ransac_order_kernel, dim3(1, 1, ws->zP), dim3(1024), 0, st, m_max, m_dev, p->thr2, mdl, ws->counters, info,
                                   ws->sc_perm, ws->sc_glen, ws->sc_mb, ws->models_s, ws->max_iters, vslot, ws->z);
                hipLaunchKernelGGL(ransac_scatter_kernel, cgrid, dim3(256), 0, st, corr8, ws->corr8s, m_max, m_dev, (const int32_t *)ws->counters, info,
                                   (const uint8_t *)ws->sc_cb, vslot, ws->z);
            }
            const int sgx = LR_SCORE_BLOCKS / 4, stotal = sgx * ws->zP;
            hipLaunchKernelGGL(ransac_score_kernel<0>, dim3(8 * lr_cdiv(stotal, 8)), dim3(256), 0, st, corr8, (const float *)ws->corr8s, m_max, m_dev, p->thr2,
                               mdl, (const float *)ws->models_s, ws->score_cnt, ws->score_ssq, ws->counters, info, (const int32_t *)ws->sc_perm, (const int32_t *)ws->sc_glen, sub,
                               ws->max_iters, vslot, sgx, stotal, may_prune ? 1 : 0, ws->z);
        }
        if (h0 == 0 && ws->timing && ws->ev_pending == 1) { LR_HIP(hipEventRecord(ws->ev[3], st)); ws->ev_pending = 2; }
        hipLaunchKernelGGL(ransac_final_kernel, dim3(1, 1, ws->zP), dim3(1024), 0, st, ws->score_cnt, ws->score_ssq,
                           sprt ? (const int32_t *)ws->model_h2 : (const int32_t *)ws->model_h, sprt ? (const double *)ws->models64_2 : (const double *)ws->models64,
                           ws->counters, m_max, m_dev, *p, h1, T_out, res, vslot, lo_groups > 1 ? reinterpret_cast<int32_t *>(ws->lo_ctl) : (int32_t *)nullptr, ws->z);
        if (p->local_opt == 1)
            // (near8 = the scratch of the pilot-ordered scoring: this batch's scoring is over, the next batch's scatter pass rewrites it before it is read)
            hipLaunchKernelGGL(ransac_lo_kernel, dim3(lo_groups, 1, ws->zP), dim3(LO_THREADS), 0, st, corr8, m_max, m_dev, *p, h1, 0, ws->counters, ws->lo_list,
                               T_out, res, reinterpret_cast<lr_lo_ctl *>(ws->lo_ctl), ws->corr8s, ws->z);
    }
    if (p->local_opt)          // final iterated least squares over the inliers
        hipLaunchKernelGGL(ransac_lo_kernel, dim3(1, 1, ws->zP), dim3(LO_THREADS), 0, st, corr8, m_max, m_dev, *p, p->iters, 1, ws->counters, ws->lo_list,
                           T_out, res, reinterpret_cast<lr_lo_ctl *>(ws->lo_ctl), (float *)nullptr, ws->z);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// sum of the block partials (fixed order) + Kabsch + result block; run by the block that finishes last
__device__ void refit_solve_body(const double *__restrict__ partial, int nblocks, const double *__restrict__ T_in,
                   const lr_ransac_result *__restrict__ gate, double *__restrict__ T_out, int32_t *__restrict__ n_inl,
                   lr_pair_result *__restrict__ pair_out, const int32_t *__restrict__ counters, int weighted)
{
    __shared__ double mom[16];
    __shared__ double stage[64 * 16];
    // partials are summed in block order (reproducible); they are staged through LDS 64 blocks at a time so the
    // 16 summing lanes do not walk a chain of dependent global loads
    double acc = 0.0;
    for (int b0 = 0; b0 < nblocks; b0 += 64) {
        const int nb = min(64, nblocks - b0);
        for (int t = threadIdx.x; t < nb * 16; t += blockDim.x) stage[t] = partial[(size_t)b0 * 16 + t];
        __syncthreads();
        if (threadIdx.x < 16)
            for (int b = 0; b < nb; ++b) acc += stage[b * 16 + threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x < 16) mom[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x != 0) return;
    const bool have_model = gate ? gate->best_h >= 0 : true;
    const double n = mom[0];
    double T[16];
    int n_used;
    if (!have_model || !(n > 0.0) || (!weighted && n < 3.0)) {
        for (int k = 0; k < 16; ++k) T[k] = T_in[k];
        n_used = have_model ? (int)n : 0;
    } else {
        double cp[3], cq[3], H[3][3];
        for (int a = 0; a < 3; ++a) { cp[a] = mom[1 + a] / n; cq[a] = mom[4 + a] / n; }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) H[a][b] = mom[7 + 3 * a + b] - (n * cp[a]) * cq[b];
        lr_rt_from_cov(H, cp, cq, T);
        n_used = (int)n;
    }
    for (int k = 0; k < 16; ++k) T_out[k] = T[k];
    if (n_inl) *n_inl = n_used;
    if (pair_out) {          // result block of lr_register_pair (fused: saves a launch)
        for (int k = 0; k < 16; ++k) { pair_out->T[k] = T[k]; pair_out->T_ransac[k] = T_in[k]; }
        pair_out->ransac = *gate;
        pair_out->n_corr = counters[LR_CNT_NCORR];
        pair_out->n_refit = n_used;
        pair_out->n_nn_fixed = counters[LR_CNT_FIX_TOTAL];
        pair_out->status = gate->best_h < 0 ? 1 : 0;
        for (int q = 0; q < 8; ++q) pair_out->reserved[q] = 0;
        {   // reserved[0]: (model, correspondence) evaluations of the scoring passes, in ppm of scanning every list in full (0: not recorded)
            const lr_ransac_state *state = reinterpret_cast<const lr_ransac_state *>(counters + LR_CNT_COUNT);
            if (state->evals_full > 0) pair_out->reserved[0] = (int32_t)((double)state->evals / (double)state->evals_full * 1e6);
            pair_out->reserved[1] = state->lo_timeouts;       // hand-off waits of the local optimisation that hit their bound (lr_ransac_state)
            // single-pair calls: the one form of the filter pass that was launched was the wrong one (forward: 1, reverse: 2) -- every row went through the exact scan
            pair_out->reserved[2] = (counters[LR_CNT_FORM_MISS_F] ? 1 : 0) | (counters[LR_CNT_FORM_MISS_R] ? 2 : 0);
        }
        for (int k = 0; k < 16; ++k) pair_out->T_icp[k] = T[k];      // overwritten by pair_icp_kernel when the ICP stage runs
        pair_out->icp.fitness = 0.0; pair_out->icp.inlier_rmse = 0.0; pair_out->icp.n_corr = 0; pair_out->icp.iterations = 0;
    }
}

// ------------------------------------------------------------------ refit (FR.py:99-111)
#define LR_REFIT_PER 4
// partial[b][0] = n, [1..3] = sum p, [4..6] = sum q, [7..15] = sum p q^T over the inliers of block b
__global__ void __launch_bounds__(256)
refit_moments_kernel(const float *__restrict__ xyz0, int n0, const float *__restrict__ xyz1, const int32_t *__restrict__ idx1,
                     const double *__restrict__ T_in, double thr2, double *__restrict__ partial,
                     const int32_t *__restrict__ idx0, const int32_t *__restrict__ m_dev,
                     const float *__restrict__ F0, const float *__restrict__ F1,
                     int32_t *__restrict__ ticket, const lr_ransac_result *__restrict__ gate, double *__restrict__ T_out,
                     int32_t *__restrict__ n_inl, lr_pair_result *__restrict__ pair_out, const int32_t *__restrict__ counters, lr_zargs z)
{
    __shared__ double sm[4][16];
    __shared__ int s_last;
    if (z.descs) {
        const lr_pair_desc d = z.descs[blockIdx.z];
        xyz0 = d.xyz0; xyz1 = d.xyz1; n0 = d.n0;
        if (F0) { F0 = d.F0; F1 = d.F1; }
    }
    lr_z(idx1, z, blockIdx.z); lr_z(T_in, z, blockIdx.z); lr_z(partial, z, blockIdx.z); lr_z(idx0, z, blockIdx.z); lr_z(m_dev, z, blockIdx.z); lr_z(ticket, z, blockIdx.z); lr_z(gate, z, blockIdx.z); lr_z(T_out, z, blockIdx.z); lr_z(n_inl, z, blockIdx.z); lr_z(counters, z, blockIdx.z);
    if (pair_out) pair_out += blockIdx.z;                  // the caller's result array, one block per pair
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = T_in[k];
    double v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = 0.0;
    // pairs (i, idx1[i]) over all n0 rows, or -- with idx0 -- the listed pairs (idx0[c], idx1[c]), c < *m_dev
    if (m_dev) n0 = min(n0, *m_dev);
    // LR_REFIT_PER pairs per thread (the 16 fp64 wave reductions below cost far more than a pair's moments: one pair per thread spent
    // most of the kernel in shuffles); the loads of all of them are issued before the first use
    int jj[LR_REFIT_PER], pp[LR_REFIT_PER];
#pragma unroll
    for (int u = 0; u < LR_REFIT_PER; ++u) {
        const int i = (blockIdx.x * LR_REFIT_PER + u) * 256 + threadIdx.x;
        jj[u] = i < n0 ? idx1[i] : -1;
        pp[u] = i < n0 ? (idx0 ? idx0[i] : i) : 0;
    }
#pragma unroll
    for (int u = 0; u < LR_REFIT_PER; ++u) {
        if (jj[u] < 0) continue;
        const int j = jj[u], pi = pp[u];
        double p[3] = { (double)xyz0[3 * pi], (double)xyz0[3 * pi + 1], (double)xyz0[3 * pi + 2] };
        double q[3] = { (double)xyz1[3 * j], (double)xyz1[3 * j + 1], (double)xyz1[3 * j + 2] };
        double r[3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
            r[a] = (((T[4 * a] * p[0] + T[4 * a + 1] * p[1]) + T[4 * a + 2] * p[2]) + T[4 * a + 3]) - q[a];
        double d2 = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
        if (d2 < thr2) {
            // weight 1 (Umeyama, FR.py:110-111) or the inverse feature distance of the pair (DGR's weighted
            // Procrustes refit, DGR/core/deep_global_registration.py:531-537)
            double w = 1.0;
            if (F0) {
                const float *fa = F0 + (size_t)pi * 32, *fb = F1 + (size_t)j * 32;
                float acc = 0.0f;
                for (int k = 0; k < 32; ++k) { const float e = fa[k] - fb[k]; const float q2 = e * e; acc = acc + q2; }
                w = 1.0 / (double)fmaxf(__builtin_sqrtf(acc), 1e-12f);
            }
            v[0] += w;
#pragma unroll
            for (int a = 0; a < 3; ++a) { v[1 + a] += w * p[a]; v[4 + a] += w * q[a]; }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) v[7 + 3 * a + b] += (w * p[a]) * q[b];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        double s = v[k];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        if (lane == 0) sm[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 16)
        partial[(size_t)blockIdx.x * 16 + threadIdx.x] = ((sm[0][threadIdx.x] + sm[1][threadIdx.x]) + sm[2][threadIdx.x]) + sm[3][threadIdx.x];
    // ---- last block done: agent-scope release of the partials, ticket, acquire, then the solve (saves a launch)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == (int)gridDim.x - 1);
        if (s_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next call
        }
    }
    __syncthreads();
    if (!s_last) return;
    refit_solve_body(partial, (int)gridDim.x, T_in, gate, T_out, n_inl, pair_out, counters, F0 ? 1 : 0);
}

int lr_refit_run(lr_workspace *ws, const float *xyz0, int n0, const float *xyz1, const int32_t *idx1,
                 const double *T_in, double thr2, double *T_out, int32_t *n_inl, const lr_ransac_result *gate,
                 hipStream_t st, lr_pair_result *pair_out, const int32_t *idx0, const int32_t *m_dev,
                 const float *F0, const float *F1)
{
    const int nb = lr_cdiv(n0, 256 * LR_REFIT_PER);
    int32_t *ticket = ws->counters + LR_CNT_REFIT_TICKET;
    if (!pair_out) LR_HIP(hipMemsetAsync(ticket, 0, sizeof(int32_t), st));      // lr_register_pair starts from cleared counters
    hipLaunchKernelGGL(refit_moments_kernel, dim3(nb, 1, ws->zP), dim3(256), 0, st, xyz0, n0, xyz1, idx1, T_in, thr2, ws->refit_part, idx0, m_dev,
                       F0, F1, ticket, gate, T_out, n_inl, pair_out, (const int32_t *)ws->counters, ws->z);
    LR_LAUNCH_CHECK();
    return LR_OK;
}

// ------------------------------------------------------------------ a13: explicit point pairs
__global__ void kabsch_points_kernel(const double *__restrict__ P, const double *__restrict__ Q, const double *__restrict__ w,
                                     int n, double *__restrict__ T_out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double cp[3] = { 0, 0, 0 }, cq[3] = { 0, 0, 0 }, W = 0.0;
    for (int i = 0; i < n; ++i) {
        double wi = w ? w[i] : 1.0;
        W = W + wi;
        for (int a = 0; a < 3; ++a) { cp[a] = cp[a] + wi * P[3 * i + a]; cq[a] = cq[a] + wi * Q[3 * i + a]; }
    }
    for (int a = 0; a < 3; ++a) { cp[a] = cp[a] / W; cq[a] = cq[a] / W; }
    double H[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } };
    for (int i = 0; i < n; ++i) {
        double wi = w ? w[i] : 1.0;
        double pc[3], qc[3];
        for (int a = 0; a < 3; ++a) { pc[a] = P[3 * i + a] - cp[a]; qc[a] = Q[3 * i + a] - cq[a]; }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) H[a][b] = H[a][b] + (wi * pc[a]) * qc[b];
    }
    double T[16];
    lr_rt_from_cov(H, cp, cq, T);
    for (int k = 0; k < 16; ++k) T_out[k] = T[k];
}

extern "C" int lr_kabsch(const double *P, const double *Q, const double *w, int n, double *T_out, void *stream)
{
    LR_REQUIRE(P && Q && T_out && n >= 1, LR_EINVAL, "lr_kabsch: bad argument");
    hipLaunchKernelGGL(kabsch_points_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, P, Q, w, n, T_out);
    LR_LAUNCH_CHECK();
    return LR_OK;
}
