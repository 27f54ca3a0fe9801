/*
 * oracle.c -- CPU restatement of the registration hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (lidarregistration_amd/) never does.
 *
 * What it restates (file:line relative to the reference tree):
 *   orc_row_norms, orc_nn_top2     Experiments/algorithms/matching.py:22-65 (find_nn / knn_dist)
 *   orc_feat_ratio                 Experiments/algorithms/matching.py:89-98
 *   orc_elc                        GC-RANSAC/src/pygcransac/include/preemption/preemption_edge_length.h:71-128
 *   orc_kabsch_points/_moments     Experiments/models/common.py:7-45, DGR/util/procrustes.py:34-56
 *                                  (R = V diag(1,1,det) U^T, t = mu_B - R mu_A), solved here through
 *                                  Horn's quaternion form; largest eigenvector in closed form (Newton on the characteristic polynomial +
 *                                  adjugate column; cyclic Jacobi for degenerate input) so that it needs only + - * / sqrt
 *   orc_ransac                     Experiments/algorithms/FR.py:122-139 -> Open3D 0.13.0
 *                                  registration_ransac_based_on_correspondence (not vendored, pinned in
 *                                  Requirements/conda_GC_full.yml:115): uniform sampling with replacement,
 *                                  edge-length checker 0.9, inlier count then RMSE, threshold 0.6 m
 *   orc_refit                      Experiments/algorithms/FR.py:99-111
 *   orc_ransac options             the GC codebase (Experiments/algorithms/GC_RANSAC.py:8-55 -> pygcransac 0.1, driver
 *                                  GC-RANSAC/src/pygcransac/src/gcransac_python.cpp:404-624; library not vendored):
 *                                  unique-index samplers, PROSAC, MSAC, SPRT pre-verification (sprt_test), GC-RANSAC's
 *                                  local optimisation at spatial-coherence weight 0 and the final iterated least squares
 *                                  (lo_optimise / lo_polish) -- restated from the published algorithms
 *   orc_icp                        Experiments/test.py:183-189 (Open3D registration_icp)
 *
 * Parity status: NN / ratio / Kabsch are pinned by golden vectors generated from the importable
 * reference (tests/golden/make_golden.py), the composed pipeline by fixture G11.  The RANSAC loop itself, its
 * options (PROSAC, SPRT, local optimisation) and ICP live in un-vendored third-party code (Open3D / pygcransac):
 * PARITY UNPINNED for those -- they are anchored on the reference's call-site parameters, the in-tree ELC source,
 * the published algorithms and on planted-model recovery properties only.
 *
 * Arithmetic contract (the HIP kernels implement exactly this; build with -ffp-contract=off):
 *   norm(x)   = chain n = fmaf(x[k], x[k], n), k = 0..D-1, n0 = +0
 *   dot(a,b)  = chain c = fmaf(a[k], b[k], c), k = 0..D-1, c0 = +0     (== v_mfma_f32_32x32x2_f32 order)
 *   d2(i,j)   = fmaf(-2, dot, n0[i] + n1[j]);  s = sqrtf(fmaxf(d2, 1e-30f))
 *   NN order  = ascending (s, j): first minimal value wins, as torch.min(dim=1) does.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ NN ---- */

ORC_API void orc_row_norms(const float *F, int n, int d, float *out)
{
    for (int i = 0; i < n; ++i) {
        float acc = 0.0f;
        for (int k = 0; k < d; ++k) acc = fmaf(F[(size_t)i * d + k], F[(size_t)i * d + k], acc);
        out[i] = acc;
    }
}

/* matching.py:25-41 + :43-65.  idx2/s2 may be NULL (return_2nd=False).
 * Returns, per query row of F0, the first and second nearest rows of F1.          */
ORC_API void orc_nn_top2(const float *F0, int n0, const float *F1, int n1, int d,
                         int32_t *idx1, int32_t *idx2, float *s1, float *s2)
{
    float *nrm0 = (float *)malloc(sizeof(float) * (size_t)n0);
    float *nrm1 = (float *)malloc(sizeof(float) * (size_t)n1);
    /* F1 transposed [k][j] so the inner loop vectorises across j; each j keeps its own chain. */
    float *F1T = (float *)malloc(sizeof(float) * (size_t)n1 * d);
    orc_row_norms(F0, n0, d, nrm0);
    orc_row_norms(F1, n1, d, nrm1);
    for (int j = 0; j < n1; ++j)
        for (int k = 0; k < d; ++k) F1T[(size_t)k * n1 + j] = F1[(size_t)j * d + k];

#pragma omp parallel
    {
        enum { JB = 64 };
        float acc[JB];
#pragma omp for schedule(static)
        for (int i = 0; i < n0; ++i) {
            const float *a = F0 + (size_t)i * d;
            float b1 = INFINITY, b2 = INFINITY;
            int32_t i1 = -1, i2 = -1;
            for (int j0 = 0; j0 < n1; j0 += JB) {
                int jn = n1 - j0 < JB ? n1 - j0 : JB;
                for (int jj = 0; jj < jn; ++jj) acc[jj] = 0.0f;
                for (int k = 0; k < d; ++k) {
                    const float ak = a[k];
                    const float *brow = F1T + (size_t)k * n1 + j0;
                    for (int jj = 0; jj < jn; ++jj) acc[jj] = fmaf(ak, brow[jj], acc[jj]);
                }
                for (int jj = 0; jj < jn; ++jj) {
                    float t = nrm0[i] + nrm1[j0 + jj];
                    float d2 = fmaf(-2.0f, acc[jj], t);
                    float s = sqrtf(fmaxf(d2, 1e-30f));
                    /* j ascends, so strict '<' keeps the first minimal value */
                    if (s < b1) { b2 = b1; i2 = i1; b1 = s; i1 = j0 + jj; }
                    else if (s < b2) { b2 = s; i2 = j0 + jj; }
                }
            }
            idx1[i] = i1;
            if (s1) s1[i] = b1;
            if (idx2) idx2[i] = i2;
            if (s2) s2[i] = b2;
        }
    }
    free(nrm0); free(nrm1); free(F1T);
}

/* matching.py:89-98: ||A-B1|| / (||A-B2|| + 1e-6), direct differences, sequential k. */
ORC_API void orc_feat_ratio(const float *F0, const float *F1, int d, int m,
                            const int32_t *i0, const int32_t *i1, const int32_t *i2, float *out)
{
    for (int c = 0; c < m; ++c) {
        const float *a = F0 + (size_t)i0[c] * d;
        const float *b1 = F1 + (size_t)i1[c] * d;
        const float *b2 = F1 + (size_t)i2[c] * d;
        float s1 = 0.0f, s2 = 0.0f;
        for (int k = 0; k < d; ++k) {
            float e1 = a[k] - b1[k];
            float e2 = a[k] - b2[k];
            float q1 = e1 * e1, q2 = e2 * e2;
            s1 = s1 + q1;
            s2 = s2 + q2;
        }
        float d1 = sqrtf(s1), d2v = sqrtf(s2);
        out[c] = d1 / (d2v + 1e-6f);
    }
}

/* -------------------------------------------------------------- Philox ---- */

static inline void philox4x32_10(uint32_t ctr[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * ctr[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr[2];
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ ctr[1] ^ k0;
        uint32_t n2 = hi0 ^ ctr[3] ^ k1;
        ctr[0] = n0; ctr[1] = lo1; ctr[2] = n2; ctr[3] = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

ORC_API void orc_philox(uint64_t seed, uint64_t h, uint32_t out[4])
{
    uint32_t c[4] = { (uint32_t)h, (uint32_t)(h >> 32), 0u, 0u };
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    memcpy(out, c, sizeof(c));
}

/* sample_size (<=4) correspondence indices in [0,m), uniform with replacement */
static inline void draw_sample(uint64_t seed, uint64_t h, int m, int ns, int32_t *s)
{
    uint32_t w[4];
    orc_philox(seed, h, w);
    for (int k = 0; k < ns; ++k) s[k] = (int32_t)(((uint64_t)w[k] * (uint64_t)(uint32_t)m) >> 32);
}

/* PROSAC (Chum & Matas, CVPR 2005; tabulated as in USAC's / GC-RANSAC's prosac_sampler.h, a third-party dependency that
 * is not vendored -- GC_RANSAC.py:19,24,39-43 only select it and sort the pairs by quality).  Growth function
 *   T_n = T_N prod_{i<ns} (n-i)/(M-i),  G[ns] = 1,  G[n+1] = G[n] + ceil(T_{n+1} - T_n)
 * and draw k (= hypothesis id + 1, k <= T_N) takes ns-1 indices uniformly (with replacement, like the uniform sampler
 * above) from the first n_k - 1 correspondences plus the n_k-th, n_k = min(M, ns + #{n in [ns, M) : G[n] <= k}).
 * T_n is evaluated in closed form, in this operation order (the HIP kernel does the same).                         */
static double prosac_Tn(int n, int M, int ns, double TN)
{
    double t = TN;
    for (int i = 0; i < ns; ++i) t = t * (double)(n - i) / (double)(M - i);
    return t;
}

/* G[ns..M], entries below ns unused; caller frees */
static int32_t *prosac_table(int M, int ns, int TN)
{
    int32_t *G = (int32_t *)malloc(sizeof(int32_t) * (size_t)(M + 2));
    long long g = 1;
    for (int n = ns; n <= M; ++n) {
        G[n] = (int32_t)(g < 0x3fffffffLL ? g : 0x3fffffffLL);
        if (n < M) {
            long long d = (long long)ceil(prosac_Tn(n + 1, M, ns, (double)TN) - prosac_Tn(n, M, ns, (double)TN));
            if (d < 1) d = 1;
            g += d;
        }
    }
    return G;
}

static inline void draw_sample_prosac(uint64_t seed, uint64_t h, int m, int ns, const int32_t *G, int TN, int32_t *s)
{
    if (!G || h >= (uint64_t)TN || m <= ns) { draw_sample(seed, h, m, ns, s); return; }
    const int k = (int)h + 1;
    int lo = ns, hi = m;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (G[mid] <= k) lo = mid + 1; else hi = mid; }
    const int n = lo < m ? lo : m;
    uint32_t w[4];
    orc_philox(seed, h, w);
    for (int j = 0; j < ns - 1; ++j) s[j] = (int32_t)(((uint64_t)w[j] * (uint64_t)(uint32_t)(n - 1)) >> 32);
    s[ns - 1] = n - 1;
}

/* ----------------------------------------------------------------- ELC ---- */

/* preemption_edge_length.h:71-128: reject when any pair of sample edges differs by more than 0.9 */
static int elc_ok(const float *src, const float *tgt, const int32_t *s, int ns)
{
    const double SIM = 0.9;
    for (int i = 0; i < ns; ++i)
        for (int j = i + 1; j < ns; ++j) {
            double sx = (double)src[3 * s[j]] - (double)src[3 * s[i]];
            double sy = (double)src[3 * s[j] + 1] - (double)src[3 * s[i] + 1];
            double sz = (double)src[3 * s[j] + 2] - (double)src[3 * s[i] + 2];
            double tx = (double)tgt[3 * s[j]] - (double)tgt[3 * s[i]];
            double ty = (double)tgt[3 * s[j] + 1] - (double)tgt[3 * s[i] + 1];
            double tz = (double)tgt[3 * s[j] + 2] - (double)tgt[3 * s[i] + 2];
            double ds = sqrt((sx * sx + sy * sy) + sz * sz);
            double dt = sqrt((tx * tx + ty * ty) + tz * tz);
            if (ds < dt * SIM || dt < ds * SIM) return 0;
        }
    return 1;
}

ORC_API int orc_elc(const float *src, const float *tgt, const int32_t *sample, int ns)
{
    return elc_ok(src, tgt, sample, ns);
}

/* -------------------------------------------------------------- Kabsch ---- */

#define JACOBI_SWEEPS 10          /* upper bound; the sweep loop stops once the off-diagonal mass is below 1e-15 of the diagonal */

/* Largest-eigenvalue eigenvector of a symmetric 4x4 by cyclic Jacobi (fixed sweep count). */
static void jacobi4_maxvec(double A[4][4], double q[4])
{
    double V[4][4] = { {1,0,0,0}, {0,1,0,0}, {0,0,1,0}, {0,0,0,1} };
    for (int sweep = 0; sweep < JACOBI_SWEEPS; ++sweep) {
        double off2 = ((((A[0][1] * A[0][1] + A[0][2] * A[0][2]) + A[0][3] * A[0][3]) + A[1][2] * A[1][2]) + A[1][3] * A[1][3]) + A[2][3] * A[2][3];
        double dia2 = ((A[0][0] * A[0][0] + A[1][1] * A[1][1]) + A[2][2] * A[2][2]) + A[3][3] * A[3][3];
        if (off2 <= 1e-30 * dia2) break;
        for (int p = 0; p < 3; ++p)
            for (int r = p + 1; r < 4; ++r) {
                double apq = A[p][r];
                if (apq == 0.0) continue;
                double theta = (A[r][r] - A[p][p]) / (2.0 * apq);
                double at = fabs(theta);
                double t = 1.0 / (at + sqrt(theta * theta + 1.0));
                if (theta < 0.0) t = -t;
                double c = 1.0 / sqrt(t * t + 1.0);
                double s = t * c;
                double tau = s / (1.0 + c);
                double h = t * apq;
                A[p][p] = A[p][p] - h;
                A[r][r] = A[r][r] + h;
                A[p][r] = 0.0; A[r][p] = 0.0;
                for (int k = 0; k < 4; ++k) {
                    if (k == p || k == r) continue;
                    double g = A[k][p], f = A[k][r];
                    double gn = g - s * (f + g * tau);
                    double fn = f + s * (g - f * tau);
                    A[k][p] = gn; A[p][k] = gn;
                    A[k][r] = fn; A[r][k] = fn;
                }
                for (int k = 0; k < 4; ++k) {
                    double g = V[k][p], f = V[k][r];
                    V[k][p] = g - s * (f + g * tau);
                    V[k][r] = f + s * (g - f * tau);
                }
            }
    }
    int im = 0;
    for (int k = 1; k < 4; ++k) if (A[k][k] > A[im][im]) im = k;
    double w = V[0][im], x = V[1][im], y = V[2][im], z = V[3][im];
    double nn = sqrt(((w * w + x * x) + y * y) + z * z);
    q[0] = w / nn; q[1] = x / nn; q[2] = y / nn; q[3] = z / nn;
}

/* Largest-eigenvalue eigenvector of a symmetric 4x4 in closed form (round 6; VERDICT r5 #1c: the cyclic Jacobi below is a chain of 36-48
 * dependent rotations of 4 divisions and 2 square roots -- 60 us of one lane for the hypothesis fits, 50 for the refit).
 *   1. the characteristic polynomial from the trace, the principal 2x2 / 3x3 minors and the determinant;
 *   2. its largest root by Newton's iteration from Gershgorin's upper bound: beyond the largest root the polynomial is positive, increasing
 *      and convex, so the iterates decrease monotonically and the loop ends when one no longer does (6-8 steps of one division; a close
 *      second eigenvalue -- near-collinear points -- takes 20-30);
 *   3. the eigenvector as the column of adj(N - lambda I) = prod(lambda_k - lambda) v v^T with the largest diagonal cofactor.
 * Only + - * / sqrt in a fixed order: bit-identical here and in lidarregistration_amd/csrc/lr_kabsch.h (same text).  Returns 0 -- the caller
 * runs Jacobi -- when the matrix is zero / not finite or the adjugate is at rounding level (a double largest eigenvalue: collinear points).
 * Accuracy: the root carries eps |N| (|N| / gap), the vector eps (|N| / gap)^2, gap = distance to the second eigenvalue: 1e-15 rad for
 * the refit's thousands of points, <= 2e-6 rad over 200 000 random three-point samples, and where it is worse the rotation about the
 * points' common line is not determined by float32 coordinates either (tests/test_oracle_ransac.py). */
static inline double det3_(double a, double b, double c, double d, double e, double f, double g, double h, double i)
{
    return (a * (e * i - f * h) - b * (d * i - f * g)) + c * (d * h - e * g);
}
static int horn4_maxvec_newton(const double N[4][4], double q[4])
{
    const double a = N[0][0], b = N[1][1], c = N[2][2], d = N[3][3];
    const double n01 = N[0][1], n02 = N[0][2], n03 = N[0][3], n12 = N[1][2], n13 = N[1][3], n23 = N[2][3];
    /* elementary symmetric functions of the eigenvalues: trace, principal 2x2 and 3x3 minors, determinant */
    const double e1 = (a + b) + (c + d);
    const double e2 = (((a * b - n01 * n01) + (a * c - n02 * n02)) + ((a * d - n03 * n03) + (b * c - n12 * n12))) + ((b * d - n13 * n13) + (c * d - n23 * n23));
    const double m0 = det3_(b, n12, n13, n12, c, n23, n13, n23, d);      /* without row / column 0 */
    const double m1 = det3_(a, n02, n03, n02, c, n23, n03, n23, d);
    const double m2 = det3_(a, n01, n03, n01, b, n13, n03, n13, d);
    const double m3 = det3_(a, n01, n02, n01, b, n12, n02, n12, c);
    const double e3 = (m0 + m1) + (m2 + m3);
    /* det N by the first row */
    const double k1 = det3_(n01, n12, n13, n02, c, n23, n03, n23, d);
    const double k2 = det3_(n01, b, n13, n02, n12, n23, n03, n13, d);
    const double k3 = det3_(n01, b, n12, n02, n12, c, n03, n13, n23);
    const double e4 = ((a * m0 - n01 * k1) + n02 * k2) - n03 * k3;
    /* Gershgorin: an upper bound of the largest eigenvalue */
    const double r0 = ((a + fabs(n01)) + fabs(n02)) + fabs(n03), r1 = ((b + fabs(n01)) + fabs(n12)) + fabs(n13);
    const double r2 = ((c + fabs(n02)) + fabs(n12)) + fabs(n23), r3 = ((d + fabs(n03)) + fabs(n13)) + fabs(n23);
    double lam = r0 > r1 ? r0 : r1; { const double r = r2 > r3 ? r2 : r3; lam = lam > r ? lam : r; }
    const double bound = lam;
    if (!(bound > 0.0 && bound < 1.0e150)) return 0;
    int it = 0;
    for (; it < 64; ++it) {
        const double p = (((lam - e1) * lam + e2) * lam - e3) * lam + e4;
        const double dp = ((4.0 * lam - 3.0 * e1) * lam + 2.0 * e2) * lam - e3;
        if (!(dp > 0.0)) break;
        const double nl = lam - p / dp;
        if (!(nl < lam)) break;
        lam = nl;
    }
    /* B = N - lam I; its adjugate is (a multiple of) v v^T */
    const double A = a - lam, B = b - lam, C = c - lam, D = d - lam;
    const double c00 = det3_(B, n12, n13, n12, C, n23, n13, n23, D);
    const double c11 = det3_(A, n02, n03, n02, C, n23, n03, n23, D);
    const double c22 = det3_(A, n01, n03, n01, B, n13, n03, n13, D);
    const double c33 = det3_(A, n01, n02, n01, B, n12, n02, n12, C);
    const double c01 = -det3_(n01, n12, n13, n02, C, n23, n03, n23, D);
    const double c02 = det3_(n01, B, n13, n02, n12, n23, n03, n13, D);
    const double c03 = -det3_(n01, B, n12, n02, n12, C, n03, n13, n23);
    const double c12 = -det3_(A, n01, n03, n02, n12, n23, n03, n13, D);
    const double c13 = det3_(A, n01, n02, n02, n12, C, n03, n13, n23);
    const double c23 = -det3_(A, n01, n02, n01, B, n12, n03, n13, n23);
    double w = c00, x = c01, y = c02, z = c03, best = fabs(c00);
    if (fabs(c11) > best) { best = fabs(c11); w = c01; x = c11; y = c12; z = c13; }
    if (fabs(c22) > best) { best = fabs(c22); w = c02; x = c12; y = c22; z = c23; }
    if (fabs(c33) > best) { best = fabs(c33); w = c03; x = c13; y = c23; z = c33; }
    if (!(best > 1.0e-6 * ((bound * bound) * bound))) return 0;
    const double nn = sqrt(((w * w + x * x) + y * y) + z * z);
    if (!(nn > 0.0)) return 0;
    q[0] = w / nn; q[1] = x / nn; q[2] = y / nn; q[3] = z / nn;
    return 1;
}

/* H = sum (p - cp)(q - cq)^T  (3x3, row = source axis, col = target axis) -> R, t with q ~ R p + t */
static void rt_from_cov(const double H[3][3], const double cp[3], const double cq[3], double T[16])
{
    double Sxx = H[0][0], Sxy = H[0][1], Sxz = H[0][2];
    double Syx = H[1][0], Syy = H[1][1], Syz = H[1][2];
    double Szx = H[2][0], Szy = H[2][1], Szz = H[2][2];
    double N[4][4];
    N[0][0] = (Sxx + Syy) + Szz;  N[0][1] = Syz - Szy;           N[0][2] = Szx - Sxz;            N[0][3] = Sxy - Syx;
    N[1][0] = N[0][1];            N[1][1] = (Sxx - Syy) - Szz;   N[1][2] = Sxy + Syx;            N[1][3] = Szx + Sxz;
    N[2][0] = N[0][2];            N[2][1] = N[1][2];             N[2][2] = (Syy - Sxx) - Szz;    N[2][3] = Syz + Szy;
    N[3][0] = N[0][3];            N[3][1] = N[1][3];             N[3][2] = N[2][3];              N[3][3] = (Szz - Sxx) - Syy;
    double q[4];
    if (!horn4_maxvec_newton(N, q)) jacobi4_maxvec(N, q);
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double R[3][3];
    R[0][0] = 1.0 - 2.0 * (y * y + z * z); R[0][1] = 2.0 * (x * y - w * z);       R[0][2] = 2.0 * (x * z + w * y);
    R[1][0] = 2.0 * (x * y + w * z);       R[1][1] = 1.0 - 2.0 * (x * x + z * z); R[1][2] = 2.0 * (y * z - w * x);
    R[2][0] = 2.0 * (x * z - w * y);       R[2][1] = 2.0 * (y * z + w * x);       R[2][2] = 1.0 - 2.0 * (x * x + y * y);
    for (int a = 0; a < 3; ++a) {
        double rc = (R[a][0] * cp[0] + R[a][1] * cp[1]) + R[a][2] * cp[2];
        T[4 * a + 0] = R[a][0]; T[4 * a + 1] = R[a][1]; T[4 * a + 2] = R[a][2];
        T[4 * a + 3] = cq[a] - rc;
    }
    T[12] = 0.0; T[13] = 0.0; T[14] = 0.0; T[15] = 1.0;
}

/* Kabsch on n explicit point pairs (minimal samples, goldens).  T is 4x4 row-major, column-vector
 * convention, maps src -> tgt.  Optional weights w (NULL = 1).                                       */
ORC_API void orc_kabsch_points(const double *P, const double *Q, const double *w, int n, double T[16])
{
    double cp[3] = {0, 0, 0}, cq[3] = {0, 0, 0}, W = 0.0;
    for (int i = 0; i < n; ++i) {
        double wi = w ? w[i] : 1.0;
        W = W + wi;
        for (int a = 0; a < 3; ++a) { cp[a] = cp[a] + wi * P[3 * i + a]; cq[a] = cq[a] + wi * Q[3 * i + a]; }
    }
    for (int a = 0; a < 3; ++a) { cp[a] = cp[a] / W; cq[a] = cq[a] / W; }
    double H[3][3] = { {0,0,0}, {0,0,0}, {0,0,0} };
    for (int i = 0; i < n; ++i) {
        double wi = w ? w[i] : 1.0;
        double pc[3], qc[3];
        for (int a = 0; a < 3; ++a) { pc[a] = P[3 * i + a] - cp[a]; qc[a] = Q[3 * i + a] - cq[a]; }
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) H[a][b] = H[a][b] + (wi * pc[a]) * qc[b];
    }
    rt_from_cov(H, cp, cq, T);
}

/* Kabsch from raw moments: n, sum p, sum q, sum p q^T (what the refit kernel accumulates). */
ORC_API void orc_kabsch_moments(double n, const double sp[3], const double sq[3], const double spq[9], double T[16])
{
    double cp[3], cq[3], H[3][3];
    for (int a = 0; a < 3; ++a) { cp[a] = sp[a] / n; cq[a] = sq[a] / n; }
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) H[a][b] = spq[3 * a + b] - (n * cp[a]) * cq[b];
    rt_from_cov(H, cp, cq, T);
}

/* -------------------------------------------------------------- RANSAC ---- */

typedef struct {
    int32_t  sample_size;     /* 3 (GC minimal solver) or 4 (FR.py:134 ransac_n)      */
    int32_t  use_elc;         /* pre-verification: 0 none, 1 edge-length check on the sample (ELC), 2 SPRT on the model */
    float    thr2;            /* squared inlier threshold (0.6 m)^2                   */
    int32_t  iters;           /* hypotheses h = 0 .. iters-1                          */
    uint64_t seed;
    float    confidence;      /* early exit between batches (FR.py:136, GC_RANSAC.py:26); >= 1 or <= 0: none */
    int32_t  batch;           /* batch length, constant (0 -> 1024, 8192, 65536, ...: eightfold) */
    int32_t  sampler;         /* 0 uniform with replacement (Open3D); 1 PROSAC (GC_RANSAC.py:24,39-43): correspondences best
                                 first; 2 uniform, unique indices (GC-RANSAC's UniformSampler).  1 and 2 reject a draw with a
                                 repeated index: it consumes its id like a failed pre-check                                  */
    int32_t  prosac_growth;   /* T_N of the PROSAC growth function (0 -> 100000)       */
    int32_t  scoring;         /* 0: inlier count, then error sum (Open3D); 1: MSAC at thr2; 2: MSAC at GC-RANSAC's TRUNCATED
                                 threshold (3/2 thr)^2 -- inlier test, cost, exit rule, local optimisation and polish all use
                                 2.25 thr2 (upstream-recalled: GCRANSAC::run sets truncated_threshold = 3/2 threshold and
                                 MSACScoringFunction tests and scores against its square)                  */
    int32_t  local_opt;       /* 0 none; 1 GC-RANSAC's local optimisation of every new best model + final iterated least
                                 squares (GC_RANSAC.py:36-37 --GC_LO True); 2 the final iterated least squares only        */
    /* gcransac_python.cpp:513-517,553-556,579-582 -- three settings whose meaning lives in un-vendored code; 0 = default */
    int32_t  lo_rounds;       /* rounds of one local optimisation (upstream max_graph_cut_number; default 10)           */
    int32_t  lo_trials;       /* least-squares fits per round, 1..20 (one reading of max_local_optimization_number = 20) */
    int32_t  lo_max_calls;    /* local optimisations per run (the other reading: upstream counts INVOCATIONS against
                                 max_local_optimization_number: 20 with a pre-verification, 50 without)                  */
    int32_t  min_iters;       /* the exit rule is not consulted before this many ids (min_iteration_number: 20 / 50)     */
} orc_ransac_params;

/* the parameters as the loop uses them: defaults resolved, the truncated threshold applied */
static orc_ransac_params eff_params(const orc_ransac_params *in)
{
    orc_ransac_params p = *in;
    if (p.scoring == 2) { p.thr2 = p.thr2 * 2.25f; p.scoring = 1; }
    if (p.lo_rounds <= 0) p.lo_rounds = 10;
    if (p.lo_trials <= 0 || p.lo_trials > 20) p.lo_trials = 20;
    if (p.lo_max_calls <= 0) p.lo_max_calls = p.use_elc ? 20 : 50;
    if (p.min_iters <= 0) p.min_iters = p.use_elc ? 20 : 50;
    return p;
}

/* is model (c, q, h) better than (bc, bq, bh)?  msac_T = (uint32)(thr2 * 2^20) for MSAC scoring, else 0 */
static int model_better(uint32_t c, uint64_t q, int64_t h, uint32_t bc, uint64_t bq, int64_t bh, uint32_t msac_T)
{
    if (msac_T == 0u) return c > bc || (c == bc && (q < bq || (q == bq && h < bh)));
    if (c == 0u) return 0;
    if (bc == 0u) return 1;
    const long long k = (long long)c * (long long)msac_T - (long long)q, bk = (long long)bc * (long long)msac_T - (long long)bq;
    return k > bk || (k == bk && h < bh);
}

typedef struct {
    int64_t  best_h;          /* winning hypothesis id, -1 if none                    */
    uint32_t best_count;      /* its inlier count                                     */
    uint64_t best_ssq;        /* sum over inliers of (uint32)(d2 * 2^20)              */
    int64_t  n_valid;         /* hypotheses that passed the pre-check                 */
    int64_t  n_ids;           /* hypothesis ids examined before the run stopped       */
} orc_ransac_result;

/* fp64 minimal-sample Kabsch for hypothesis h; returns 0 when the pre-check rejects it */
static int hypothesis_T(const float *src, const float *tgt, int m, const orc_ransac_params *p,
                        uint64_t h, double T[16], int32_t *sample_out, const int32_t *G)
{
    int32_t s[4];
    draw_sample_prosac(p->seed, h, m, p->sample_size, G, p->prosac_growth > 0 ? p->prosac_growth : 100000, s);
    if (sample_out) memcpy(sample_out, s, sizeof(int32_t) * p->sample_size);
    if (p->sampler != 0)       /* unique-index samplers: a repeated index rejects the draw */
        for (int a = 0; a < p->sample_size; ++a)
            for (int b = a + 1; b < p->sample_size; ++b) if (s[a] == s[b]) return 0;
    if (p->use_elc == 1 && !elc_ok(src, tgt, s, p->sample_size)) return 0;
    double P[12], Q[12];
    for (int k = 0; k < p->sample_size; ++k)
        for (int a = 0; a < 3; ++a) { P[3 * k + a] = (double)src[3 * s[k] + a]; Q[3 * k + a] = (double)tgt[3 * s[k] + a]; }
    orc_kabsch_points(P, Q, NULL, p->sample_size, T);
    return 1;
}

ORC_API int orc_hypothesis(const float *src, const float *tgt, int m, const orc_ransac_params *p,
                           uint64_t h, double T[16], int32_t sample[4])
{
    int32_t *G = p->sampler == 1 ? prosac_table(m, p->sample_size, p->prosac_growth > 0 ? p->prosac_growth : 100000) : NULL;
    const int ok = hypothesis_T(src, tgt, m, p, h, T, sample, G);
    free(G);
    return ok;
}

/* inlier count + fixed-point squared-error sum of one fp32 model over all m correspondences */
static void score_model(const float *src, const float *tgt, int m, const float Rt[12], float thr2,
                        uint32_t *count, uint64_t *ssq)
{
    uint32_t c = 0; uint64_t q = 0;
    for (int i = 0; i < m; ++i) {
        float px = src[3 * i], py = src[3 * i + 1], pz = src[3 * i + 2];
        float x = fmaf(Rt[0], px, fmaf(Rt[1], py, fmaf(Rt[2], pz, Rt[3])));
        float y = fmaf(Rt[4], px, fmaf(Rt[5], py, fmaf(Rt[6], pz, Rt[7])));
        float z = fmaf(Rt[8], px, fmaf(Rt[9], py, fmaf(Rt[10], pz, Rt[11])));
        float dx = x - tgt[3 * i], dy = y - tgt[3 * i + 1], dz = z - tgt[3 * i + 2];
        float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        if (d2 < thr2) { c += 1; q += (uint64_t)(uint32_t)(d2 * 1048576.0f); }
    }
    *count = c; *ssq = q;
}

ORC_API void orc_score(const float *src, const float *tgt, int m, const double T[16], float thr2,
                       uint32_t *count, uint64_t *ssq)
{
    float Rt[12];
    for (int k = 0; k < 12; ++k) Rt[k] = (float)T[k];
    score_model(src, tgt, m, Rt, thr2, count, ssq);
}


/* ------------------------------------------------------------------ SPRT ---- */

/* Natural logarithm from + - * / only: the values the decisions compare against (confidence exit, SPRT design) must be the same
 * bits on the host and on the device, and libm's and ocml's log differ in the last place.  Same text as lr_det_log
 * (lidarregistration_amd/csrc/lr_kabsch.h): x = m 2^e with m in [sqrt(1/2), sqrt(2)), log x = e ln 2 + 2 atanh((m-1)/(m+1)). */
static double det_log(double x)
{
    if (!(x > 0.0)) return x == 0.0 ? -HUGE_VAL : NAN;
    if (x > 1.7976931348623157e308) return HUGE_VAL;      /* +inf */
    unsigned long long b;
    memcpy(&b, &x, 8);
    int e = (int)((b >> 52) & 0x7ffull);
    if (e == 0) { x = x * 18014398509481984.0; memcpy(&b, &x, 8); e = (int)((b >> 52) & 0x7ffull) - 54; }
    e -= 1023;
    b = (b & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m;
    memcpy(&m, &b, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double t = (m - 1.0) / (m + 1.0), t2 = t * t;
    double s = 1.0 / 25.0;
    s = s * t2 + 1.0 / 23.0; s = s * t2 + 1.0 / 21.0; s = s * t2 + 1.0 / 19.0; s = s * t2 + 1.0 / 17.0; s = s * t2 + 1.0 / 15.0;
    s = s * t2 + 1.0 / 13.0; s = s * t2 + 1.0 / 11.0; s = s * t2 + 1.0 / 9.0; s = s * t2 + 1.0 / 7.0; s = s * t2 + 1.0 / 5.0;
    s = s * t2 + 1.0 / 3.0; s = s * t2 + 1.0;
    return (double)e * 0.6931471805599453 + (2.0 * t) * s;
}
ORC_API double orc_det_log(double x) { return det_log(x); }

/* --fast_rejection SPRT (GC_RANSAC.py:29-34 -> use_sprt with min_inlier_ratio_for_sprt = 0.1; gcransac_python.cpp:534-568
 * instantiates GC-RANSAC's SPRTPreemptiveVerfication): Wald's sequential probability ratio test on the model's residuals
 * (Matas & Chum, "Randomized RANSAC with sequential probability ratio test", ICCV 2005; Chum & Matas, PAMI 2008).  The
 * library is not vendored: PARITY UNPINNED; restated from the papers:
 *   lambda = prod over the verified points of  delta/eps (consistent point)  or  (1-delta)/(1-eps) (inconsistent point);
 *   the model is rejected as soon as lambda > A;   A solves A = K + ln A,  K = t_M C / m_S + 1,
 *   C = (1-delta) ln((1-delta)/(1-eps)) + delta ln(delta/eps)   (t_M = 200 verifications per model estimate, m_S = 1 model
 *   per sample);  eps starts at min_inlier_ratio_for_sprt and follows the best model's inlier ratio, delta starts at 0.01
 *   and follows the inlier ratio observed on rejected models (re-designed when it moves by more than 5 %).
 * Restated for a parallel machine: (i) eps / delta / A are frozen inside a batch of hypothesis ids and updated between
 * batches (like the confidence test and the local optimisation); delta = sum of consistent points / sum of verified points
 * over all models rejected so far; (ii) the test runs over the first SPRT_HORIZON correspondences in list order (a bad
 * model is rejected after ~15-70 points; a model that survives the horizon is scored in full and competes as usual).   */
#define SPRT_HORIZON 256
#define SPRT_EPS0 0.1
#define SPRT_DELTA0 0.01

static double sprt_threshold(double eps, double delta)
{
    const double C = (1.0 - delta) * det_log((1.0 - delta) / (1.0 - eps)) + delta * det_log(delta / eps);
    const double K = (200.0 * C) / 1.0 + 1.0;
    double A = K;
    for (int i = 0; i < 10; ++i) A = K + det_log(A);
    return A;
}

/* 1: the model survives the horizon; 0: rejected after *k points of which *inl were consistent */
static int sprt_test(const float *src, const float *tgt, int m, const float Rt[12], float thr2, double eps, double delta, double A,
                     int *k_out, int *inl_out)
{
    const double fin = delta / eps, fout = (1.0 - delta) / (1.0 - eps);
    const int n = m < SPRT_HORIZON ? m : SPRT_HORIZON;
    double lambda = 1.0;
    int inl = 0;
    for (int i = 0; i < n; ++i) {
        float px = src[3 * i], py = src[3 * i + 1], pz = src[3 * i + 2];
        float x = fmaf(Rt[0], px, fmaf(Rt[1], py, fmaf(Rt[2], pz, Rt[3])));
        float y = fmaf(Rt[4], px, fmaf(Rt[5], py, fmaf(Rt[6], pz, Rt[7])));
        float z = fmaf(Rt[8], px, fmaf(Rt[9], py, fmaf(Rt[10], pz, Rt[11])));
        float dx = x - tgt[3 * i], dy = y - tgt[3 * i + 1], dz = z - tgt[3 * i + 2];
        float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        if (d2 < thr2) { inl += 1; lambda = lambda * fin; } else lambda = lambda * fout;
        if (lambda > A) { *k_out = i + 1; *inl_out = inl; return 0; }
    }
    *k_out = n; *inl_out = inl;
    return 1;
}

ORC_API double orc_sprt_threshold(double eps, double delta) { return sprt_threshold(eps, delta); }


/* ------------------------------------------------- local optimisation ---- */
/* GC-RANSAC's local optimisation (Barath & Matas, "Graph-Cut RANSAC", CVPR 2018, Alg. 2; driver GC-RANSAC/src/pygcransac/
 * src/gcransac_python.cpp:404-624 selects it with `neighborhood == 0`, GC_RANSAC.py:36-37) as the reference runs it: spatial
 * coherence weight 0 (GC_RANSAC.py:15, test.py:302), where the graph-cut labelling reduces to its unary term, i.e. to the
 * points within the threshold of the current model.  The library itself (graph-cut-ransac, pygcransac 0.1) is NOT vendored:
 * parity unpinned; constants are those of gcransac_python.cpp:508-515 and the library's defaults.
 *   rounds (<= 10, max_graph_cut_number):  I = inliers of the best model;  inner RANSAC: LO_TRIALS (20,
 *   max_local_optimization_number) least-squares fits on LO_SAMPLE (7 x minimal sample = 21) points drawn uniformly without
 *   repetition from I (all of I when it is not larger: one fit), each scored over ALL correspondences; the best of them
 *   replaces the model if its score is strictly better, otherwise the optimisation stops.
 * The 20 trials of a round depend only on I, so they are independent: that is what the HIP kernel exploits.
 * Final polish (iterated least squares, <= 10 fits): refit on all inliers; a fit that loses inliers is discarded and ends
 * the iteration, a fit with the same number of inliers is kept and ends it (converged), a fit with more is kept and
 * refitted again.                                                                                                       */
#define LO_ROUNDS 10
#define LO_TRIALS 20
#define LO_SAMPLE 21
#define LO_POLISH 10
#define LO_RED 1024         /* fp64 sums run as LO_RED strided partials + a fixed halving tree: the order the HIP block uses */

/* sums v[k] over the listed points in the block-reduction order: partial[t] = sum_{e = t, t+LO_RED, ...}, then the tree */
static void lo_moments(const float *src, const float *tgt, const int32_t *list, int n, double mom[16])
{
    static __thread double part[LO_RED][16];
    for (int t = 0; t < LO_RED; ++t) {
        double v[16];
        for (int k = 0; k < 16; ++k) v[k] = 0.0;
        for (int e = t; e < n; e += LO_RED) {
            const int i = list[e];
            const double p[3] = { src[3 * i], src[3 * i + 1], src[3 * i + 2] }, q[3] = { tgt[3 * i], tgt[3 * i + 1], tgt[3 * i + 2] };
            v[0] += 1.0;
            for (int a = 0; a < 3; ++a) { v[1 + a] += p[a]; v[4 + a] += q[a]; }
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) v[7 + 3 * a + b] += p[a] * q[b];
        }
        for (int k = 0; k < 16; ++k) part[t][k] = v[k];
    }
    for (int sft = LO_RED / 2; sft >= 1; sft >>= 1)
        for (int t = 0; t < sft; ++t) for (int k = 0; k < 16; ++k) part[t][k] += part[t + sft][k];
    for (int k = 0; k < 16; ++k) mom[k] = part[0][k];
}

/* inliers of the fp32 model (the scoring arithmetic), in index order; returns their number */
static int lo_inliers(const float *src, const float *tgt, int m, const double T[16], float thr2, int32_t *list)
{
    float Rt[12];
    for (int k = 0; k < 12; ++k) Rt[k] = (float)T[k];
    int n = 0;
    for (int i = 0; i < m; ++i) {
        float px = src[3 * i], py = src[3 * i + 1], pz = src[3 * i + 2];
        float x = fmaf(Rt[0], px, fmaf(Rt[1], py, fmaf(Rt[2], pz, Rt[3])));
        float y = fmaf(Rt[4], px, fmaf(Rt[5], py, fmaf(Rt[6], pz, Rt[7])));
        float z = fmaf(Rt[8], px, fmaf(Rt[9], py, fmaf(Rt[10], pz, Rt[11])));
        float dx = x - tgt[3 * i], dy = y - tgt[3 * i + 1], dz = z - tgt[3 * i + 2];
        float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        if (d2 < thr2) list[n++] = i;
    }
    return n;
}

static int lo_fit_all(const float *src, const float *tgt, const int32_t *list, int n, double T[16])
{
    double mom[16];
    lo_moments(src, tgt, list, n, mom);
    if (!(mom[0] >= 3.0)) return 0;
    orc_kabsch_moments(mom[0], mom + 1, mom + 4, mom + 7, T);
    return 1;
}

/* LO_SAMPLE distinct positions of [0, n): word w of the stream (seed, call, round, trial) -> position mulhi(word, n), a
 * repeated position is skipped; after 128 words the sample is completed with the lowest positions not yet taken */
static void lo_draw(uint64_t seed, int call, int round, int trial, int n, int32_t *pos)
{
    int got = 0;
    for (int blk = 0; blk < 32 && got < LO_SAMPLE; ++blk) {
        uint32_t w[4];
        orc_philox(seed ^ 0x4c4f43414c4f5054ull, ((uint64_t)call << 40) | ((uint64_t)round << 32) | ((uint64_t)trial << 8) | (uint64_t)blk, w);
        for (int k = 0; k < 4 && got < LO_SAMPLE; ++k) {
            const int32_t c = (int32_t)(((uint64_t)w[k] * (uint64_t)(uint32_t)n) >> 32);
            int dup = 0;
            for (int j = 0; j < got; ++j) if (pos[j] == c) dup = 1;
            if (!dup) pos[got++] = c;
        }
    }
    for (int c = 0; got < LO_SAMPLE; ++c) {
        int dup = 0;
        for (int j = 0; j < got; ++j) if (pos[j] == c) dup = 1;
        if (!dup) pos[got++] = c;
    }
}

/* one local optimisation of (T, c, q); returns 1 when the model was replaced */
static int lo_optimise(const float *src, const float *tgt, int m, const orc_ransac_params *p, int call, double T[16],
                       uint32_t *c, uint64_t *q, int32_t *list)
{
    const uint32_t msac_T = p->scoring == 1 ? (uint32_t)(p->thr2 * 1048576.0f) : 0u;
    int changed = 0;
    for (int round = 0; round < p->lo_rounds; ++round) {
        const int nI = lo_inliers(src, tgt, m, T, p->thr2, list);
        if (nI <= p->sample_size) break;
        const int ntrial = nI > LO_SAMPLE ? p->lo_trials : 1;
        int bt = -1; uint32_t bc = 0; uint64_t bq = 0; double bT[16];
        for (int t = 0; t < ntrial; ++t) {
            double Tt[16];
            if (nI > LO_SAMPLE) {
                int32_t pos[LO_SAMPLE];
                lo_draw(p->seed, call, round, t, nI, pos);
                double P[3 * LO_SAMPLE], Q[3 * LO_SAMPLE];
                for (int k = 0; k < LO_SAMPLE; ++k) {
                    const int i = list[pos[k]];
                    for (int a = 0; a < 3; ++a) { P[3 * k + a] = (double)src[3 * i + a]; Q[3 * k + a] = (double)tgt[3 * i + a]; }
                }
                orc_kabsch_points(P, Q, NULL, LO_SAMPLE, Tt);
            } else if (!lo_fit_all(src, tgt, list, nI, Tt)) break;
            uint32_t tc; uint64_t tq;
            orc_score(src, tgt, m, Tt, p->thr2, &tc, &tq);
            if (tc == 0) continue;
            if (bt < 0 || model_better(tc, tq, t, bc, bq, bt, msac_T)) { bt = t; bc = tc; bq = tq; memcpy(bT, Tt, sizeof(bT)); }
        }
        /* strictly better than the current model (an equal score keeps it: the id of the current model counts as lower) */
        if (bt < 0 || !model_better(bc, bq, 1, *c, *q, 0, msac_T)) break;
        memcpy(T, bT, sizeof(bT)); *c = bc; *q = bq; changed = 1;
    }
    return changed;
}

/* final iterated least squares */
static void lo_polish(const float *src, const float *tgt, int m, const orc_ransac_params *p, double T[16], uint32_t *c, uint64_t *q,
                      int32_t *list)
{
    for (int it = 0; it < LO_POLISH; ++it) {
        const int nI = lo_inliers(src, tgt, m, T, p->thr2, list);
        double Tn[16];
        if (nI <= p->sample_size || !lo_fit_all(src, tgt, list, nI, Tn)) break;
        uint32_t tc; uint64_t tq;
        orc_score(src, tgt, m, Tn, p->thr2, &tc, &tq);
        if (tc < *c) break;
        const int same = tc == *c;
        memcpy(T, Tn, sizeof(Tn)); *c = tc; *q = tq;
        if (same) break;
    }
}

ORC_API void orc_lo_sample(uint64_t seed, int call, int round, int trial, int n, int32_t *pos) { lo_draw(seed, call, round, trial, n, pos); }

/* Hypothesise-and-verify loop with Open3D ordering: more inliers wins, then lower error, then lower h.
 * Ids are processed in batches; after the batch ending at id e the loop stops when
 * e >= log(1-conf)/log(1-(inl/M)^n) for the best model so far (the exit rule of Open3D's
 * RegistrationRANSACBasedOnCorrespondence / GC-RANSAC, applied at batch granularity so that a parallel
 * evaluation is deterministic).  confidence >= 1 disables it: every id is evaluated.              */
ORC_API void orc_ransac(const float *src, const float *tgt, int m, const orc_ransac_params *p_in,
                        double T_best[16], orc_ransac_result *res)
{
    const orc_ransac_params pe = eff_params(p_in), *p = &pe;
    int64_t best_h = -1; uint32_t best_c = 0; uint64_t best_q = 0; int64_t n_valid = 0, n_ids = 0;
    /* with local optimisation the best model is no longer the minimal-sample fit of best_h: it is carried along */
    double Tb[16]; int have_T = 0, lo_calls = 0;
    int32_t *lo_list = p->local_opt ? (int32_t *)malloc(sizeof(int32_t) * (size_t)(m > 0 ? m : 1)) : NULL;
    /* SPRT design of the current batch and the statistics of the rejected models (use_elc == 2) */
    double sprt_eps = SPRT_EPS0, sprt_delta = SPRT_DELTA0;
    uint64_t rej_inl = 0, rej_pts = 0;
    const int use_exit = p->confidence > 0.0f && p->confidence < 1.0f;
    /* batch lengths: the given one, constant; by default 1024, 8192, 65536, ... (eightfold: the exit test is fine-grained where an
     * easy pair stops, and a long run still takes few batches) */
    const int geometric = use_exit && p->batch <= 0;
    int64_t B = use_exit ? (p->batch > 0 ? p->batch : 1024) : (p->iters > 0 ? p->iters : 1);
    int32_t *G = p->sampler == 1 ? prosac_table(m, p->sample_size, p->prosac_growth > 0 ? p->prosac_growth : 100000) : NULL;
    const uint32_t msac_T = p->scoring == 1 ? (uint32_t)(p->thr2 * 1048576.0f) : 0u;
    for (int64_t h0 = 0; h0 < p->iters; h0 += B, B = geometric ? 8 * B : B) {
        const int64_t h1 = h0 + B < p->iters ? h0 + B : p->iters;
        int64_t bb_h = -1; uint32_t bb_c = 0; uint64_t bb_q = 0;      /* winner of this batch */
        const double sprt_A = p->use_elc == 2 ? sprt_threshold(sprt_eps, sprt_delta) : 0.0;
        uint64_t b_inl = 0, b_pts = 0;
#pragma omp parallel
        {
            int64_t lh = -1; uint32_t lc = 0; uint64_t lq = 0; int64_t lv = 0;
            uint64_t l_inl = 0, l_pts = 0;
#pragma omp for schedule(dynamic, 64)
            for (int64_t h = h0; h < h1; ++h) {
                double T[16];
                if (!hypothesis_T(src, tgt, m, p, (uint64_t)h, T, NULL, G)) continue;
                float Rt[12];
                for (int k = 0; k < 12; ++k) Rt[k] = (float)T[k];
                if (p->use_elc == 2) {
                    int kk, ii;
                    if (!sprt_test(src, tgt, m, Rt, p->thr2, sprt_eps, sprt_delta, sprt_A, &kk, &ii)) { l_inl += (uint64_t)ii; l_pts += (uint64_t)kk; continue; }
                }
                lv += 1;
                uint32_t c; uint64_t q;
                score_model(src, tgt, m, Rt, p->thr2, &c, &q);
                if (c == 0) continue;
                if (lh < 0 || model_better(c, q, h, lc, lq, lh, msac_T)) { lh = h; lc = c; lq = q; }
            }
#pragma omp critical
            {
                n_valid += lv; b_inl += l_inl; b_pts += l_pts;
                if (lh >= 0 && (bb_h < 0 || model_better(lc, lq, lh, bb_c, bb_q, bb_h, msac_T))) { bb_h = lh; bb_c = lc; bb_q = lq; }
            }
        }
        rej_inl += b_inl; rej_pts += b_pts;
        /* the batch winner replaces the best so far when it scores better (an optimised model keeps the id of its seed) */
        if (bb_h >= 0 && (best_h < 0 || model_better(bb_c, bb_q, bb_h, best_c, best_q, best_h, msac_T))) {
            best_h = bb_h; best_c = bb_c; best_q = bb_q;
            hypothesis_T(src, tgt, m, p, (uint64_t)best_h, Tb, NULL, G); have_T = 1;
            if (p->local_opt == 1 && lo_calls < p->lo_max_calls) { lo_optimise(src, tgt, m, p, lo_calls, Tb, &best_c, &best_q, lo_list); lo_calls += 1; }
        }
        n_ids = h1;
        if (p->use_elc == 2) {
            /* re-design the test for the next batch: eps follows the best model, delta the rejected ones */
            if (best_c > 0) { const double e = (double)best_c / (double)m; if (e > sprt_eps && e < 1.0) sprt_eps = e; }
            if (rej_pts > 0) {
                const double d = (double)rej_inl / (double)rej_pts;
                if (d > 0.0 && d < 0.9 * sprt_eps && fabs(d - sprt_delta) > 0.05 * sprt_delta) sprt_delta = d;
            }
            if (!(sprt_delta < 0.9 * sprt_eps)) sprt_delta = 0.9 * sprt_eps * 0.5;
        }
        if (use_exit && best_c > 0) {
            double f = (double)best_c / (double)m;
            double fn = f;
            for (int k = 1; k < p->sample_size; ++k) fn = fn * f;
            double kk = det_log(1.0 - (double)p->confidence) / det_log(1.0 - fn);
            if ((double)h1 >= kk && h1 >= (int64_t)p->min_iters) break;
        }
    }
    for (int k = 0; k < 16; ++k) T_best[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (best_h >= 0 && have_T) {
        if (p->local_opt) lo_polish(src, tgt, m, p, Tb, &best_c, &best_q, lo_list);
        memcpy(T_best, Tb, sizeof(Tb));
    }
    free(G); free(lo_list);
    res->best_h = best_h; res->best_count = best_c; res->best_ssq = best_q; res->n_valid = n_valid; res->n_ids = n_ids;
}

/* SEQUENTIAL-SEMANTICS REFERENCE MODE.  The loop above (and the HIP kernels) evaluate hypothesis ids in batches: exit rule, local
 * optimisation and SPRT re-design happen between batches of >= 8192 ids, and the SPRT looks at the first 256 correspondences only.
 * The third-party loops behind the reference work per iteration (Open3D RegistrationRANSACBasedOnCorrespondence, FR.py:128-137;
 * GCRANSAC::run behind GC_RANSAC.py:24-37 / gcransac_python.cpp:513-517).  This function restates THAT order over the same
 * hypothesis stream (id h -> the same sample, pre-check and model):
 *   for h = 0, 1, ...: stop when h >= min_iters and h >= k(best) = log(1-conf)/log(1-(inl/M)^n)   [GCRANSAC::run's while condition]
 *     sample h -> pre-check -> model -> (SPRT: Wald's test over ALL correspondences, re-designed after every rejected model and
 *     every new best) -> score -> a strictly better model becomes the best at once, is locally optimised at once (at most
 *     lo_max_calls times) and k is updated from the optimised model
 *   then the final iterated least squares.
 * tests/test_oracle_ransac.py measures how far the batched result is from this one (final T after the last least squares) and how
 * many more ids the batched loop examines.  n_ids = ids examined.                                                                */
ORC_API void orc_ransac_seq(const float *src, const float *tgt, int m, const orc_ransac_params *p_in,
                            double T_best[16], orc_ransac_result *res)
{
    const orc_ransac_params pe = eff_params(p_in), *p = &pe;
    int64_t best_h = -1; uint32_t best_c = 0; uint64_t best_q = 0; int64_t n_valid = 0, n_ids = 0;
    double Tb[16]; int have_T = 0, lo_calls = 0;
    int32_t *lo_list = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m > 0 ? m : 1));
    double sprt_eps = SPRT_EPS0, sprt_delta = SPRT_DELTA0, sprt_A = p->use_elc == 2 ? sprt_threshold(SPRT_EPS0, SPRT_DELTA0) : 0.0;
    uint64_t rej_inl = 0, rej_pts = 0;
    const int use_exit = p->confidence > 0.0f && p->confidence < 1.0f;
    int32_t *G = p->sampler == 1 ? prosac_table(m, p->sample_size, p->prosac_growth > 0 ? p->prosac_growth : 100000) : NULL;
    const uint32_t msac_T = p->scoring == 1 ? (uint32_t)(p->thr2 * 1048576.0f) : 0u;
    double k_needed = INFINITY;
    int64_t h = 0;
    for (; h < p->iters; ++h) {
        if (use_exit && h >= (int64_t)p->min_iters && (double)h >= k_needed) break;
        double T[16];
        if (!hypothesis_T(src, tgt, m, p, (uint64_t)h, T, NULL, G)) continue;
        float Rt[12];
        for (int k = 0; k < 12; ++k) Rt[k] = (float)T[k];
        if (p->use_elc == 2) {
            /* Wald's test to the end of the list */
            const double fin = sprt_delta / sprt_eps, fout = (1.0 - sprt_delta) / (1.0 - sprt_eps);
            double lambda = 1.0; int inl = 0, rejected = 0, kk = m;
            for (int i = 0; i < m; ++i) {
                float px = src[3 * i], py = src[3 * i + 1], pz = src[3 * i + 2];
                float x = fmaf(Rt[0], px, fmaf(Rt[1], py, fmaf(Rt[2], pz, Rt[3])));
                float y = fmaf(Rt[4], px, fmaf(Rt[5], py, fmaf(Rt[6], pz, Rt[7])));
                float z = fmaf(Rt[8], px, fmaf(Rt[9], py, fmaf(Rt[10], pz, Rt[11])));
                float dx = x - tgt[3 * i], dy = y - tgt[3 * i + 1], dz = z - tgt[3 * i + 2];
                float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
                if (d2 < p->thr2) { inl += 1; lambda = lambda * fin; } else lambda = lambda * fout;
                if (lambda > sprt_A) { rejected = 1; kk = i + 1; break; }
            }
            if (rejected) {
                rej_inl += (uint64_t)inl; rej_pts += (uint64_t)kk;
                const double d = (double)rej_inl / (double)rej_pts;
                if (d > 0.0 && d < 0.9 * sprt_eps && fabs(d - sprt_delta) > 0.05 * sprt_delta) { sprt_delta = d; sprt_A = sprt_threshold(sprt_eps, sprt_delta); }
                continue;
            }
        }
        n_valid += 1;
        uint32_t c; uint64_t q;
        score_model(src, tgt, m, Rt, p->thr2, &c, &q);
        if (c == 0) continue;
        if (best_h < 0 || model_better(c, q, h, best_c, best_q, best_h, msac_T)) {
            best_h = h; best_c = c; best_q = q; memcpy(Tb, T, sizeof(Tb)); have_T = 1;
            if (p->local_opt == 1 && lo_calls < p->lo_max_calls) { lo_optimise(src, tgt, m, p, lo_calls, Tb, &best_c, &best_q, lo_list); lo_calls += 1; }
            if (p->use_elc == 2) {
                const double e = (double)best_c / (double)m;
                if (e > sprt_eps && e < 1.0) { sprt_eps = e; if (!(sprt_delta < 0.9 * sprt_eps)) sprt_delta = 0.9 * sprt_eps * 0.5; sprt_A = sprt_threshold(sprt_eps, sprt_delta); }
            }
            if (use_exit) {
                double f = (double)best_c / (double)m, fn = f;
                for (int k = 1; k < p->sample_size; ++k) fn = fn * f;
                k_needed = det_log(1.0 - (double)p->confidence) / det_log(1.0 - fn);
            }
        }
    }
    n_ids = h;
    for (int k = 0; k < 16; ++k) T_best[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (best_h >= 0 && have_T) {
        if (p->local_opt) lo_polish(src, tgt, m, p, Tb, &best_c, &best_q, lo_list);
        memcpy(T_best, Tb, sizeof(Tb));
    }
    free(G); free(lo_list);
    res->best_h = best_h; res->best_count = best_c; res->best_ssq = best_q; res->n_valid = n_valid; res->n_ids = n_ids;
}

/* --------------------------------------------------------------- refit ---- */

/* FR.py:99-111: inliers of T over the ORIGINAL nn pairs (i, idx1[i]) within thr2 (fp64), then a
 * least-squares rigid fit on them.  Returns the inlier count; T_out = T_in when fewer than 3.     */
static int refit_impl(const float *xyz0, int n0, const float *xyz1, const int32_t *idx1,
                      const double T_in[16], double thr2, double T_out[16], const float *F0, const float *F1);

ORC_API int orc_refit(const float *xyz0, int n0, const float *xyz1, const int32_t *idx1,
                      const double T_in[16], double thr2, double T_out[16])
{
    return refit_impl(xyz0, n0, xyz1, idx1, T_in, thr2, T_out, NULL, NULL);
}

/* DGR/core/deep_global_registration.py:519-537: the same inlier set, weighted by 1 / ||F0[i] - F1[idx1[i]]|| */
ORC_API int orc_refit_weighted(const float *xyz0, int n0, const float *xyz1, const int32_t *idx1, const double T_in[16],
                               double thr2, double T_out[16], const float *F0, const float *F1)
{
    return refit_impl(xyz0, n0, xyz1, idx1, T_in, thr2, T_out, F0, F1);
}

static int refit_impl(const float *xyz0, int n0, const float *xyz1, const int32_t *idx1,
                      const double T_in[16], double thr2, double T_out[16], const float *F0, const float *F1)
{
    double n = 0.0, sp[3] = {0,0,0}, sq[3] = {0,0,0}, spq[9] = {0,0,0,0,0,0,0,0,0};
    int cnt = 0;
    for (int i = 0; i < n0; ++i) {
        double p[3] = { xyz0[3 * i], xyz0[3 * i + 1], xyz0[3 * i + 2] };
        int j = idx1[i];
        double q[3] = { xyz1[3 * j], xyz1[3 * j + 1], xyz1[3 * j + 2] };
        double r[3];
        for (int a = 0; a < 3; ++a)
            r[a] = (((T_in[4 * a] * p[0] + T_in[4 * a + 1] * p[1]) + T_in[4 * a + 2] * p[2]) + T_in[4 * a + 3]) - q[a];
        double d2 = (r[0] * r[0] + r[1] * r[1]) + r[2] * r[2];
        if (d2 < thr2) {
            double w = 1.0;
            if (F0) {
                const float *fa = F0 + (size_t)i * 32, *fb = F1 + (size_t)j * 32;
                float acc = 0.0f;
                for (int k = 0; k < 32; ++k) { float e = fa[k] - fb[k]; float q2 = e * e; acc = acc + q2; }
                w = 1.0 / (double)fmaxf(sqrtf(acc), 1e-12f);
            }
            n += w; cnt += 1;
            for (int a = 0; a < 3; ++a) { sp[a] += w * p[a]; sq[a] += w * q[a]; }
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) spq[3 * a + b] += (w * p[a]) * q[b];
        }
    }
    if (!(n > 0.0) || (!F0 && n < 3.0)) { memcpy(T_out, T_in, sizeof(double) * 16); return cnt; }
    orc_kabsch_moments(n, sp, sq, spq, T_out);
    return cnt;
}

/* ----------------------------------------------------------------- ICP ---- */

/* Point-to-point ICP as Open3D 0.13.0's RegistrationICP runs it for Experiments/test.py:183-189 (third-party, not
 * vendored: parity unpinned): nearest target within max_dist (strict) of every transformed source point, Umeyama update,
 * at most max_iter updates, stop when fitness and inlier RMSE both move by less than the relative criteria.
 * Brute-force neighbour search (test sizes only).  out[0]=fitness, out[1]=rmse, out[2]=n_corr, out[3]=updates.     */
ORC_API void orc_icp(const float *src, int n0, const float *tgt, int n1, const double T_init[16], double max_dist,
                     int max_iter, double rel_fit, double rel_rmse, double T_out[16], double out[4])
{
    double T[16];
    memcpy(T, T_init, sizeof(T));
    double prev_fit = 0.0, prev_rmse = 0.0;
    const double max_d2 = max_dist * max_dist;
    int k = 0;
    for (;; ++k) {
        double n = 0.0, sp[3] = {0,0,0}, sq[3] = {0,0,0}, spq[9] = {0,0,0,0,0,0,0,0,0}, err2 = 0.0;
#pragma omp parallel
        {
            double ln = 0.0, lsp[3] = {0,0,0}, lsq[3] = {0,0,0}, lspq[9] = {0,0,0,0,0,0,0,0,0}, lerr = 0.0;
#pragma omp for schedule(static)
            for (int i = 0; i < n0; ++i) {
                double px = src[3 * i], py = src[3 * i + 1], pz = src[3 * i + 2], p[3];
                for (int a = 0; a < 3; ++a) p[a] = ((T[4 * a] * px + T[4 * a + 1] * py) + T[4 * a + 2] * pz) + T[4 * a + 3];
                double best = max_d2; int bj = -1;
                for (int j = 0; j < n1; ++j) {
                    double qx = (double)tgt[3 * j] - p[0], qy = (double)tgt[3 * j + 1] - p[1], qz = (double)tgt[3 * j + 2] - p[2];
                    double d2 = (qx * qx + qy * qy) + qz * qz;
                    if (d2 < best) { best = d2; bj = j; }
                }
                if (bj >= 0) {
                    double q[3] = { tgt[3 * bj], tgt[3 * bj + 1], tgt[3 * bj + 2] };
                    ln += 1.0; lerr += best;
                    for (int a = 0; a < 3; ++a) { lsp[a] += p[a]; lsq[a] += q[a]; }
                    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) lspq[3 * a + b] += p[a] * q[b];
                }
            }
#pragma omp critical
            {
                n += ln; err2 += lerr;
                for (int a = 0; a < 3; ++a) { sp[a] += lsp[a]; sq[a] += lsq[a]; }
                for (int a = 0; a < 9; ++a) spq[a] += lspq[a];
            }
        }
        double fit = n / (double)n0, rmse = n > 0.0 ? sqrt(err2 / n) : 0.0;
        out[0] = fit; out[1] = rmse; out[2] = n;
        int done = 0;
        if (k > 0 && fabs(prev_fit - fit) < rel_fit && fabs(prev_rmse - rmse) < rel_rmse) done = 1;
        if (k >= max_iter || n < 3.0) done = 1;
        if (done) break;
        double U[16], Tn[16];
        orc_kabsch_moments(n, sp, sq, spq, U);
        for (int a = 0; a < 3; ++a) {
            for (int b = 0; b < 3; ++b) Tn[4 * a + b] = (U[4 * a] * T[b] + U[4 * a + 1] * T[4 + b]) + U[4 * a + 2] * T[8 + b];
            Tn[4 * a + 3] = ((U[4 * a] * T[3] + U[4 * a + 1] * T[7]) + U[4 * a + 2] * T[11]) + U[4 * a + 3];
        }
        for (int q = 0; q < 12; ++q) T[q] = Tn[q];
        prev_fit = fit; prev_rmse = rmse;
    }
    memcpy(T_out, T, sizeof(T));
    T_out[12] = 0; T_out[13] = 0; T_out[14] = 0; T_out[15] = 1;
    out[3] = (double)k;
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
