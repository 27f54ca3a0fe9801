// fp64 Kabsch through Horn's quaternion form (largest eigenvector in closed form; cyclic Jacobi for degenerate input), written with + - * / sqrt only and an explicit op order so
// that the device result is bit-identical to oracle/oracle.c (both compiled with -ffp-contract=off).
// Reference semantics: R = V diag(1,1,det) U^T, t = mu_B - R mu_A (Experiments/models/common.py:7-45).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

// ------------------------------------------------------------------ logarithm for the decisions (fp64, + - * / only)
// The confidence exit and the SPRT design compare against values made of logarithms; libm's and ocml's log differ in the last
// place, which can flip such a comparison on one side only (found by tools/soak_gc.py).  Same text as oracle.c (det_log):
// x = m 2^e with m in [sqrt(1/2), sqrt(2)), log x = e ln 2 + 2 atanh((m - 1) / (m + 1)), the odd series up to t^25.
__host__ __device__ __forceinline__ double lr_det_log(double x)
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

// ------------------------------------------------------------------ Kabsch (fp64, + - * / sqrt only)
#define LR_JACOBI_SWEEPS 10      // upper bound; sweeps stop once the off-diagonal mass is below 1e-15 of the diagonal

__device__ __forceinline__ void lr_jacobi4_maxvec(double A[4][4], double q[4])
{
    double V[4][4] = { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } };
    for (int sweep = 0; sweep < LR_JACOBI_SWEEPS; ++sweep) {
        double off2 = ((((A[0][1] * A[0][1] + A[0][2] * A[0][2]) + A[0][3] * A[0][3]) + A[1][2] * A[1][2]) + A[1][3] * A[1][3]) + A[2][3] * A[2][3];
        double dia2 = ((A[0][0] * A[0][0] + A[1][1] * A[1][1]) + A[2][2] * A[2][2]) + A[3][3] * A[3][3];
        if (off2 <= 1e-30 * dia2) break;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int r = p + 1; r < 4; ++r) {
                double apq = A[p][r];
                if (apq != 0.0) {
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
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (k == p || k == r) continue;
                        double g = A[k][p], f = A[k][r];
                        double gn = g - s * (f + g * tau);
                        double fn = f + s * (g - f * tau);
                        A[k][p] = gn; A[p][k] = gn;
                        A[k][r] = fn; A[r][k] = fn;
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        double g = V[k][p], f = V[k][r];
                        V[k][p] = g - s * (f + g * tau);
                        V[k][r] = f + s * (g - f * tau);
                    }
                }
            }
    }
    double w = V[0][0], x = V[1][0], y = V[2][0], z = V[3][0], best = A[0][0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (A[k][k] > best) { best = A[k][k]; w = V[0][k]; x = V[1][k]; y = V[2][k]; z = V[3][k]; }
    double nn = sqrt(((w * w + x * x) + y * y) + z * z);
    q[0] = w / nn; q[1] = x / nn; q[2] = y / nn; q[3] = z / nn;
}

// Largest-eigenvalue eigenvector in closed form: Newton on the characteristic polynomial from Gershgorin's bound + the adjugate column with the
// largest diagonal cofactor (oracle/oracle.c, horn4_maxvec_newton: same text, same bits; derivation and accuracy there).  0: the caller runs
// Jacobi (zero / non-finite matrix, or a double largest eigenvalue).
__device__ __forceinline__ double lr_det3_(double a, double b, double c, double d, double e, double f, double g, double h, double i)
{
    return (a * (e * i - f * h) - b * (d * i - f * g)) + c * (d * h - e * g);
}
__device__ __forceinline__ int lr_horn4_maxvec_newton(const double N[4][4], double q[4])
{
    const double a = N[0][0], b = N[1][1], c = N[2][2], d = N[3][3];
    const double n01 = N[0][1], n02 = N[0][2], n03 = N[0][3], n12 = N[1][2], n13 = N[1][3], n23 = N[2][3];
    /* elementary symmetric functions of the eigenvalues: trace, principal 2x2 and 3x3 minors, determinant */
    const double e1 = (a + b) + (c + d);
    const double e2 = (((a * b - n01 * n01) + (a * c - n02 * n02)) + ((a * d - n03 * n03) + (b * c - n12 * n12))) + ((b * d - n13 * n13) + (c * d - n23 * n23));
    const double m0 = lr_det3_(b, n12, n13, n12, c, n23, n13, n23, d);      /* without row / column 0 */
    const double m1 = lr_det3_(a, n02, n03, n02, c, n23, n03, n23, d);
    const double m2 = lr_det3_(a, n01, n03, n01, b, n13, n03, n13, d);
    const double m3 = lr_det3_(a, n01, n02, n01, b, n12, n02, n12, c);
    const double e3 = (m0 + m1) + (m2 + m3);
    /* det N by the first row */
    const double k1 = lr_det3_(n01, n12, n13, n02, c, n23, n03, n23, d);
    const double k2 = lr_det3_(n01, b, n13, n02, n12, n23, n03, n13, d);
    const double k3 = lr_det3_(n01, b, n12, n02, n12, c, n03, n13, n23);
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
    const double c00 = lr_det3_(B, n12, n13, n12, C, n23, n13, n23, D);
    const double c11 = lr_det3_(A, n02, n03, n02, C, n23, n03, n23, D);
    const double c22 = lr_det3_(A, n01, n03, n01, B, n13, n03, n13, D);
    const double c33 = lr_det3_(A, n01, n02, n01, B, n12, n02, n12, C);
    const double c01 = -lr_det3_(n01, n12, n13, n02, C, n23, n03, n23, D);
    const double c02 = lr_det3_(n01, B, n13, n02, n12, n23, n03, n13, D);
    const double c03 = -lr_det3_(n01, B, n12, n02, n12, C, n03, n13, n23);
    const double c12 = -lr_det3_(A, n01, n03, n02, n12, n23, n03, n13, D);
    const double c13 = lr_det3_(A, n01, n02, n02, n12, C, n03, n13, n23);
    const double c23 = -lr_det3_(A, n01, n02, n01, B, n12, n03, n13, n23);
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

// H[a][b] = sum (p-cp)_a (q-cq)_b  ->  T (row-major 4x4, q ~ R p + t)
__device__ __forceinline__ void lr_rt_from_cov(const double H[3][3], const double cp[3], const double cq[3], double T[16])
{
    double Sxx = H[0][0], Sxy = H[0][1], Sxz = H[0][2];
    double Syx = H[1][0], Syy = H[1][1], Syz = H[1][2];
    double Szx = H[2][0], Szy = H[2][1], Szz = H[2][2];
    double N[4][4];
    N[0][0] = (Sxx + Syy) + Szz; N[0][1] = Syz - Szy;         N[0][2] = Szx - Sxz;         N[0][3] = Sxy - Syx;
    N[1][0] = N[0][1];           N[1][1] = (Sxx - Syy) - Szz; N[1][2] = Sxy + Syx;         N[1][3] = Szx + Sxz;
    N[2][0] = N[0][2];           N[2][1] = N[1][2];           N[2][2] = (Syy - Sxx) - Szz; N[2][3] = Syz + Szy;
    N[3][0] = N[0][3];           N[3][1] = N[1][3];           N[3][2] = N[2][3];           N[3][3] = (Szz - Sxx) - Syy;
    double q[4];
    if (!lr_horn4_maxvec_newton(N, q)) lr_jacobi4_maxvec(N, q);
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double R[3][3];
    R[0][0] = 1.0 - 2.0 * (y * y + z * z); R[0][1] = 2.0 * (x * y - w * z);       R[0][2] = 2.0 * (x * z + w * y);
    R[1][0] = 2.0 * (x * y + w * z);       R[1][1] = 1.0 - 2.0 * (x * x + z * z); R[1][2] = 2.0 * (y * z - w * x);
    R[2][0] = 2.0 * (x * z - w * y);       R[2][1] = 2.0 * (y * z + w * x);       R[2][2] = 1.0 - 2.0 * (x * x + y * y);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double rc = (R[a][0] * cp[0] + R[a][1] * cp[1]) + R[a][2] * cp[2];
        T[4 * a + 0] = R[a][0]; T[4 * a + 1] = R[a][1]; T[4 * a + 2] = R[a][2];
        T[4 * a + 3] = cq[a] - rc;
    }
    T[12] = 0.0; T[13] = 0.0; T[14] = 0.0; T[15] = 1.0;
}

