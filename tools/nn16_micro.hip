// Timing harness for the nn16 pass A / pass B kernels (development tool).  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/nn16_micro.hip -o tools/bin/nn16_micro
#include "../lidarregistration_amd/csrc/lr_nn16.hip"
#include <vector>
#include <random>
#include <algorithm>
void lr_set_error(const char *, ...) {}
int lr_nn_fix_rows(lr_workspace *, const float *, const float *, const float *, const float *, int, int32_t *, int32_t *, float *, float *, hipStream_t) { return 0; }
template <class F> float timeit(F f, int reps = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) f();
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < reps; ++r) { hipEventRecord(e0, 0); f(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
    return best;
}
int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 30000;
    int strips = argc > 2 ? atoi(argv[2]) : 8;
    std::vector<float> h((size_t)n * 32);
    std::mt19937 rng(1); std::normal_distribution<float> nd;
    for (size_t r = 0; r < (size_t)n; ++r) { double s = 0; for (int k = 0; k < 32; ++k) { h[r*32+k] = nd(rng); s += h[r*32+k]*h[r*32+k]; } for (int k = 0; k < 32; ++k) h[r*32+k] /= (float)sqrt(s); }
    float *F, *Fp, *nrm, *pu1, *pu2, *tau; _Float16 *H; uint32_t *mx; int32_t *cnt, *cand;
    hipMalloc(&F, (size_t)n*128); hipMalloc(&Fp, (size_t)n*128); hipMalloc(&H, (size_t)n*64); hipMalloc(&nrm, n*4); hipMalloc(&mx, 8);
    hipMalloc(&pu1, (size_t)n*4*8); hipMalloc(&pu2, (size_t)n*4*8); hipMalloc(&tau, n*4); hipMalloc(&cnt, n*4*8); hipMalloc(&cand, (size_t)n*4*8*16);
    hipMemcpy(F, h.data(), (size_t)n*128, hipMemcpyHostToDevice);
    hipMemset(mx, 0, 8);
    float *bmax; hipMalloc(&bmax, (n/32+2)*4);
    hipLaunchKernelGGL(nn16_prep_kernel, dim3((n+31)/32), dim3(256), 0, 0, F, n, H, nrm, bmax, F, 0, H, nrm, bmax, (uint32_t*)nullptr, (int32_t*)nullptr, 0);
    int ntiles = (n + 31) / 32;
    int row_blocks = (n + LR_BLOCK_ROWS - 1) / LR_BLOCK_ROWS;
    for (int stride : {1, 2, 4}) {
        int tps = ((ntiles + strips - 1) / strips + stride - 1) / stride * stride;
        dim3 grid(row_blocks, strips);
        float ms = timeit([&] { hipLaunchKernelGGL(nn16_passa_kernel, grid, dim3(256), 0, 0, H, n, H, nrm, n, tps, stride, n, pu1, pu2, cnt); });
        printf("passA stride %d strips %d: %.3f ms\n", stride, strips, ms);
        {   // check: 2nd largest g of a few rows against a host computation over the same sampled columns
            std::vector<float> h1((size_t)n * strips), h2((size_t)n * strips);
            hipMemcpy(h1.data(), pu1, h1.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), pu2, h2.size() * 4, hipMemcpyDeviceToHost);
            for (int row : {0, 1, 77, 12345}) {
                float a1 = -1e30f, a2 = -1e30f;
                for (int s2 = 0; s2 < strips; ++s2) { float c1 = h1[(size_t)s2 * n + row], c2 = h2[(size_t)s2 * n + row]; float hi = fmaxf(a1, c1), lo = fminf(a1, c1); a2 = fmaxf(lo, fmaxf(a2, c2)); a1 = hi; }
                float r1 = -1e30f, r2 = -1e30f;
                for (int sp = 0; sp < strips; ++sp)
                    for (int t = sp * tps; t < (sp + 1) * tps && t < ntiles; t += stride)
                        for (int j = t * 32; j < t * 32 + 32 && j < n; ++j) {
                            double d = 0, nn = 0; for (int k = 0; k < 32; ++k) { d += (double)h[(size_t)row*32+k] * h[(size_t)j*32+k]; nn += (double)h[(size_t)j*32+k] * h[(size_t)j*32+k]; }
                            float g = (float)(d - 0.5 * nn);
                            if (g > r1) { r2 = r1; r1 = g; } else if (g > r2) r2 = g;
                        }
                printf("   row %5d  device (%.5f, %.5f)  host (%.5f, %.5f)\n", row, a1, a2, r1, r2);
            }
        }
        // thresholds from this pass A, then pass B
        lr_thr_in thr = { pu1, pu2, nrm, bmax, strips, n, (n+31)/32, 2 };
        int tpsb = (ntiles + strips - 1) / strips;
        float msp = timeit([&] { hipLaunchKernelGGL(nn16_passb_kernel, grid, dim3(256), 0, 0, H, n, (const int32_t*)nullptr, (const int32_t*)nullptr, H, nrm, n, tpsb, (const float*)nullptr, cnt, cand, (const int32_t*)nullptr, (const float*)nullptr, (const uint32_t*)nullptr, thr); });
        std::vector<int32_t> c1(n);
        hipMemset(cnt, 0, n*4);
        hipLaunchKernelGGL(nn16_passb_kernel, grid, dim3(256), 0, 0, H, n, (const int32_t*)nullptr, (const int32_t*)nullptr, H, nrm, n, tpsb, (const float*)nullptr, cnt, cand, (const int32_t*)nullptr, (const float*)nullptr, (const uint32_t*)nullptr, thr);
        hipMemcpy(c1.data(), cnt, n*4, hipMemcpyDeviceToHost);
        double tot = 0; int mxc = 0;
        for (int i = 0; i < n; ++i) { tot += c1[i]; mxc = c1[i] > mxc ? c1[i] : mxc; }
        printf("passB (thresholds from stride %d): %.3f ms   candidates/row avg %.2f max %d\n", stride, msp, tot / n, mxc);
    }
    // pass B with no candidates at all (pure fast path)
    {
        std::vector<float> t(n, -1e30f); hipMemcpy(tau, t.data(), n*4, hipMemcpyHostToDevice);
        int tpsb = (ntiles + strips - 1) / strips; dim3 grid(row_blocks, strips);
        float msp = timeit([&] { hipLaunchKernelGGL(nn16_passb_kernel, grid, dim3(256), 0, 0, H, n, (const int32_t*)nullptr, (const int32_t*)nullptr, H, nrm, n, tpsb, (const float*)tau, cnt, cand, (const int32_t*)nullptr, (const float*)nullptr, (const uint32_t*)nullptr, lr_thr_in{}); });
        printf("passB no candidates: %.3f ms\n", msp);
    }
    return 0;
}
