// Timing harness for the batched filter pass (nn16_passb_kernel: sample phase + walk) (development tool): P identical pairs in P arenas.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 [-DLR_PB_PROBE] tools/pb_micro.hip -o tools/bin/pb_micro
//   (-DLR_PB_PROBE: hit statistics.  The round-5 ablation variants -- -DLR_PB_EXP=... -- are built from that round's source: tools/pb_variant.sh)
//   usage: pb_micro [n=30000] [P=32] [strips=1]
#ifndef PB_SRC
#define PB_SRC "../lidarregistration_amd/csrc/lr_nn16.hip"
#endif
#include PB_SRC
#ifdef PB_OLD_ABI      // kernels of round 3: no pooled-threshold argument
#define PB_YS
#define PB_YS_ARG
#define PB_BMIN_ARG(p)
#define PB_NEW_ABI 0
#else
#define PB_YS ysh, (int32_t *)nullptr,
#define PB_YS_ARG (uint32_t*)nullptr,
#define PB_BMIN_ARG(p) (float *)((char *)bmin + (p) * stride),
#define PB_NEW_ABI 1
#endif
#include <string.h>
#include <vector>
#include <algorithm>
#include <random>
void lr_set_error(const char *, ...) {}
static void (*g_pre)() = nullptr;      // untimed set-up before every timed call (resets of what the prep kernel initialises in the library)
template <class F> float timeit(F f, int reps = 8) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) { if (g_pre) g_pre(); f(); }
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < reps; ++r) { if (g_pre) { g_pre(); hipDeviceSynchronize(); } hipEventRecord(e0, 0); f(); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
    return best;
}
static void print_clk(int) {}      // (round 5 printed the per-block shader clock of an -DLR_PB_EXP=8 build here; the library has lr_workspace_clock now)
#if PB_NEW_ABI
#define PB_THR(name, nq, need, ss) lr_thr_in name = { nq, nrange, need, ss }
#define PB_LAUNCH(...) do { hipLaunchKernelGGL(nn16_passb_kernel<true>, __VA_ARGS__); hipLaunchKernelGGL(nn16_passb_kernel<false>, __VA_ARGS__); } while (0)
#else
#define PB_THR(name, nq, need, ss) lr_thr_in name = { nq, bmax, (n+31)/32, need, ss }
#define PB_LAUNCH(...) hipLaunchKernelGGL(nn16_passb_kernel, __VA_ARGS__)
#endif
int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 30000, P = argc > 2 ? atoi(argv[2]) : 32, strips = argc > 3 ? atoi(argv[3]) : 1;
    std::vector<float> h((size_t)n * 32);
    std::mt19937 rng(1); std::normal_distribution<float> nd;
    for (size_t r = 0; r < (size_t)n; ++r) { double s = 0; for (int k = 0; k < 32; ++k) { h[r*32+k] = nd(rng); s += h[r*32+k]*h[r*32+k]; } for (int k = 0; k < 32; ++k) h[r*32+k] /= (float)sqrt(s); }
    // arena layout (bytes): H | nrm | bmax | pu1 | pu2 | tau | cnt | cand
    size_t off = 0; auto take = [&](size_t b) { size_t o = off; off = (off + b + 255) & ~size_t(255); return o; };
    const size_t oH = take((size_t)n * 64), oN = take((size_t)n * 4), oB = take((size_t)(n / 32 + 2) * 4), oBm = take((size_t)(n / 32 + 2) * 4), oR = take(64), o1 = take((size_t)n * 4 * 8), o2 = take((size_t)n * 4 * 8),
                 oT = take((size_t)n * 4), oY = take((size_t)n * 4 * 8), oS = take((size_t)n * 4), oA = take((size_t)(n / 256 + 2) * 4), oC = take(LR_NN16_CNT_INTS(n) * 4), oD = take(LR_NN16_SEG_INTS(n) * 4);
    const size_t stride = off;
    char *base; hipMalloc(&base, stride * P); hipMemset(base, 0, stride * P);
    float *F; hipMalloc(&F, (size_t)n * 128); hipMemcpy(F, h.data(), (size_t)n * 128, hipMemcpyHostToDevice);
    _Float16 *H = (_Float16 *)(base + oH); float *nrm = (float *)(base + oN), *bmax = (float *)(base + oB), *bmin = (float *)(base + oBm), *nrange = (float *)(base + oR), *pu1 = (float *)(base + o1), *pu2 = (float *)(base + o2), *tau = (float *)(base + oT);
    int32_t *cnt = (int32_t *)(base + oC), *cand = (int32_t *)(base + oD);
    uint32_t *ysh = (uint32_t *)(base + oS); (void)ysh;
    int32_t *arr = (int32_t *)(base + oA); (void)arr;
    lr_zargs z0 = { 0, nullptr };
    for (int p = 0; p < P; ++p)
        hipLaunchKernelGGL(nn16_prep_kernel, dim3((n+31)/32), dim3(256), 0, 0, F, n, (_Float16 *)((char *)H + p * stride), (float *)((char *)nrm + p * stride), (float *)((char *)bmax + p * stride), PB_BMIN_ARG(p)
                           F, 0, H, nrm, bmax, PB_BMIN_ARG(0) (uint32_t*)nullptr, (unsigned long long*)nullptr, (int32_t*)nullptr, 0, (n + 31) / 32, PB_YS_ARG z0);
#if PB_NEW_ABI
    for (int p = 0; p < P; ++p)
        hipLaunchKernelGGL(nn16_range_kernel, dim3(1), dim3(256), 0, 0, n, (const float *)((char *)bmax + p * stride), (const float *)((char *)bmin + p * stride), 0, (const float *)bmax,
                           (const float *)bmin, (float *)((char *)nrange + p * stride), (int32_t *)nullptr, z0);
#endif
    (void)bmin; (void)nrange;
    hipDeviceSynchronize();
    lr_zargs z = { stride, nullptr };
    PB_THR(thr0, nullptr, 2, 16);
    const int ntiles = (n + 31) / 32, row_blocks = (n + LR_BLOCK_ROWS - 1) / LR_BLOCK_ROWS;
    const int total = row_blocks * strips * P;
    dim3 grid(8 * ((total + 7) / 8));
    const int tps = (ntiles + strips - 1) / strips;
    printf("TIGHTEN=%d n=%d P=%d strips=%d blocks=%d\n", LR_PB_TIGHTEN, n, P, strips, total);
    const bool only_nohit = getenv("PB_ONLY") && !strcmp(getenv("PB_ONLY"), "nohit");      // (counter runs: only the candidate-free walk on real accumulators is repeated)
    for (int need : {2, 1})
    for (int sstride : {2, 4, 8, 16, 32, 64}) {
        if (only_nohit) break;
        PB_THR(thr, nrm, need, sstride);
        static char *s_ysh, *s_arr; static size_t s_stride; static int s_P, s_n; s_ysh = (char *)ysh; s_arr = (char *)arr; s_stride = stride; s_P = P; s_n = n;
        static int s_pre_calls; s_pre_calls = 0;
        // (PB_NOPRE=1: only the first reset happens, so every later run starts from the pooled FINAL thresholds of the run before it -- what a
        // perfect start of the strips would buy)
        g_pre = [] { if (getenv("PB_NOPRE") && s_pre_calls++ > 0) return; for (int p = 0; p < s_P; ++p) { hipMemsetD32Async((hipDeviceptr_t)(s_ysh + p * s_stride), 0xff800000u, s_n, 0); hipMemsetD32Async((hipDeviceptr_t)(s_arr + p * s_stride), 0, s_n / 256 + 2, 0); } };
        auto run = [&] { PB_LAUNCH(grid, dim3(256), 0, 0, lr_pb_tail{ nullptr }, (unsigned long long *)nullptr, H, n, (const int32_t*)nullptr, (const int32_t*)nullptr, H, nrm, n, tps, (const float*)nullptr, cnt, cand,
                                            (const int32_t*)nullptr, (const float*)nullptr, (const uint32_t*)nullptr, (const int32_t*)nullptr, (const uint32_t*)nullptr, (float*)nullptr, 0, PB_YS thr, lr_pb_grid{ row_blocks, strips, total, 0, 0 }, z); };
        float msp = timeit([&] { run(); });
        g_pre(); run();
        const int nseg = row_blocks * 4 * (strips + 1);
        std::vector<int32_t> c1(nseg); hipMemcpy(c1.data(), cnt, (size_t)nseg * 4, hipMemcpyDeviceToHost);
        double tot = 0; int over = 0; for (int i = 0; i < nseg; ++i) if (i % (strips + 1) != strips) { if (c1[i] < 0) over++; else tot += c1[i]; }
        printf("filter pass need=%d sample stride %2d: %8.3f ms  = %6.1f us/pair   list entries/row %.2f  overflowed segments %d\n", need, sstride, msp, msp * 1e3 / P, tot / n, over);
        print_clk(total);
#ifdef LR_PB_PROBE
        {
            unsigned long long z8[16] = {0}, st[16];
            if (g_pre) g_pre();
            hipMemcpyToSymbol(HIP_SYMBOL(lr_pb_stat), z8, sizeof z8); run(); hipDeviceSynchronize();
            hipMemcpyFromSymbol(st, HIP_SYMBOL(lr_pb_stat), sizeof st);
            printf("    per wave: tests %.0f  slow-path visits %.1f (one per %.1f tests)  hits %.1f (%.2f per row)  derive rounds %.1f  16-entry groups %.1f\n",
                   (double)st[1] / st[0], (double)st[2] / st[0], (double)st[1] / (double)(st[2] ? st[2] : 1), (double)st[3] / st[0], (double)st[3] / st[0] / 64.0, (double)st[4] / st[0], (double)st[5] / st[0]);
            printf("    per wave, us: lifetime %.1f  before the walk (thresholds / sample phase) %.1f  derive rounds %.1f  list flushes %.1f\n",
                   (double)st[9] / st[0] / 100.0, (double)st[8] / st[0] / 100.0, (double)st[6] / st[0] / 100.0, (double)st[7] / st[0] / 100.0);
            printf("    per wave, us: of the derive rounds: waiting for the gathers %.1f  threshold update + exchange %.1f\n", (double)st[10] / st[0] / 100.0, (double)st[11] / st[0] / 100.0);
        }
#endif
    }
    {
        // the walk alone on REAL thresholds: the final thresholds of a tightening run (stride 16) given as tau -> no sample phase, no
        // tightening, only the true candidates hit (accumulators hold real values, unlike the -1e30 run below)
        float *yf = (float *)(base + oY);
        PB_THR(thr, nrm, 2, 16);
        PB_LAUNCH(grid, dim3(256), 0, 0, lr_pb_tail{ nullptr }, (unsigned long long *)nullptr, H, n, (const int32_t*)nullptr, (const int32_t*)nullptr, H, nrm, n, tps, (const float*)nullptr, cnt, cand,
                           (const int32_t*)nullptr, (const float*)nullptr, (const uint32_t*)nullptr, (const int32_t*)nullptr, (const uint32_t*)nullptr, yf, n, PB_YS thr, lr_pb_grid{ row_blocks, strips, total, 0, 0 }, z);
        hipDeviceSynchronize();
        std::vector<float> y((size_t)n * strips), t(n);
        hipMemcpy(y.data(), yf, (size_t)n * strips * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) { float m = y[i]; for (int sidx = 1; sidx < strips; ++sidx) m = std::min(m, y[(size_t)sidx * n + i]); t[i] = 2.0f * m; }
        for (int p = 0; p < P; ++p) hipMemcpy((char *)tau + p * stride, t.data(), (size_t)n * 4, hipMemcpyHostToDevice);
        auto run = [&] { PB_LAUNCH(grid, dim3(256), 0, 0, lr_pb_tail{ nullptr }, (unsigned long long *)nullptr, H, n, (const int32_t*)nullptr, (const int32_t*)nullptr, H, nrm, n, tps, (const float*)tau, cnt, cand,
                                            (const int32_t*)nullptr, (const float*)nullptr, (const uint32_t*)nullptr, (const int32_t*)nullptr, (const uint32_t*)nullptr, (float*)nullptr, 0, PB_YS thr0, lr_pb_grid{ row_blocks, strips, total, 0, 0 }, z); };
        float msp = only_nohit ? 0.0f : timeit([&] { run(); });
        const int nseg = row_blocks * 4 * (strips + 1);
        std::vector<int32_t> c1(nseg); hipMemcpy(c1.data(), cnt, (size_t)nseg * 4, hipMemcpyDeviceToHost);
        double tot = 0; for (int i = 0; i < nseg; ++i) if (i % (strips + 1) != strips && c1[i] > 0) tot += c1[i];
        printf("walk only, final thresholds given: %8.3f ms  = %6.1f us/pair   list entries/row %.2f\n", msp, msp * 1e3 / P, tot / n);
        print_clk(total);
        // the same thresholds moved out of reach (tau - 0.5: accumulators of the same magnitude, nothing passes)
        for (int i = 0; i < n; ++i) t[i] -= 0.5f;
        for (int p = 0; p < P; ++p) hipMemcpy((char *)tau + p * stride, t.data(), (size_t)n * 4, hipMemcpyHostToDevice);
        msp = timeit([&] { run(); });
        printf("walk only, final thresholds - 0.5 (no hits, real accumulators): %8.3f ms  = %6.1f us/pair\n", msp, msp * 1e3 / P);
        print_clk(total);
    }
    if (!only_nohit) {
        std::vector<float> t(n, -1e30f);
        for (int p = 0; p < P; ++p) hipMemcpy((char *)tau + p * stride, t.data(), (size_t)n * 4, hipMemcpyHostToDevice);
        float msp = timeit([&] { PB_LAUNCH(grid, dim3(256), 0, 0, lr_pb_tail{ nullptr }, (unsigned long long *)nullptr, H, n, (const int32_t*)nullptr, (const int32_t*)nullptr, H, nrm, n, tps, (const float*)tau, cnt, cand,
                                                    (const int32_t*)nullptr, (const float*)nullptr, (const uint32_t*)nullptr, (const int32_t*)nullptr, (const uint32_t*)nullptr, (float*)nullptr, 0, PB_YS thr0, lr_pb_grid{ row_blocks, strips, total, 0, 0 }, z); });
        printf("walk only, no candidates:       %8.3f ms  = %6.1f us/pair\n", msp, msp * 1e3 / P);
        print_clk(total);
    }
    return 0;
}
