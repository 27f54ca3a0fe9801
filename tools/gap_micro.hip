// What fits in the shadow of a v_mfma_f32_32x32x16_f16 on gfx950?  (development tool)
// One wave per SIMD (256 blocks x 256 threads) or several (argv[1] = blocks per CU); each wave runs ITER iterations of
// four [MFMA ; filler] gaps and reports s_memtime cycles per gap.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/gap_micro.hip -o tools/bin/gap_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define ITER 2000

// VARIANT: 0 MFMA only; 1 + 4 v_max3 (independent regs); 2 + v_cmp + 4 v_max (tree) ; 3 = 2 + s_cmp + s_cbranch (never taken)
//          4 = 3 but fillers read the accumulators written two MFMAs earlier; 5 = 4 with C != D (fresh C each tile)
template <int V>
__global__ void __launch_bounds__(256) gap_kernel(const float *in, float *out, unsigned long long *cyc, int never)
{
    const int lane = threadIdx.x & 63;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)in[lane + i]; b[i] = (_Float16)in[64 + lane + i]; }
    f32x16 accA, accB, y;
    for (int i = 0; i < 16; ++i) { accA[i] = in[i]; accB[i] = in[16 + i]; y[i] = in[32 + i]; }
    float f0 = in[lane], f1 = in[lane + 1], f2 = in[lane + 2], f3 = in[lane + 3], f4 = in[lane + 4], f5 = in[lane + 5], f6 = in[lane + 6],
          f7 = in[lane + 7];
    float m = 0.f, x = in[200] + 1e30f, keep = 0.f;
    int parked = 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    f32x16 accC = y, accD = y;
    auto filler = [&](const f32x16 &src, int o) {
        if (V == 1) {
            float t1, t2, t3, t4;
            asm volatile("v_max3_f32 %0, %4, %5, %6\n\tv_max3_f32 %1, %7, %8, %9\n\tv_max3_f32 %2, %4, %7, %10\n\tv_max3_f32 %3, %5, %8, %11"
                         : "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
            keep += t1 + t2 + t3 + t4;
        }
        if (V >= 2) {
            unsigned long long hit;
            float mm, t1, t2;
            if (V >= 4)
                asm volatile("v_cmp_ge_f32_e64 %0, %4, %5\n\tv_max3_f32 %1, %6, %7, %8\n\tv_max3_f32 %2, %9, %10, %11\n\tv_max_f32 %3, %12, %13\n\tv_max3_f32 %1, %1, %2, %3"
                             : "=&s"(hit), "=&v"(mm), "=&v"(t1), "=&v"(t2)
                             : "v"(m), "v"(x), "v"(src[o]), "v"(src[o + 1]), "v"(src[o + 2]), "v"(src[o + 3]), "v"(src[o + 4]), "v"(src[o + 5]), "v"(src[o + 6]), "v"(src[o + 7]));
            else
                asm volatile("v_cmp_ge_f32_e64 %0, %4, %5\n\tv_max3_f32 %1, %6, %7, %8\n\tv_max3_f32 %2, %9, %10, %11\n\tv_max_f32 %3, %12, %13\n\tv_max3_f32 %1, %1, %2, %3"
                             : "=&s"(hit), "=&v"(mm), "=&v"(t1), "=&v"(t2)
                             : "v"(m), "v"(x), "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f6), "v"(f7));
            if (V >= 3) {
                if (__builtin_expect(hit != 0ull, 0)) { parked += never; out[parked] = mm; }
            } else keep += (float)(unsigned)hit;
            m = mm;
        }
    };
    // one tile: four MFMAs into (n0, n1), fillers over the previous tile (c0, c1) -- the layout of nn16_passb_kernel
    auto tile = [&](f32x16 &n0, f32x16 &n1, const f32x16 &c0, const f32x16 &c1) {
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, V == 5 ? y : n0, 0, 0, 0);
        filler(c0, 0);
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, V == 5 ? y : n1, 0, 0, 0);
        filler(c0, 8);
        n0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, n0, 0, 0, 0);
        filler(c1, 0);
        n1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, n1, 0, 0, 0);
        filler(c1, 8);
    };
    for (int it = 0; it < ITER / 2; ++it) {
        tile(accA, accB, accC, accD);
        tile(accC, accD, accA, accB);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = keep + m;
    for (int i = 0; i < 16; ++i) s += accA[i] + accB[i] + accC[i] + accD[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + parked;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V> void run(int blocks, const float *in, float *out, unsigned long long *cyc)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(gap_kernel<V>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(gap_kernel<V>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    // s_memtime counts at 100 MHz on gfx9 (constant clock); report both
    printf("variant %d  blocks %d: %.3f ms  -> %.1f ns per gap per wave;  counter ticks/gap %.3f\n", V, blocks, ms,
           ms * 1e6 / (ITER * 4.0), s / h.size() / (ITER * 4.0));
}

int main(int argc, char **argv)
{
    int per_cu = argc > 1 ? atoi(argv[1]) : 1;
    int blocks = 256 * per_cu;
    float *in, *out; unsigned long long *cyc;
    hipMalloc(&in, 4096); hipMalloc(&out, (size_t)blocks * 256 * 4 + 4096); hipMalloc(&cyc, blocks * 4 * 8);
    std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (float)(i % 7) * 0.125f;
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    run<0>(blocks, in, out, cyc); run<1>(blocks, in, out, cyc); run<2>(blocks, in, out, cyc);
    run<3>(blocks, in, out, cyc); run<4>(blocks, in, out, cyc); run<5>(blocks, in, out, cyc);
    return 0;
}
