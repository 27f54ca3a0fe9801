// Do MFMA and VALU instructions of different waves of one SIMD overlap? (development tool)  Each wave loops over NM independent
// v_mfma_f32_16x16x32_f16 and NV VALU ops (v_max3_f32, or v_and_b32 of the fast class) on registers of their own; the table
// gives nanoseconds of SIMD time per loop iteration for MFMA only, VALU only and both, at 1..4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/coissue_micro.hip -o tools/bin/coissue_micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NM, int NV, int KIND>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *clk, int iters)
{
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (threadIdx.x + e)); b[e] = (_Float16)(0.02f * (threadIdx.x - e)); }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{ 0.f, 0.f, 0.f, 0.f };
    float v[8]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.5f + i; u[i] = threadIdx.x * 77u + i; }
    const float c0 = out[0], c1 = out[1], ninf = -__builtin_huge_valf() + out[0];
    unsigned long long cm[4] = { 0, 0, 0, 0 };
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (s < NM) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[s]) : "v"(a), "v"(b));
#pragma unroll
            for (int j = s * NV / 8; j < (s + 1) * NV / 8; ++j) {       // consecutive ops on different registers: no dependent issue
                if (KIND == 0 || KIND == 2 || KIND == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[j % 8]) : "v"(c0), "v"(c1));
                else asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[j % 8]) : "v"(c0));
            }
            // KIND 2: a comparison + a never-taken scalar branch on its result after every second MFMA (the filter walk's test groups);
            // KIND 3: the four comparisons into scalar registers, one branch per iteration
            if (KIND == 2 && (s & 1)) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_cbranch_vccnz 1f\n1:" :: "v"(v[s]), "v"(ninf) : "vcc");
            if (KIND == 3 && (s & 1)) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(cm[s >> 1]) : "v"(v[s]), "v"(ninf));
            if (KIND == 3 && s == 7) { const unsigned long long any = cm[0] | cm[1] | cm[2] | cm[3]; if (any) out[5] = 1.0f; }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0; for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i] + (float)u[i];
    out[2 + blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int NM, int NV, int KIND> double run(int waves_per_simd, float *out, unsigned long long *clk)
{
    const int blocks = 256 * waves_per_simd, iters = 4000;      // a block = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, KIND>), dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<NM, NV, KIND>), dim3(blocks), dim3(256), 0, 0, out, clk, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)ms * 1e6 / ((double)iters * waves_per_simd);        // ns of SIMD time per loop iteration of one wave
}
int main()
{
    float *out; unsigned long long *clk; hipMalloc(&out, 4 * (2 + 1024 * 256)); hipMalloc(&clk, 8 * 1024); hipMemset(out, 0, 8);
    printf("ns of SIMD time per loop iteration of one wave (kernel wall time / iterations / waves per SIMD; all waves run the same loop)\n");
    printf("%-34s %8s %8s %8s %8s\n", "loop body", "1 wave", "2 waves", "3 waves", "4 waves");
#define ROW(NM, NV, KIND, name) { printf("%-34s", name); for (int w = 1; w <= 4; ++w) printf(" %8.1f", run<NM, NV, KIND>(w, out, clk)); printf("\n"); }
    ROW(8, 0, 0, "8 MFMA 16x16x32");
    ROW(0, 20, 0, "20 v_max3_f32");
    ROW(8, 20, 0, "8 MFMA + 20 v_max3_f32");
    ROW(0, 20, 1, "20 v_and_b32");
    ROW(8, 20, 1, "8 MFMA + 20 v_and_b32");
    ROW(8, 8, 0, "8 MFMA + 8 v_max3_f32");
    ROW(8, 16, 2, "8 MFMA + 16 max3 + 4 (cmp, branch)");
    ROW(8, 16, 3, "8 MFMA + 16 max3 + 4 cmp, 1 branch");
    ROW(0, 16, 2, "16 max3 + 4 (cmp, branch)");
    ROW(8, 40, 1, "8 MFMA + 40 v_and_b32");
    return 0;
}
