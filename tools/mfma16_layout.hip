// Layout check of v_mfma_f32_16x16x32_f16 (development tool): prints how many of the 64 x 4 result registers match the layout
// nn16_passb_kernel assumes -- lane l supplies row/column l % 16, receives column l % 16, rows 4 (l / 16) + 0..3.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma16_layout.hip -o tools/bin/mfma16_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *out)
{
    const int l = threadIdx.x, c = l & 15, kb = l >> 4;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)((kb == 1 && e == 3) ? c + 1 : 0); b[e] = (_Float16)((kb == 1 && e == 3) ? 32 * (c + 1) : 0); }
    f32x4 y = { 1000.f * (4 * kb + 0), 1000.f * (4 * kb + 1), 1000.f * (4 * kb + 2), 1000.f * (4 * kb + 3) };
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, y, 0, 0, 0);
    for (int g = 0; g < 4; ++g) out[l * 4 + g] = d[g];
}
int main()
{
    float *d; hipMalloc(&d, 1024); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int l = 0; l < 64; ++l) for (int g = 0; g < 4; ++g) {
        const int row = 4 * (l >> 4) + g, col = l & 15;
        const float want = 1000.f * row + (row + 1) * 32.f * (col + 1);
        if (h[l * 4 + g] == want) ++ok; else if (ok < 8) printf("lane %d reg %d: got %g want %g\n", l, g, h[l * 4 + g], want);
    }
    printf("%d of 256 registers as assumed\n", ok);
    return ok == 256 ? 0 : 1;
}
