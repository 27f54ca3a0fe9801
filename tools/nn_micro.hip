// Stand-alone timing harness for nn_strip_kernel variants (development tool, not part of the product).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off [-DVARIANT...] tools/nn_micro.hip -o tools/bin/nn_micro_X
#include "../lidarregistration_amd/csrc/lr_nn.hip"
#include <vector>
#include <random>
void lr_set_error(const char *, ...) {}
int main(int argc, char **argv)
{
    int n = argc > 1 ? atoi(argv[1]) : 30000;
    int strips = argc > 2 ? atoi(argv[2]) : 2;
    int reps = argc > 3 ? atoi(argv[3]) : 20;
    std::vector<float> h((size_t)n * 32);
    std::mt19937 rng(1); std::normal_distribution<float> nd;
    for (size_t r = 0; r < (size_t)n; ++r) { double s = 0; for (int k = 0; k < 32; ++k) { h[r*32+k] = nd(rng); s += h[r*32+k]*h[r*32+k]; } for (int k = 0; k < 32; ++k) h[r*32+k] /= (float)sqrt(s); }
    float *F, *Fp, *nrm, *pb1, *pb2, *pb3; int32_t *pi1, *pi2;
    hipMalloc(&F, (size_t)n*128); hipMalloc(&Fp, (size_t)n*128); hipMalloc(&nrm, n*4);
    hipMalloc(&pb1, (size_t)n*4*8); hipMalloc(&pb2, (size_t)n*4*8); hipMalloc(&pb3, (size_t)n*4*8); hipMalloc(&pi1, (size_t)n*4*8); hipMalloc(&pi2, (size_t)n*4*8);
    hipMemcpy(F, h.data(), (size_t)n*128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(nn_prep_kernel, dim3((n+255)/256), dim3(256), 0, 0, F, n, Fp, nrm);
    int cols = ((n + strips - 1) / strips + 31) / 32 * 32;
    dim3 grid((n + 127) / 128, strips);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(nn_strip_kernel, grid, dim3(256), 0, 0, Fp, nrm, n, Fp, nrm, n, cols, n, pb1, pb2, pb3, pi1, pi2);
    hipDeviceSynchronize();
    float best = 1e9, tot = 0;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(nn_strip_kernel, grid, dim3(256), 0, 0, Fp, nrm, n, Fp, nrm, n, cols, n, pb1, pb2, pb3, pi1, pi2);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; tot += ms;
    }
    double fl = 2.0 * 32 * n * (double)n;
    printf("%s n=%d strips=%d  avg %.3f ms  min %.3f ms  -> %.1f TFLOP/s (min)  %.1f%% of 157.3\n", argv[0], n, strips, tot/reps, best, fl/best/1e9, fl/best/1e9/157.3*100);
    return 0;
}
