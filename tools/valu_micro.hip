// VALU issue rate on gfx950: wall cycles per wave64 instruction per SIMD for a list of opcodes at 1 / 2 / 4 waves per SIMD
// (256 blocks = one per CU, 256 * wps threads, 8 independent destination registers per opcode, 2000 x 64 instructions per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define BODY3(OPSTR) OPSTR " %0, %0, %8, %9\n" OPSTR " %1, %1, %8, %9\n" OPSTR " %2, %2, %8, %9\n" OPSTR " %3, %3, %8, %9\n" \
                     OPSTR " %4, %4, %8, %9\n" OPSTR " %5, %5, %8, %9\n" OPSTR " %6, %6, %8, %9\n" OPSTR " %7, %7, %8, %9\n"
#define BODY2(OPSTR) OPSTR " %0, %0, %8\n" OPSTR " %1, %1, %8\n" OPSTR " %2, %2, %8\n" OPSTR " %3, %3, %8\n" \
                     OPSTR " %4, %4, %8\n" OPSTR " %5, %5, %8\n" OPSTR " %6, %6, %8\n" OPSTR " %7, %7, %8\n"
#define BODY1(OPSTR) OPSTR " %0, %0\n" OPSTR " %1, %1\n" OPSTR " %2, %2\n" OPSTR " %3, %3\n" \
                     OPSTR " %4, %4\n" OPSTR " %5, %5\n" OPSTR " %6, %6\n" OPSTR " %7, %7\n"
#define BODYC(OPSTR) OPSTR " vcc, %0, %8\n" OPSTR " vcc, %1, %8\n" OPSTR " vcc, %2, %8\n" OPSTR " vcc, %3, %8\n" \
                     OPSTR " vcc, %4, %8\n" OPSTR " vcc, %5, %8\n" OPSTR " vcc, %6, %8\n" OPSTR " vcc, %7, %8\n"
#define BODYM(OPSTR) OPSTR " %0, %0, %8, vcc\n" OPSTR " %1, %1, %8, vcc\n" OPSTR " %2, %2, %8, vcc\n" OPSTR " %3, %3, %8, vcc\n" \
                     OPSTR " %4, %4, %8, vcc\n" OPSTR " %5, %5, %8, vcc\n" OPSTR " %6, %6, %8, vcc\n" OPSTR " %7, %7, %8, vcc\n"

#define KERNEL(NAME, BODY, REGT, CONS)                                                                                      \
    __global__ void __launch_bounds__(1024) NAME(float *out, int iters)                                                    \
    {                                                                                                                       \
        REGT a0 = (REGT)(threadIdx.x), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        const REGT c = (REGT)out[0] + 1, d = (REGT)out[1] + 2;                                                              \
        for (int i = 0; i < iters; ++i) {                                                                                   \
            _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                                   \
                asm volatile(BODY : "+" CONS(a0), "+" CONS(a1), "+" CONS(a2), "+" CONS(a3), "+" CONS(a4), "+" CONS(a5), "+" CONS(a6), "+" CONS(a7) \
                             : CONS(c), CONS(d) : "vcc");                                                                   \
        }                                                                                                                   \
        REGT s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                                     \
        if (*(float *)&s == 12345.678f) out[2] = *(float *)&s;                                                              \
    }
#define V(x) "v"(x)
KERNEL(k_fma, BODY3("v_fma_f32"), float, V)
KERNEL(k_add, BODY2("v_add_f32"), float, V)
KERNEL(k_sub, BODY2("v_sub_f32"), float, V)
KERNEL(k_mul, BODY2("v_mul_f32"), float, V)
KERNEL(k_max, BODY2("v_max_f32"), float, V)
KERNEL(k_min, BODY2("v_min_f32"), float, V)
KERNEL(k_max3, BODY3("v_max3_f32"), float, V)
KERNEL(k_med3, BODY3("v_med3_f32"), float, V)
KERNEL(k_maximum3, BODY3("v_maximum3_f32"), float, V)
KERNEL(k_maxi, BODY2("v_max_i32"), int, V)
KERNEL(k_maxu, BODY2("v_max_u32"), int, V)
KERNEL(k_max3i, BODY3("v_max3_i32"), int, V)
KERNEL(k_max3u, BODY3("v_max3_u32"), int, V)
KERNEL(k_addu, BODY2("v_add_u32"), int, V)
KERNEL(k_add3u, BODY3("v_add3_u32"), int, V)
KERNEL(k_or, BODY2("v_or_b32"), int, V)
KERNEL(k_or3, BODY3("v_or3_b32"), int, V)
KERNEL(k_andor, BODY3("v_and_or_b32"), int, V)
KERNEL(k_alignbit, BODY3("v_alignbit_b32"), int, V)
KERNEL(k_lshladd, BODY3("v_lshl_add_u32"), int, V)
KERNEL(k_bfe, BODY3("v_bfe_u32"), int, V)
KERNEL(k_perm, BODY3("v_perm_b32"), int, V)
KERNEL(k_mov, BODY1("v_mov_b32"), int, V)
KERNEL(k_cvtu, BODY1("v_cvt_u32_f32"), int, V)
KERNEL(k_cvtpk, BODY2("v_cvt_pk_f16_f32"), int, V)
KERNEL(k_cmpf, BODYC("v_cmp_gt_f32"), float, V)
KERNEL(k_cmpu, BODYC("v_cmp_gt_u32"), int, V)
KERNEL(k_cmpi, BODYC("v_cmp_gt_i32"), int, V)
KERNEL(k_cnd, BODYM("v_cndmask_b32"), int, V)
#define BODYA "v_addc_co_u32 %0, vcc, %0, %8, vcc\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n"
KERNEL(k_addc, BODYA, int, V)
KERNEL(k_pkmaxh, BODY2("v_pk_max_f16"), int, V)
KERNEL(k_pkaddh, BODY2("v_pk_add_f16"), int, V)
KERNEL(k_pkfmah, BODY3("v_pk_fma_f16"), int, V)
KERNEL(k_dot2, BODY3("v_dot2_f32_f16"), int, V)
KERNEL(k_madu24, BODY3("v_mad_u32_u24"), int, V)
KERNEL(k_sad, BODY3("v_sad_u32"), int, V)
KERNEL(k_bcnt, BODY2("v_bcnt_u32_b32"), int, V)
KERNEL(k_mbcnt, BODY2("v_mbcnt_lo_u32_b32"), int, V)
KERNEL(k_lshr, BODY2("v_lshrrev_b32"), int, V)
KERNEL(k_ashr, BODY2("v_ashrrev_i32"), int, V)
KERNEL(k_xor, BODY2("v_xor_b32"), int, V)
#define BODYB "v_bitop3_b32 %0, %0, %8, %9 bitop3:0x80\n v_bitop3_b32 %1, %1, %8, %9 bitop3:0x80\n v_bitop3_b32 %2, %2, %8, %9 bitop3:0x80\n v_bitop3_b32 %3, %3, %8, %9 bitop3:0x80\n v_bitop3_b32 %4, %4, %8, %9 bitop3:0x80\n v_bitop3_b32 %5, %5, %8, %9 bitop3:0x80\n v_bitop3_b32 %6, %6, %8, %9 bitop3:0x80\n v_bitop3_b32 %7, %7, %8, %9 bitop3:0x80\n"
KERNEL(k_bitop3, BODYB, int, V)
KERNEL(k_subu, BODY2("v_sub_u32"), int, V)
KERNEL(k_minu, BODY2("v_min_u32"), int, V)
KERNEL(k_fmac, BODY2("v_fmac_f32"), float, V)
KERNEL(k_mulu24, BODY2("v_mul_u32_u24"), int, V)
KERNEL(k_mullo, BODY2("v_mul_lo_u32"), int, V)

typedef void (*kfn)(float *, int);
static void run(const char *name, kfn f, float *out)
{
    const int iters = 2000;
    printf("%-18s", name);
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int threads = 256 * wps;
        hipLaunchKernelGGL(f, dim3(256), dim3(threads), 0, 0, out, iters);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(f, dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double n_inst = 4.0 * iters * 64 * wps;
        printf("  %dw: %5.2f", wps, ms * 1e6 / n_inst * 2.4);
    }
    printf("   (cycles @2.4 GHz per wave64 instruction per SIMD)\n");
}
#define R(k) run(#k, k, out)
int main()
{
    float *out;
    hipMalloc(&out, 64); hipMemset(out, 0, 64);
    R(k_fma); R(k_fmac); R(k_add); R(k_sub); R(k_mul); R(k_max); R(k_min); R(k_max3); R(k_med3); R(k_maximum3); R(k_maxi); R(k_maxu); R(k_max3i); R(k_max3u); R(k_minu);
    R(k_addu); R(k_subu); R(k_add3u); R(k_or); R(k_xor); R(k_or3); R(k_andor); R(k_bitop3); R(k_alignbit); R(k_lshladd); R(k_lshr); R(k_ashr); R(k_bfe); R(k_perm); R(k_mov);
    R(k_cvtu); R(k_cvtpk); R(k_cmpf); R(k_cmpu); R(k_cmpi); R(k_cnd); R(k_addc); R(k_pkmaxh); R(k_pkaddh); R(k_pkfmah); R(k_dot2); R(k_madu24); R(k_sad);
    R(k_bcnt); R(k_mbcnt); R(k_mulu24); R(k_mullo);
    return 0;
}
