// Issue-rate and clock microbenchmark for the RDF hot loop's instruction mix (round 2).
//   * real engine clock under each load: s_memtime (clock64) against s_memrealtime (wall_clock64, 100 MHz)
//     inside the kernel, printed next to the HIP-event time; run under
//     `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` to get the counter-based clock as well;
//   * SALU issue rate (independent s_add_u32 / s_and_b32 / s_lshr_b32), alone and beside VALU;
//   * the hot step's mix: 10 plain VALU + v_sqrt_f32 + 2 v_cmp + 10 SALU per 13 VALU.
// hipcc -O2 --offload-arch=gfx950 scripts/issue_bench.hip -o /tmp/issue_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void kern(float *out, long long *clk, int iters)
{
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    float b = 1.0001f, c = 0.5f;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    v2f pb = {b, b}, pc = {c, c};
    unsigned s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;
    const long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 8 independent v_fma_f32
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (MODE == 1) {   // 8 independent SALU
            REP16(asm volatile("s_add_u32 %0, %0, 3\n s_and_b32 %1, %1, 0xffff\n s_lshr_b32 %2, %2, 1\n s_add_u32 %3, %3, 5\n"
                               "s_add_u32 %0, %0, 7\n s_xor_b32 %1, %1, 0x55\n s_lshl_b32 %2, %2, 1\n s_sub_u32 %3, %3, 2\n"
                               : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");)
        } else if (MODE == 2) {   // 8 VALU + 8 SALU interleaved
            REP16(asm volatile("v_fma_f32 %0, %0, %12, %13\n s_add_u32 %8, %8, 3\n v_fma_f32 %1, %1, %12, %13\n s_and_b32 %9, %9, 0xffff\n"
                               "v_fma_f32 %2, %2, %12, %13\n s_lshr_b32 %10, %10, 1\n v_fma_f32 %3, %3, %12, %13\n s_add_u32 %11, %11, 5\n"
                               "v_fma_f32 %4, %4, %12, %13\n s_add_u32 %8, %8, 7\n v_fma_f32 %5, %5, %12, %13\n s_xor_b32 %9, %9, 0x55\n"
                               "v_fma_f32 %6, %6, %12, %13\n s_lshl_b32 %10, %10, 1\n v_fma_f32 %7, %7, %12, %13\n s_sub_u32 %11, %11, 2\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                                 "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                               : "v"(b), "v"(c) : "scc");)
        } else if (MODE == 3) {   // the hot step: 10 plain VALU + sqrt + 2 cmp (13 VALU), no SALU
            REP16(asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_mul_f32 %3, %0, %0\n"
                               "v_fma_f32 %3, %1, %1, %3\n v_fma_f32 %3, %2, %2, %3\n v_sqrt_f32 %4, %3\n v_fma_f32 %4, %4, %8, %9\n"
                               "v_fract_f32 %5, %4\n v_cmp_gt_f32 vcc, %3, %9\n v_cmp_gt_f32 vcc, %5, %9\n v_cvt_i32_f32 %6, %4\n v_lshlrev_b32 %7, 2, %6\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
        } else if (MODE == 4) {   // the same with 10 SALU spread through it
            REP16(asm volatile("v_sub_f32 %0, %0, %12\n s_add_u32 %8, %8, 3\n v_sub_f32 %1, %1, %12\n s_and_b32 %9, %9, 0xffff\n v_sub_f32 %2, %2, %12\n s_lshr_b32 %10, %10, 1\n v_mul_f32 %3, %0, %0\n"
                               "s_add_u32 %11, %11, 5\n v_fma_f32 %3, %1, %1, %3\n s_add_u32 %8, %8, 7\n v_fma_f32 %3, %2, %2, %3\n s_xor_b32 %9, %9, 0x55\n v_sqrt_f32 %4, %3\n s_lshl_b32 %10, %10, 1\n v_fma_f32 %4, %4, %12, %13\n"
                               "s_sub_u32 %11, %11, 2\n v_fract_f32 %5, %4\n s_add_u32 %8, %8, 1\n v_cmp_gt_f32 vcc, %3, %13\n s_add_u32 %9, %9, 1\n v_cmp_gt_f32 vcc, %5, %13\n v_cvt_i32_f32 %6, %4\n v_lshlrev_b32 %7, 2, %6\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                                 "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                               : "v"(b), "v"(c) : "vcc", "scc");)
        } else if (MODE == 5) {   // 8 independent v_pk_fma_f32 (two f32 per lane each)
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                               "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
        } else if (MODE == 6) {   // 4 v_pk_add_f32 + 4 v_pk_mul_f32
            REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %9\n v_pk_add_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %9\n"
                               "v_pk_add_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %9\n v_pk_add_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %9\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb), "v"(pc));)
        } else if (MODE == 7) {   // 4 v_sub_u32 + 4 v_cvt_f32_i32 (fixed-point minimum image)
            REP16(asm volatile("v_sub_u32 %0, %0, %8\n v_cvt_f32_i32 %1, %0\n v_sub_u32 %2, %2, %8\n v_cvt_f32_i32 %3, %2\n"
                               "v_sub_u32 %4, %4, %8\n v_cvt_f32_i32 %5, %4\n v_sub_u32 %6, %6, %8\n v_cvt_f32_i32 %7, %6\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        }
    }
    const long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = w1 - w0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + float(s0 + s1 + s2 + s3) +
        (p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7).x + (p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7).y;
}
template <int MODE> void run(const char *name, int n_valu, int n_salu)
{
    float *d; hipMalloc(&d, 256 * 64 * 64 * 4 * 8);
    long long *clk; hipMalloc(&clk, 16);
    for (int waves = 1; waves <= 8; waves *= 2) {
        if (waves == 8) waves = 6;
        dim3 grid(256), block(64 * 4 * (waves > 4 ? 4 : waves));
        if (waves == 6) { grid = dim3(512); block = dim3(64 * 4 * 3); }   // 2 blocks of 12 waves per CU
        const int iters = 4000;
        kern<MODE><<<grid, block>>>(d, clk, 10);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        kern<MODE><<<grid, block>>>(d, clk, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double mhz_memtime = double(h[0]) / double(h[1]) * 100.0;
        const double steps_per_simd = double(iters) * 16 * waves;
        printf("%-22s waves/SIMD=%d  %.3f ms  per 16-instr group and SIMD: %.1f ns = %.1f cycles at 2.4 GHz  "
               "(VALU %d SALU %d per group; s_memtime/s_memrealtime -> %.0f MHz)\n",
               name, waves, ms, ms * 1e6 / steps_per_simd, ms * 1e-3 * 2.4e9 / steps_per_simd, n_valu, n_salu, mhz_memtime);
        if (waves == 6) break;
    }
    hipFree(d); hipFree(clk);
}
int main()
{
    run<0>("8 v_fma_f32", 8, 0);
    run<1>("8 SALU", 0, 8);
    run<2>("8 v_fma_f32 + 8 SALU", 8, 8);
    run<3>("hot step 13 VALU", 13, 0);
    run<4>("hot step 13 VALU+10 SALU", 13, 10);
    run<5>("8 v_pk_fma_f32", 8, 0);
    run<6>("4 pk_add + 4 pk_mul f32", 8, 0);
    run<7>("4 v_sub_u32+4 cvt_f32_i32", 8, 0);
    return 0;
}
