// VALU issue-rate microbenchmark: scalar vs packed f32 ops, 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void kern(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    float b = 1.0001f, c = 0.5f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    f2 pb = {b, b}, pc = {c, c};
    double d0 = a0, d1 = 1., d2 = 2., d3 = 3., d4 = 4., d5 = 5., d6 = 6., d7 = 7., db = 1.0001, dc = 0.5;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 q0 = {0, 0, 0, 0}, q1 = q0, q2 = q0, q3 = q0;
    __shared__ float lds[16384];
    if (MODE >= 6) {
        for (int i = threadIdx.x; i < 16384; i += blockDim.x)
            lds[i] = i;
        __syncthreads();
    }
    const unsigned lds_addr = MODE == 7 ? (threadIdx.x & 63) * 16 : 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {   // 8 independent v_fma_f32 x 16
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
        } else if (MODE == 1) {   // 4 independent v_pk_fma_f32 x 16 (same flops as 8 scalar)
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
        } else if (MODE == 2) {   // 8 v_sqrt_f32
            REP16(asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n"
                               "v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 3) {   // 4 v_pk_add_f32
            REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));)
        } else if (MODE == 4) {   // 8 v_cmp_gt_f32 into vcc
            REP16(asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %2\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %3, %0\n"
                               "v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %2\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %3, %0\n"
                               :: "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");)
        } else if (MODE == 5) {   // 8 independent v_fma_f64
            REP16(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                               "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                               : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db), "v"(dc));)
        } else if (MODE == 6 || MODE == 7) {   // 8 ds_read_b128: one address for the wave / 16 B per lane
            REP16(asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n"
                               "ds_read_b128 %0, %4 offset:4096\n ds_read_b128 %1, %4 offset:5120\n ds_read_b128 %2, %4 offset:6144\n ds_read_b128 %3, %4 offset:7168\n"
                               "s_waitcnt lgkmcnt(0)\n"
                               : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(lds_addr) : "memory");)
        }
    }
    a0 += (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + q0.x + q1.y + q2.z + q3.w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE> void run(const char *name, int n_instr_per_iter)
{
    float *d; hipMalloc(&d, 256 * 64 * 64 * 4 * 8);
    for (int waves = 1; waves <= 8; waves *= 2) {
        // one block of (waves*4) waves per CU -> `waves` per SIMD
        dim3 grid(256), block(64 * 4 * waves);
        const int iters = 2000;
        kern<MODE><<<grid, block>>>(d, 10);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        kern<MODE><<<grid, block>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr_per_simd = double(iters) * 16 * n_instr_per_iter * waves;
        double cycles = ms * 1e-3 * 2.4e9;
        printf("%-14s waves/SIMD=%d  %.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", name, waves, cycles / instr_per_simd);
    }
    hipFree(d);
}
int main()
{
    run<0>("v_fma_f32", 8);
    run<1>("v_pk_fma_f32", 4);
    run<2>("v_sqrt_f32", 8);
    run<3>("v_pk_add_f32", 4);
    run<4>("v_cmp_gt_f32", 8);
    run<5>("v_fma_f64", 8);
    run<6>("ds_read_b128 bc", 8);
    run<7>("ds_read_b128 ln", 8);
    return 0;
}
