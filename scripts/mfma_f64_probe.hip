#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(double *out, int ka, int kb)
{
    const int lane = threadIdx.x;
    // assumed: A[i][k] in lane i + 16 k; B[k][j] in lane j + 16 k
    const int i = lane & 15, kk = lane >> 4;
    double a = (kk == ka) ? double(i + 1) : 0.0;
    double b = (kk == kb) ? double((lane & 15) + 1) * 100.0 : 0.0;
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v)
        out[lane * 4 + v] = c[v];
}
__global__ void rate(double *out, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
template <int NV, bool DP>
__global__ void mix(double *out, int iters)
{
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    v4d c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float f[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    double g[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (DP)
                asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(g[k & 7]) : "v"(b), "v"(a));
            else
                asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[k & 7]) : "v"(f[(k + 1) & 7]));
        }
    }
    double s = c0[0] + c1[1] + c2[2] + c3[3];
    for (int k = 0; k < 8; ++k) s += f[k] + g[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NV, bool DP> void run_mix(double *d)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    mix<NV, DP><<<256, 512>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    mix<NV, DP><<<256, 512>>>(d, 20000);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("4 MFMA + %d %s VALU per iteration, 2 waves/SIMD: %.1f cycles per iteration per wave-slot (2.4 GHz)\n", NV,
           DP ? "f64" : "f32", ms * 1e-3 * 2.4e9 / (20000.0 * 2));
}
int main()
{
    {
        double *d; hipMalloc(&d, 256 * 1024 * 8);
        run_mix<0, false>(d); run_mix<16, false>(d); run_mix<32, false>(d); run_mix<64, false>(d);
        run_mix<16, true>(d); run_mix<32, true>(d);
        hipFree(d);
    }
    {
        double *d; hipMalloc(&d, 256 * 1024 * 8);
        for (int waves = 1; waves <= 4; waves *= 2) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            rate<<<256, 256 * waves>>>(d, 10);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            rate<<<256, 256 * waves>>>(d, 20000);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double mfma_per_simd = 20000.0 * 4 * waves;
            printf("mfma f64 16x16x4: %d waves/SIMD: %.1f cycles per MFMA per SIMD at 2.4 GHz, %.1f TFLOP/s\n", waves,
                   ms * 1e-3 * 2.4e9 / mfma_per_simd, mfma_per_simd * 1024 * 2048 / (ms * 1e-3) / 1e12);
        }
        hipFree(d);
    }
    double *d; hipMalloc(&d, 64 * 4 * 8);
    double h[256];
    for (int ka = 0; ka < 4; ka += 3)
        for (int kb = 0; kb < 4; kb += 3) {
            k<<<1, 64>>>(d, ka, kb);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            printf("ka=%d kb=%d\n", ka, kb);
            int ok_layout1 = 1, nonzero = 0;
            for (int lane = 0; lane < 64; ++lane)
                for (int v = 0; v < 4; ++v) {
                    double x = h[lane * 4 + v];
                    if (x != 0) nonzero++;
                    int i = 4 * v + (lane >> 4), j = lane & 15;     // candidate layout
                    double want = (ka == kb) ? (i + 1) * (j + 1) * 100.0 : 0.0;
                    if (x != want) ok_layout1 = 0;
                }
            printf("  nonzero=%d candidate(i=4*v+lane/16, j=lane%%16) %s\n", nonzero, ok_layout1 ? "MATCH" : "no");
            if (ka == kb && !ok_layout1) {
                for (int lane = 0; lane < 64; lane += 7)
                    printf("  lane %d: %g %g %g %g\n", lane, h[lane*4], h[lane*4+1], h[lane*4+2], h[lane*4+3]);
            }
        }
    return 0;
}
