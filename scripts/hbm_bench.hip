// Achievable HBM bandwidth on this part: streaming read (sum), copy (read + write) and write,
// 4 GiB buffers, grid-stride float4 accesses.  hipcc -O2 --offload-arch=gfx950 scripts/hbm_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read(const float4 *__restrict__ a, size_t n, float *out)
{
    float s = 0.f;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x) {
        const float4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f)
        out[0] = s;
}
__global__ void k_copy(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
        b[i] = a[i];
}
__global__ void k_write(float4 *__restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
        b[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
int main()
{
    const size_t bytes = size_t(4) << 30, n = bytes / 16;
    float4 *a, *b;
    float *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 32768}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) k_read<<<blocks, 256>>>(a, n, o);
                else if (mode == 1) k_copy<<<blocks, 256>>>(a, b, n);
                else k_write<<<blocks, 256>>>(b, n);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double moved = (mode == 1 ? 2.0 : 1.0) * bytes;
            printf("%-5s %6d blocks: %.2f TB/s\n", mode == 0 ? "read" : mode == 1 ? "copy" : "write", blocks, moved / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
