// Can the half-transformed Y of the MSD transform live in the Infinity Cache?  Per batch: kernel A reads `in_bytes`
// of fresh input (HBM) and writes `y_bytes` of Y; kernel B reads Y back.  Y is either one buffer reused by every
// batch (candidate: stays in the 256 MiB cache) or a fresh slice of a 24 GiB buffer per batch (today's pattern).
// hipcc -O2 --offload-arch=gfx950 scripts/mall_bench.hip -o scripts/mall_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k_a(const float4 *__restrict__ in, size_t n_in, float4 *__restrict__ y, size_t n_y)
{
    const size_t stride = size_t(gridDim.x) * blockDim.x, t = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = t; i < n_in; i += stride) {
        const float4 v = in[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    for (size_t i = t; i < n_y; i += stride)
        y[i] = acc;
}
__global__ __launch_bounds__(256) void k_b(const float4 *__restrict__ y, size_t n_y, float *out)
{
    const size_t stride = size_t(gridDim.x) * blockDim.x;
    float s = 0.f;
    for (size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x; i < n_y; i += stride) {
        const float4 v = y[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 123.456f) out[0] = s;
}
int main()
{
    const size_t total_y = size_t(24) << 30, total_in = size_t(12) << 30;
    float4 *in, *y; float *o;
    if (hipMalloc(&in, total_in) != hipSuccess || hipMalloc(&y, total_y) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&o, 4);
    hipMemset(in, 1, total_in); hipMemset(y, 0, total_y);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t y_mb : {32, 64, 96, 128, 192, 384, 1024}) {
        const size_t y_bytes = y_mb << 20, in_bytes = y_bytes / 2;
        const int batches = int(total_y / y_bytes);
        for (int reuse = 1; reuse >= 0; --reuse) {
            for (int blocks : {1024, 4096}) {
                float best = 1e30f;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    for (int b = 0; b < batches; ++b) {
                        float4 *yb = reuse ? y : y + size_t(b) * (y_bytes / 16);
                        k_a<<<blocks, 256>>>(in + size_t(b) * (in_bytes / 16), in_bytes / 16, yb, y_bytes / 16);
                        k_b<<<blocks, 256>>>(yb, y_bytes / 16, o);
                    }
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                printf("Y batch %4zu MiB x %4d batches, %s, %4d blocks: %.2f ms for 12 GiB in + 24 GiB Y written + read  (%.2f TB/s of those bytes)\n",
                       y_mb, batches, reuse ? "one reused Y buffer" : "fresh Y slice      ", blocks, best,
                       double(total_in + 2 * total_y) / best * 1e-9);
            }
        }
    }
    return 0;
}
