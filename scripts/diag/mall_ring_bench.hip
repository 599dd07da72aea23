// Does the 256 MB memory-side cache absorb a write-then-read round trip?  Each block writes a private region and
// reads it back; "ring" reuses the same footprint every iteration, "stream" advances through a 16 GB buffer.
// Build: hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mall_ring scripts/diag/mall_ring_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void roundtrip(double2 *buf, size_t region16, size_t foot16, size_t total16,
                                                 int iters, int ring, int do_write, int do_read, double *sink)
{
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        size_t base = size_t(blockIdx.x) * region16 + (ring ? 0 : (size_t(it) * foot16) % total16);
        double2 *p = buf + base;
        if (do_write)
            for (size_t i = threadIdx.x; i < region16; i += 256 * 8) {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    p[i + 256 * u] = make_double2(double(i + it), 1.0);
            }

        __syncthreads();
        if (do_read)
            for (size_t i = threadIdx.x; i < region16; i += 256 * 8) {
                double2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = p[i + 256 * u];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    acc += v[u].x + v[u].y;
            }
        __syncthreads();
    }
    if (acc == 1.2345)
        sink[0] = acc;
}

int main()
{
    const size_t total = size_t(16) << 30;
    double2 *buf;
    double *sink;
    CK(hipMalloc(&buf, total));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(buf, 0, total));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int G = 2048;
    const int feet_mb[] = {64, 96, 128, 160, 192, 224, 256, 384, 512, 1024, 2048};
    printf("footprint_MB mode write+read_TB/s write_only_TB/s read_only_TB/s\n");
    for (int f : feet_mb) {
        const size_t foot = size_t(f) << 20, region16 = foot / G / 16, foot16 = foot / 16;
        const int iters = (int)((size_t(8) << 30) / foot);
        for (int ring = 1; ring >= 0; --ring) {
            double tb[3];
            for (int m = 0; m < 3; ++m) {
                const int w = m != 2, r = m != 1;
                roundtrip<<<G, 256>>>(buf, region16, foot16, total / 16 - foot16, 4, ring, w, r, sink);
                CK(hipEventRecord(a));
                roundtrip<<<G, 256>>>(buf, region16, foot16, total / 16 - foot16, iters, ring, w, r, sink);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                tb[m] = double(foot) * iters * (w + r) / (ms * 1e-3) / 1e12;
            }
            printf("%5d %s %.2f %.2f %.2f\n", f, ring ? "ring  " : "stream", tb[0], tb[1], tb[2]);
        }
    }
    return 0;
}
