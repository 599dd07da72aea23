// Does the 128-byte piece pattern of the MSD pass-A stores bound it?  Stores only, same block structure as
// msd_fft_cols400_fused_kernel (block = 8 pair groups x a range of columns; per (column, pair group) 400 pieces of
// 128 B), three layouts of Y.  hipcc -O2 --offload-arch=gfx950 scripts/ywrite_bench.hip -o scripts/ywrite_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int LAYOUT>
__global__ __launch_bounds__(512) void k(double2 *__restrict__ Y, int n_pg, int cols_per_block)
{
    const int sg = blockIdx.x, tid = threadIdx.x;
    const int c0 = blockIdx.y * cols_per_block;
    const int p = tid & 7;
    for (int n2 = c0; n2 < c0 + cols_per_block && n2 < 512; ++n2)
        for (int q = 0; q < 8; ++q) {
            const int pg = sg * 8 + q;
            if (pg >= n_pg) break;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int k1 = (tid >> 3) + 64 * i;
                if (k1 < 400) {
                    int64_t line;
                    if (LAYOUT == 0) line = (int64_t(k1) * n_pg + pg) * 512 + n2;                      // [k1][pg][n2]
                    else if (LAYOUT == 1) line = (int64_t(pg) * 512 + n2) * 400 + k1;                   // [pg][n2][k1]
                    else if (LAYOUT == 2) line = ((int64_t(pg) * 64 + (n2 >> 3)) * 400 + k1) * 8 + (n2 & 7);  // [pg][n2/8][k1][n2%8]
                    else line = ((int64_t(k1 >> 3) * n_pg + pg) * 512 + n2) * 8 + (k1 & 7);              // [k1/8][pg][n2][k1%8]
                    Y[line * 8 + p] = make_double2(double(k1), double(n2));
                }
            }
        }
}
int main()
{
    const int n_pg = 938;
    const size_t bytes = size_t(400) * n_pg * 512 * 128;
    double2 *Y;
    if (hipMalloc(&Y, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(Y, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n_sg = (n_pg + 7) / 8;
    for (int split : {8, 9, 16, 32}) {
        const int cpb = (512 + split - 1) / split;
        dim3 grid(n_sg, split);
        for (int layout = 0; layout < 4; ++layout) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (layout == 0) k<0><<<grid, 512>>>(Y, n_pg, cpb);
                else if (layout == 1) k<1><<<grid, 512>>>(Y, n_pg, cpb);
                else if (layout == 2) k<2><<<grid, 512>>>(Y, n_pg, cpb);
                else k<3><<<grid, 512>>>(Y, n_pg, cpb);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("split %2d (%5d blocks) layout %d: %.3f ms  %.2f TB/s\n", split, n_sg * split, layout, best, bytes / best * 1e-9);
        }
    }
    hipFree(Y);
    return 0;
}
