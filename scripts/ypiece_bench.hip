// Can pass A of the MSD transform drop its block-wide transpositions?  Two questions, stores only / loads only, with
// the block structure of msd_fft_cols400_fused_kernel (block = 8 waves = the 8 pairs of a pair group; per (column,
// pair group) 400 lines of 128 B out, 200 rows of 128 B in):
//   stores: today a wave instruction writes 8 whole 128-byte lines (lane = (k1 line, pair)).  Variant PIECES: wave w
//           writes only ITS pair — 64 lanes, 64 different lines, 16 bytes each; the line is completed by the other
//           seven waves of the block, a little earlier or later (DRIFT: wave w sleeps w * 64 clocks first).
//   loads : today a wave instruction reads 8 whole rows.  Variant PIECES: wave w reads only its pair's 16 bytes of 64 rows.
// hipcc -O2 --offload-arch=gfx950 scripts/ypiece_bench.hip -o scripts/ypiece_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <bool PIECES, bool DRIFT>
__global__ __launch_bounds__(512) void k_store(double2 *__restrict__ Y, int n_pg, int cols_per_block)
{
    const int sg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.y * cols_per_block;
    for (int n2 = c0; n2 < c0 + cols_per_block && n2 < 512; ++n2)
        for (int q = 0; q < 8; ++q) {
            const int pg = sg * 8 + q;
            if (pg >= n_pg) break;
            if (DRIFT)
                for (int w = 0; w < wave; ++w)
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int k1 = PIECES ? lane + 64 * i : (tid >> 3) + 64 * i;
                const int p = PIECES ? wave : tid & 7;
                if (k1 < 400) {
                    const int64_t line = (int64_t(k1) * n_pg + pg) * 512 + n2;                      // [k1][pg][n2]
                    Y[line * 8 + p] = make_double2(double(k1), double(n2));
                }
            }
        }
}

// The whole-line store with Y laid out [pair group][k1][n2][8 pairs] instead of [k1][pair group][n2][8 pairs]: the 400 lines
// of an iteration are then 64 KB apart inside one 26 MB region instead of 61 MB apart.
__global__ __launch_bounds__(512) void k_store_pgmajor(double2 *__restrict__ Y, int n_pg, int cols_per_block)
{
    const int sg = blockIdx.x, tid = threadIdx.x;
    const int c0 = blockIdx.y * cols_per_block;
    for (int n2 = c0; n2 < c0 + cols_per_block && n2 < 512; ++n2)
        for (int q = 0; q < 8; ++q) {
            const int pg = sg * 8 + q;
            if (pg >= n_pg) break;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int k1 = (tid >> 3) + 64 * i;
                const int p = tid & 7;
                if (k1 < 400) {
                    const int64_t line = (int64_t(pg) * 400 + k1) * 512 + n2;
                    Y[line * 8 + p] = make_double2(double(k1), double(n2));
                }
            }
        }
}

template <bool PIECES>
__global__ __launch_bounds__(512) void k_load(const double2 *__restrict__ X, int64_t row_pairs, int n_pg,
                                              int cols_per_block, double *__restrict__ sink)
{
    const int sg = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = blockIdx.y * cols_per_block;
    double acc = 0.0;
    for (int n2 = c0; n2 < c0 + cols_per_block && n2 < 512; ++n2)
        for (int q = 0; q < 8; ++q) {
            const int pg = sg * 8 + q;
            if (pg >= n_pg) break;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = PIECES ? lane + 64 * i : (tid >> 3) + 64 * i;
                const int p = PIECES ? wave : tid & 7;
                if (row < 200) {
                    const int64_t t = int64_t(row) * 512 + n2;
                    if (t < 100000) {
                        const double2 v = X[t * row_pairs + int64_t(pg) * 8 + p];
                        acc += v.x + v.y;
                    }
                }
            }
        }
    if (acc == 1.2345e300) sink[0] = acc;
}

// HALF variants (what 256-thread blocks of four pairs would do): a block owns pairs 4 h .. 4 h + 3 of the pair groups of its
// super group, a wave instruction touches 16 lines with 64 bytes each.  SAME_XCD: the two halves are blocks i and i + 8
// (one XCD, round-robin dispatch), else neighbours i, i + 1 (two XCDs).
template <bool SAME_XCD>
__global__ __launch_bounds__(256) void k_store_half(double2 *__restrict__ Y, int n_pg, int cols_per_block)
{
    const int bx = blockIdx.x;
    const int half = SAME_XCD ? (bx >> 3) & 1 : bx & 1;
    const int sg = SAME_XCD ? ((bx >> 4) << 3) + (bx & 7) : bx >> 1;
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * cols_per_block;
    if (sg * 8 >= n_pg) return;
    for (int n2 = c0; n2 < c0 + cols_per_block && n2 < 512; ++n2)
        for (int q = 0; q < 8; ++q) {
            const int pg = sg * 8 + q;
            if (pg >= n_pg) break;
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int k1 = (tid >> 2) + 64 * i;
                const int p = 4 * half + (tid & 3);
                if (k1 < 400) {
                    const int64_t line = (int64_t(k1) * n_pg + pg) * 512 + n2;
                    Y[line * 8 + p] = make_double2(double(k1), double(n2));
                }
            }
        }
}

template <bool SAME_XCD>
__global__ __launch_bounds__(256) void k_load_half(const double2 *__restrict__ X, int64_t row_pairs, int n_pg,
                                                   int cols_per_block, double *__restrict__ sink)
{
    const int bx = blockIdx.x;
    const int half = SAME_XCD ? (bx >> 3) & 1 : bx & 1;
    const int sg = SAME_XCD ? ((bx >> 4) << 3) + (bx & 7) : bx >> 1;
    const int tid = threadIdx.x;
    const int c0 = blockIdx.y * cols_per_block;
    if (sg * 8 >= n_pg) return;
    double acc = 0.0;
    for (int n2 = c0; n2 < c0 + cols_per_block && n2 < 512; ++n2)
        for (int q = 0; q < 8; ++q) {
            const int pg = sg * 8 + q;
            if (pg >= n_pg) break;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (tid >> 2) + 64 * i;
                const int p = 4 * half + (tid & 3);
                if (row < 200) {
                    const int64_t t = int64_t(row) * 512 + n2;
                    if (t < 100000) {
                        const double2 v = X[t * row_pairs + int64_t(pg) * 8 + p];
                        acc += v.x + v.y;
                    }
                }
            }
        }
    if (acc == 1.2345e300) sink[0] = acc;
}

int main()
{
    const int n_pg = 938;
    const size_t ybytes = size_t(400) * n_pg * 512 * 128;
    const int64_t row_pairs = int64_t(n_pg) * 8 * 2;       // a frame row holds two groups' worth of pairs (10 000 particles)
    const size_t xbytes = size_t(100000) * row_pairs * 16;
    double2 *Y, *X;
    double *sink;
    if (hipMalloc(&Y, ybytes) != hipSuccess || hipMalloc(&X, xbytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) {
        printf("alloc failed\n");
        return 1;
    }
    hipMemset(Y, 0, ybytes);
    hipMemset(X, 0, xbytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int n_sg = (n_pg + 7) / 8;
    const int split = 17, cpb = (512 + split - 1) / split;
    dim3 grid(n_sg, split);
    auto timeit = [&](const char *name, auto launch, size_t bytes) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-46s %.3f ms  %.2f TB/s\n", name, best, bytes / best * 1e-9);
    };
    timeit("stores, whole lines per instruction (today)", [&] { k_store<false, false><<<grid, 512>>>(Y, n_pg, cpb); }, ybytes);
    timeit("stores, 16-byte pieces per wave", [&] { k_store<true, false><<<grid, 512>>>(Y, n_pg, cpb); }, ybytes);
    timeit("stores, 16-byte pieces per wave, waves drifting", [&] { k_store<true, true><<<grid, 512>>>(Y, n_pg, cpb); }, ybytes);
    const size_t rbytes = size_t(100000) * n_pg * 128;
    timeit("loads, whole rows per instruction (today)", [&] { k_load<false><<<grid, 512>>>(X, row_pairs, n_pg, cpb, sink); }, rbytes);
    timeit("loads, 16-byte pieces per wave", [&] { k_load<true><<<grid, 512>>>(X, row_pairs, n_pg, cpb, sink); }, rbytes);
    const int n_sg16 = (n_sg + 7) / 8 * 8;     // whole groups of 16 blocks for the SAME_XCD numbering
    dim3 gridh(2 * n_sg16, split);
    timeit("stores, 64-byte pieces, halves on two XCDs", [&] { k_store_half<false><<<gridh, 256>>>(Y, n_pg, cpb); }, ybytes);
    timeit("stores, 64-byte pieces, halves on one XCD", [&] { k_store_half<true><<<gridh, 256>>>(Y, n_pg, cpb); }, ybytes);
    timeit("loads, 64-byte pieces, halves on two XCDs", [&] { k_load_half<false><<<gridh, 256>>>(X, row_pairs, n_pg, cpb, sink); }, rbytes);
    timeit("loads, 64-byte pieces, halves on one XCD", [&] { k_load_half<true><<<gridh, 256>>>(X, row_pairs, n_pg, cpb, sink); }, rbytes);
    timeit("stores, whole lines, Y[pair group][k1][n2][8]", [&] { k_store_pgmajor<<<grid, 512>>>(Y, n_pg, cpb); }, ybytes);
    {   // pass A's whole memory side with no arithmetic at all: the store and the load kernel side by side on two streams
        hipStream_t s0, s1;
        hipStreamCreate(&s0);
        hipStreamCreate(&s1);
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(e0, s0);
            k_store<false, false><<<grid, 512, 0, s0>>>(Y, n_pg, cpb);
            k_load<false><<<grid, 512, 0, s1>>>(X, row_pairs, n_pg, cpb, sink);
            hipStreamSynchronize(s1);
            hipEventRecord(e1, s0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-46s %.3f ms  %.2f TB/s\n", "stores and loads (whole lines) side by side", best, (ybytes + rbytes) / best * 1e-9);
        best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipDeviceSynchronize();
            hipEventRecord(e0, s0);
            k_store_pgmajor<<<grid, 512, 0, s0>>>(Y, n_pg, cpb);
            k_load<false><<<grid, 512, 0, s1>>>(X, row_pairs, n_pg, cpb, sink);
            hipStreamSynchronize(s1);
            hipEventRecord(e1, s0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-46s %.3f ms  %.2f TB/s\n", "the same with Y[pair group][k1][n2][8]", best, (ybytes + rbytes) / best * 1e-9);
    }
    hipFree(Y);
    hipFree(X);
    return 0;
}
