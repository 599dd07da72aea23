// mdx_msd_fft.hpp — the MSD engine's own forward transform for n_fft = 2^18.
//
// The power spectrum sum_series |F_k|^2 of ~30 000 zero-padded real series of 10^5 points is
// HBM traffic, not arithmetic.  Through rocFFT the pipeline moves ~17.7 MB per series
// (gather with explicit zero padding, three transform passes each reading and writing the
// padded complex array, a separate |F|^2 pass).  Here it is 4.8 MB per series:
//
//   * two real series travel as ONE complex series z = x_a + i x_b; since x_a, x_b are real,
//     (|Z_k|^2 + |Z_{N-k}|^2) / 2 = |X_a,k|^2 + |X_b,k|^2, which is all the engine accumulates;
//   * four-step transform N = 512 x 512, index n = 512 n1 + n2, k = k1 + 512 k2:
//       pass A (msd_fft_cols_kernel): reads the positions where they lie ([frame][particle][xyz],
//         16 consecutive coordinates = 128 B per frame row, only the t < T_b rows — the zero
//         padding is never materialised), 512-point transforms over n1 in LDS, twiddle
//         W_N^(n2 k1), writes Y[k1][pair group][n2][pair] (8 pairs = 128 B contiguous);
//       pass B (msd_fft_rows_power_kernel): streams Y once (64 KB contiguous per step),
//         512-point transforms over n2 in LDS,
//         accumulates |Z|^2 over all pairs in registers — the spectrum itself is never written;
//   * msd_power_fold_kernel folds the N-point sums into the half spectrum the inverse step uses.
//
// One wave owns one 512-point transform (radix-8 Stockham, three in-place stages in an 8 KB LDS
// buffer; a wave's LDS operations execute in order, so no barrier is needed inside a transform).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace msdfft {

constexpr int R = 512;             // rows = columns
constexpr int N = R * R;           // 262144
constexpr int PG = 8;              // pairs per block, one wave each
constexpr int ZSTRIDE = R + 1;     // complex elements per pair buffer (+1: bank spread)
constexpr int THREADS = 64 * PG;

__device__ inline double2 cmul(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ inline double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ inline double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ inline double2 mul_mi(double2 a) { return make_double2(a.y, -a.x); }   // * (-i)

// forward 8-point transform, natural order in and out
__device__ inline void dft8(double2 (&a)[8])
{
    const double s = 0.70710678118654752440;
    const double2 b0 = cadd(a[0], a[4]), b4 = csub(a[0], a[4]);
    const double2 b1 = cadd(a[1], a[5]);
    double2 b5 = csub(a[1], a[5]);
    b5 = make_double2(s * (b5.x + b5.y), s * (b5.y - b5.x));          // * (1 - i)/sqrt2
    const double2 b2 = cadd(a[2], a[6]), b6 = mul_mi(csub(a[2], a[6]));
    const double2 b3 = cadd(a[3], a[7]);
    double2 b7 = csub(a[3], a[7]);
    b7 = make_double2(s * (b7.y - b7.x), -s * (b7.x + b7.y));         // * (-1 - i)/sqrt2
    const double2 c0 = cadd(b0, b2), c2 = csub(b0, b2);
    const double2 c1 = cadd(b1, b3), c3 = mul_mi(csub(b1, b3));
    const double2 c4 = cadd(b4, b6), c6 = csub(b4, b6);
    const double2 c5 = cadd(b5, b7), c7 = mul_mi(csub(b5, b7));
    a[0] = cadd(c0, c1);
    a[4] = csub(c0, c1);
    a[2] = cadd(c2, c3);
    a[6] = csub(c2, c3);
    a[1] = cadd(c4, c5);
    a[5] = csub(c4, c5);
    a[3] = cadd(c6, c7);
    a[7] = csub(c6, c7);
}

// exp(-2 pi i m / 512) from the half table h[m] (m < 256): the upper half is its negative
__device__ inline double2 tw512_at(const double2 *h, int m)
{
    const double2 t = h[m & 255];
    return (m & 256) ? make_double2(-t.x, -t.y) : t;
}

__device__ inline void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// In-place forward 512-point transform of z[0..511] (LDS) by one wave; entries with index
// >= n_live are taken as zero without being read.  tw: exp(-2 pi i m / 512), m < 256, in LDS
// (global twiddle loads would share the vmcnt queue with the streaming loads of the callers
// and make every transform wait for HBM).
__device__ inline void fft512_wave(double2 *z, const double2 *__restrict__ tw, int lane, int n_live)
{
    double2 v[8];
    // stage Ns = 1
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int idx = lane + 64 * r;
        v[r] = idx < n_live ? z[idx] : make_double2(0.0, 0.0);
    }
    dft8(v);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; ++r)
        z[lane * 8 + r] = v[r];
    wave_lds_fence();
    // stage Ns = 8
    {
        const int k = lane & 7;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            v[r] = z[lane + 64 * r];
            if (r)
                v[r] = cmul(v[r], tw512_at(tw, (r * k * 8) & 511));
        }
        dft8(v);
        wave_lds_fence();
        const int j0 = (lane >> 3) * 64 + k;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            z[j0 + 8 * r] = v[r];
        wave_lds_fence();
    }
    // stage Ns = 64
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        v[r] = z[lane + 64 * r];
        if (r)
            v[r] = cmul(v[r], tw512_at(tw, (r * lane) & 511));
    }
    dft8(v);
    wave_lds_fence();
#pragma unroll
    for (int r = 0; r < 8; ++r)
        z[lane + 64 * r] = v[r];
    wave_lds_fence();
}

// Pass A.  grid (pair groups, COLS_SPLIT column ranges, blocks of the trajectory), 512 threads.
// pos: float64 [B * t_block][n_total][3]; the chunk's coordinates are e in [0, n_elem) behind
// particle `first`; coordinate e belongs to pair e / 2 (real part: even e).
// A block walks its columns in order: while one column is transformed the next column's rows
// are already in flight (registers).
constexpr int COLS_SPLIT = 4;
constexpr int COLS_PER_BLOCK = R / COLS_SPLIT;

__global__ __launch_bounds__(THREADS, 4) void msd_fft_cols_kernel(
    const double *__restrict__ pos, int64_t n_total, int64_t first, int64_t n_elem, int64_t t_block,
    int zero_dims, int p_pad, const double2 *__restrict__ tw512, const double2 *__restrict__ twN,
    double2 *__restrict__ Y)
{
    __shared__ double2 zb[PG][ZSTRIDE];
    __shared__ double2 s_h[R / 2];   // exp(-2 pi i m / 512), m < 256
    __shared__ double2 s_n[R];       // exp(-2 pi i m / N),   m < 512
    const int pg = blockIdx.x, b = blockIdx.z;
    const int n2_begin = blockIdx.y * COLS_PER_BLOCK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < R / 2)
        s_h[tid] = tw512[tid];
    s_n[tid] = twN[tid];

    const int s = tid & 15, row0 = tid >> 4;
    const int64_t e = int64_t(pg) * 16 + s;
    const bool live = e < n_elem && !((zero_dims >> int(e % 3)) & 1);
    const double *src = pos + (int64_t(b) * t_block * n_total + first) * 3 + e;
    const int64_t row_stride = n_total * 3;
    double *dst = reinterpret_cast<double *>(&zb[s >> 1][0]) + (s & 1);
    const int p = tid & 7, kbase = tid >> 3;
    double2 *out = Y + ((int64_t(b) * R * (p_pad / PG) + pg) * R) * PG + p;
    const int64_t k1_stride = int64_t(p_pad / PG) * R * PG;

    // rows n1 = row0 + 32 i of column n2, i < 8 (t_block <= N / 2: at most 256 live rows)
    double x0, x1, x2, x3, x4, x5, x6, x7;
#define MDX_COLS_LOAD1(X, I, N2)                                                  \
    {                                                                             \
        const int64_t t = int64_t(row0 + 32 * (I)) * R + (N2);                    \
        X = (live && t < t_block) ? src[t * row_stride] : 0.0;                    \
    }
#define MDX_COLS_LOAD(N2)                                                         \
    MDX_COLS_LOAD1(x0, 0, N2) MDX_COLS_LOAD1(x1, 1, N2) MDX_COLS_LOAD1(x2, 2, N2) \
    MDX_COLS_LOAD1(x3, 3, N2) MDX_COLS_LOAD1(x4, 4, N2) MDX_COLS_LOAD1(x5, 5, N2) \
    MDX_COLS_LOAD1(x6, 6, N2) MDX_COLS_LOAD1(x7, 7, N2)
#define MDX_COLS_PUT()                                                            \
    dst[2 * (row0)] = x0, dst[2 * (row0 + 32)] = x1, dst[2 * (row0 + 64)] = x2,   \
    dst[2 * (row0 + 96)] = x3, dst[2 * (row0 + 128)] = x4, dst[2 * (row0 + 160)] = x5, \
    dst[2 * (row0 + 192)] = x6, dst[2 * (row0 + 224)] = x7
// result of column N2 for k1 = kbase + 64 I, twiddled by W_N^(N2 k1)
#define MDX_COLS_OUT(I, N2)                                                       \
    ([&]() {                                                                      \
        const int k1 = kbase + 64 * (I);                                          \
        const unsigned m = unsigned(k1) * unsigned(N2);                           \
        return cmul(zb[p][k1], cmul(tw512_at(s_h, int(m >> 9)), s_n[m & 511]));   \
    }())

    MDX_COLS_LOAD(n2_begin)
    __syncthreads();
    for (int n2 = n2_begin; n2 < n2_begin + COLS_PER_BLOCK; ++n2) {
        MDX_COLS_PUT();
        __syncthreads();
        {
            const int nxt = min(n2 + 1, R - 1);   // the last column reloads itself
            MDX_COLS_LOAD(nxt)
        }
        fft512_wave(zb[wave], s_h, lane, 256);
        __syncthreads();
        double2 *o = out + int64_t(kbase) * k1_stride + int64_t(n2) * PG;
#define MDX_COLS_STORE(I) o[int64_t(64 * (I)) * k1_stride] = MDX_COLS_OUT(I, n2);
        MDX_COLS_STORE(0) MDX_COLS_STORE(1) MDX_COLS_STORE(2) MDX_COLS_STORE(3)
        MDX_COLS_STORE(4) MDX_COLS_STORE(5) MDX_COLS_STORE(6) MDX_COLS_STORE(7)
#undef MDX_COLS_STORE
        __syncthreads();
    }
#undef MDX_COLS_LOAD1
#undef MDX_COLS_LOAD
#undef MDX_COLS_PUT
#undef MDX_COLS_OUT
}

// Pass B.  grid (k1 = 512, blocks of the trajectory), 512 threads; thread = k2.
// Pfull[b][k1][k2] = sum over all pairs of |Z_{k1 + 512 k2}|^2  (overwritten).
__global__ __launch_bounds__(THREADS, 4) void msd_fft_rows_power_kernel(
    const double2 *__restrict__ Y, int p_pad, const double2 *__restrict__ tw512,
    double *__restrict__ Pfull)
{
    __shared__ double2 zb[PG][ZSTRIDE];
    __shared__ double2 s_tw[R / 2];
    const int k1 = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < R / 2)
        s_tw[tid] = tw512[tid];
    // Y[b][k1][pair group][n2][pair]: one group's 512 x 8 values are 64 KB contiguous
    const int n_groups = p_pad / PG;
    const double2 *src = Y + (int64_t(b) * R + k1) * n_groups * (R * PG) + tid;
    double acc = 0.0;
    // eight named registers (an indexed array ends up in scratch memory here)
#define MDX_ROWS_LOAD(OFF)                                                               \
    p0 = src[(OFF)], p1 = src[(OFF) + THREADS], p2 = src[(OFF) + 2 * THREADS],            \
    p3 = src[(OFF) + 3 * THREADS], p4 = src[(OFF) + 4 * THREADS], p5 = src[(OFF) + 5 * THREADS], \
    p6 = src[(OFF) + 6 * THREADS], p7 = src[(OFF) + 7 * THREADS]
#define MDX_ROWS_PUT(I, V)                       \
    {                                            \
        const int idx = tid + THREADS * (I);     \
        zb[idx & 7][idx >> 3] = (V);             \
    }
    double2 p0, p1, p2, p3, p4, p5, p6, p7;
    MDX_ROWS_LOAD(0);
    for (int pg = 0; pg < n_groups; ++pg) {
        MDX_ROWS_PUT(0, p0) MDX_ROWS_PUT(1, p1) MDX_ROWS_PUT(2, p2) MDX_ROWS_PUT(3, p3)
        MDX_ROWS_PUT(4, p4) MDX_ROWS_PUT(5, p5) MDX_ROWS_PUT(6, p6) MDX_ROWS_PUT(7, p7)
        __syncthreads();
        {   // the next group's rows are in flight during this transform (the last iteration
            // reloads its own group: no branch)
            const int64_t nxt = int64_t(min(pg + 1, n_groups - 1)) * (R * PG);
            MDX_ROWS_LOAD(nxt);
        }
        fft512_wave(zb[wave], s_tw, lane, R);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PG; ++q) {
            const double2 v = zb[q][tid];
            acc = fma(v.x, v.x, fma(v.y, v.y, acc));
        }
        __syncthreads();
    }
#undef MDX_ROWS_LOAD
#undef MDX_ROWS_PUT
    Pfull[(int64_t(b) * R + k1) * R + tid] = acc;
}

// P[b][k] += (Pfull[b][k] + Pfull[b][N - k]) / 2 for the half spectrum k <= N/2, with Pfull
// stored as [k1][k2], k = k1 + 512 k2.
__global__ __launch_bounds__(256) void msd_power_fold_kernel(const double *__restrict__ Pfull,
                                                            int64_t nc, double *__restrict__ P)
{
    const int64_t k = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (k >= nc)
        return;
    const int64_t km = (N - k) & (N - 1);
    const double *pf = Pfull + int64_t(b) * N;
    const double s = 0.5 * (pf[(k & (R - 1)) * R + (k >> 9)] + pf[(km & (R - 1)) * R + (km >> 9)]);
    P[int64_t(b) * nc + k] += s;
}

}  // namespace msdfft
