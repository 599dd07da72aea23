// mdx_msd_fft.hpp — the MSD engine's own forward transforms for n_fft = 400 x R2 (R2 = 2 .. 1024: 800 .. 409 600),
// 2^13 .. 2^16, 2^18, 2^19, 2^20.
//
// The power spectrum sum_series |F_k|^2 of ~30 000 zero-padded real series of 10^5 points is
// HBM traffic, not arithmetic.  Through rocFFT the pipeline moves ~17.7 MB per series at 2^18
// (gather with explicit zero padding, three transform passes each reading and writing the
// padded complex array, a separate |F|^2 pass).  Here it is 4.8 MB per series:
//
//   * two real series travel as ONE complex series z = x_a + i x_b; since x_a, x_b are real,
//     (|Z_k|^2 + |Z_{N-k}|^2) / 2 = |X_a,k|^2 + |X_b,k|^2, which is all the engine accumulates;
//   * four-step transform N = R1 x R2, index n = R2 n1 + n2, k = k1 + R1 k2:
//       pass A (msd_fft_cols_kernel): reads the positions where they lie ([frame][particle][xyz],
//         16 consecutive coordinates = 128 B per frame row, only the t < T_b rows — the zero
//         padding is never materialised), R1-point transforms over n1 in LDS, twiddle
//         W_N^(n2 k1), writes Y[k1][pair group][n2][pair] (8 pairs = 128 B contiguous) — the 400-, 512- and 1024-point
//         first factors with rows of >= 32 points Y[pair group][k1][n2][pair]: the lines an iteration stores then lie R2 x 128 B
//         apart inside one pair group's block instead of a whole k1 row apart (`pg_major`);
//       pass B (msd_fft_rows512_power_kernel; msd_fft_rows_power_kernel for 1024-point rows): streams
//         Y once (R2 x 128 B contiguous per step), R2-point transforms over n2 — first stage on the
//         registers the loads land in, middle stage in LDS, last stage back in registers —
//         accumulates |Z|^2 over all pairs in registers: the spectrum itself is never written;
//   * msd_power_fold_kernel folds the N-point sums into the half spectrum the inverse step uses.
//
// One wave owns one transform (radix-8 Stockham stages, a radix-16 last stage for 1024 points,
// in place in a wave-private LDS buffer; a wave's LDS operations execute in order, so there is
// no barrier inside a transform).  Shapes: 400 x R2 — pass A msd_fft_cols400_fused_kernel<R2> (in-place DIF stages
// 10, 10, 4 of the 400-point columns, the per-frame sums fused in), pass B msd_fft_rows_tiny_power_kernel (R2 = 2, 4), msd_fft_rows_short_power_kernel (8,
// 16, 32, 64), msd_fft_rows_mid_power_kernel (128, 256), msd_fft_rows512_power_kernel (512),
// msd_fft_rows_power_kernel (1024) —, 2^13 = 64 x 128,
// 2^14 = 64 x 256, 2^15 = 64 x 512, 2^16 = 64 x 1024 (8 short column transforms per wave at a time, the rows as above),
// 2^18 = 512 x 512, 2^19 = 1024 x 512, 2^20 = 1024 x 1024.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace msdfft {

constexpr int PG = 8;              // pairs per block, one wave each
constexpr int THREADS = 64 * PG;

__device__ inline double2 cmul(double2 a, double2 b)
{
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ inline double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ inline double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ inline double2 mul_mi(double2 a) { return make_double2(a.y, -a.x); }   // * (-i)

// w[r] = w1^r, r = 1 .. R - 1, by squarings and single steps (depth log2 R + 1, not R - 2): the twiddles of one butterfly
// from ONE table entry.  The LDS pipe of a CU — one for sixteen waves — is what the transform kernels queue for, the four
// VALUs are a third idle: R - 2 complex products (relative error <= R ulp) instead of R - 2 more 16-byte LDS reads.
template <int R> __device__ __forceinline__ void tw_powers(double2 w1, double2 *w)
{
    w[1] = w1;
#pragma unroll
    for (int r = 2; r < R; ++r)
        w[r] = (r & 1) ? cmul(w[r - 1], w1) : cmul(w[r / 2], w[r / 2]);
}

// forward 8-point transform, natural order in and out
__device__ __forceinline__ void dft8(double2 *a)
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

// forward 16-point transform, natural order in and out: two 8-point transforms of the even and
// odd samples, combined with W_16^k
__device__ __forceinline__ void dft16(double2 *a)
{
    double2 e[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        e[i] = a[2 * i];
        o[i] = a[2 * i + 1];
    }
    dft8(e);
    dft8(o);
    const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173;   // cos, sin(pi/8)
    const double r = 0.70710678118654752440;
    const double2 w[8] = {make_double2(1.0, 0.0),  make_double2(c1, -s1), make_double2(r, -r),
                          make_double2(s1, -c1),   make_double2(0.0, -1.0), make_double2(-s1, -c1),
                          make_double2(-r, -r),    make_double2(-c1, -s1)};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const double2 t = cmul(o[k], w[k]);
        a[k] = cadd(e[k], t);
        a[k + 8] = csub(e[k], t);
    }
}

template <int RADIX> __device__ __forceinline__ void dft(double2 *a)
{
    if (RADIX == 8)
        dft8(a);
    else
        dft16(a);
}

// exp(-2 pi i m / M) from the half table h[m] (m < M/2): the upper half is its negative
template <int M> __device__ inline double2 tw_at(const double2 *h, int m)
{
    const double2 t = h[m & (M / 2 - 1)];
    return (m & (M / 2)) ? make_double2(-t.x, -t.y) : t;
}

__device__ inline void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One in-place Stockham stage of an M-point transform by one wave: M / RADIX butterflies,
// M / RADIX / 64 per lane, all loaded before any is stored.  NS: product of the earlier radices.
// FIRST: entries with index >= n_live are taken as zero without being read.
template <int M, int RADIX, int NS, bool FIRST>
__device__ __forceinline__ void stockham_stage(double2 *z, const double2 *tw, int lane, int n_live)
{
    constexpr int T = M / RADIX;     // butterflies
    constexpr int PER = T / 64;      // per lane
    static_assert(PER >= 1, "at least one butterfly per lane");
    double2 v[PER][RADIX];
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        const int j = lane + 64 * p;
        const int k = j & (NS - 1);
#pragma unroll
        for (int r = 0; r < RADIX; ++r) {
            const int idx = j + r * T;
            v[p][r] = (!FIRST || idx < n_live) ? z[idx] : make_double2(0.0, 0.0);
        }
        if (NS > 1) {   // W_(RADIX NS)^(r k), r = 1 .. RADIX - 1, as powers of one table entry
            double2 w[RADIX];
            tw_powers<RADIX>(tw_at<M>(tw, (k * (M / (RADIX * NS))) & (M - 1)), w);
#pragma unroll
            for (int r = 1; r < RADIX; ++r)
                v[p][r] = cmul(v[p][r], w[r]);
        }
        dft<RADIX>(v[p]);
    }
    wave_lds_fence();
#pragma unroll
    for (int p = 0; p < PER; ++p) {
        const int j = lane + 64 * p;
        const int k = j & (NS - 1);
        const int j0 = (j - k) * RADIX + k;
#pragma unroll
        for (int r = 0; r < RADIX; ++r)
            z[j0 + r * NS] = v[p][r];
    }
    wave_lds_fence();
}

// The same stage for NC independent M-point transforms held by one wave (M / RADIX * NC == 64:
// every lane owns one butterfly), transform c at z + c * zs.
template <int M, int NC, int RADIX, int NS, bool FIRST>
__device__ __forceinline__ void stockham_stage_batch(double2 *z, int zs, const double2 *tw, int lane,
                                                     int n_live)
{
    constexpr int T = M / RADIX;
    static_assert(T * NC <= 64, "at most one butterfly per lane");
    const int c = lane / T, j = lane % T;
    const bool busy = T * NC == 64 || c < NC;       // the other lanes idle (R1 = 16: 32 columns)
    const int k = j & (NS - 1);
    double2 *zc = z + (busy ? c : 0) * zs;
    double2 v[RADIX];
#pragma unroll
    for (int r = 0; r < RADIX; ++r) {
        const int idx = j + r * T;
        v[r] = (!FIRST || idx < n_live) ? zc[idx] : make_double2(0.0, 0.0);
    }
    if (NS > 1) {
        double2 w[RADIX];
        tw_powers<RADIX>(tw_at<M>(tw, (k * (M / (RADIX * NS))) & (M - 1)), w);
#pragma unroll
        for (int r = 1; r < RADIX; ++r)
            v[r] = cmul(v[r], w[r]);
    }
    dft<RADIX>(v);
    wave_lds_fence();
    const int j0 = (j - k) * RADIX + k;
    if (busy)
#pragma unroll
        for (int r = 0; r < RADIX; ++r)
            zc[j0 + r * NS] = v[r];
    wave_lds_fence();
}

// forward 5-point transform, natural order in and out
__device__ __forceinline__ void dft5(double2 *a)
{
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;   // cos(2 pi/5), cos(4 pi/5)
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;    // sin(2 pi/5), sin(4 pi/5)
    const double2 t1 = cadd(a[1], a[4]), t2 = cadd(a[2], a[3]);
    const double2 d1 = csub(a[1], a[4]), d2 = csub(a[2], a[3]);
    const double2 a0 = a[0];
    const double2 m1 = make_double2(a0.x + c1 * t1.x + c2 * t2.x, a0.y + c1 * t1.y + c2 * t2.y);
    const double2 m2 = make_double2(a0.x + c2 * t1.x + c1 * t2.x, a0.y + c2 * t1.y + c1 * t2.y);
    // -i (s1 d1 + s2 d2) and -i (s2 d1 - s1 d2)
    const double2 u1 = make_double2(s1 * d1.y + s2 * d2.y, -(s1 * d1.x + s2 * d2.x));
    const double2 u2 = make_double2(s2 * d1.y - s1 * d2.y, -(s2 * d1.x - s1 * d2.x));
    a[0] = cadd(a0, cadd(t1, t2));
    a[1] = cadd(m1, u1);
    a[4] = csub(m1, u1);
    a[2] = cadd(m2, u2);
    a[3] = csub(m2, u2);
}

// forward 10-point transform, natural order in and out, decimation in frequency: five radix-2
// butterflies b_n = a_n + a_{n+5}, c_n = (a_n - a_{n+5}) W_10^n, then X[2m] = dft5(b)[m] and
// X[2m+1] = dft5(c)[m] — everything in place, ten values live.  HALF: a[5..9] are structurally zero.
template <bool HALF> __device__ __forceinline__ void dft10(double2 *a)
{
    const double c1 = 0.80901699437494742410, s1 = 0.58778525229247312917;   // cos, sin(pi/5)
    const double c2 = 0.30901699437494742410, s2 = 0.95105651629515357212;   // cos, sin(2 pi/5)
    const double2 w[5] = {make_double2(1.0, 0.0), make_double2(c1, -s1), make_double2(c2, -s2),
                          make_double2(-c2, -s2), make_double2(-c1, -s1)};
    double2 b[5], c[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        b[n] = HALF ? a[n] : cadd(a[n], a[n + 5]);
        const double2 d = HALF ? a[n] : csub(a[n], a[n + 5]);
        c[n] = n ? cmul(d, w[n]) : d;
    }
    dft5(b);
    dft5(c);
#pragma unroll
    for (int m = 0; m < 5; ++m) {
        a[2 * m] = b[m];
        a[2 * m + 1] = c[m];
    }
}

// forward 4-point transform, natural order in and out
__device__ __forceinline__ void dft4(double2 *a)
{
    const double2 t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]);
    const double2 t2 = cadd(a[1], a[3]), t3 = mul_mi(csub(a[1], a[3]));
    a[0] = cadd(t0, t2);
    a[1] = cadd(t1, t3);
    a[2] = csub(t0, t2);
    a[3] = csub(t1, t3);
}

// One in-place decimation-in-frequency stage of an M-point transform held by one wave: sub-transforms
// of length MS, radix RADIX.  Butterfly (block, t) reads z[base + t + r MS/RADIX], forms the
// RADIX-point transform, multiplies output k by W_MS^(t k) and writes it back to THE SAME slot
// base + t + k MS/RADIX.  A butterfly touches no slot of another one, so the rounds of a stage need
// no "all loaded before any is stored": one butterfly's RADIX values are all that is live (20 VGPRs
// for radix 5 where the Stockham stage above holds 40), and no fence is needed between rounds.
// After the last stage slot p holds X[digit-reversed p] (dif_slot below).  tw[m] = exp(-2 pi i m / M).
// FIRST: slots >= M / 2 are structurally zero (the zero padding) and are not read.
#ifndef DIF_GENERATED_TWIDDLES
#define DIF_GENERATED_TWIDDLES 1
#endif
template <int M, int MS, int RADIX, bool FIRST>
__device__ __forceinline__ void dif_stage(double2 *z, const double2 *tw, int lane)
{
    constexpr int T = M / RADIX;            // butterflies
    constexpr int SUB = MS / RADIX;         // butterflies per sub-transform = slot stride
    constexpr int ROUNDS = (T + 63) / 64;
#pragma unroll
    for (int p = 0; p < ROUNDS; ++p) {
        const int b = lane + 64 * p;
        if (T % 64 == 0 || b < T) {
            const int blk = SUB == 1 ? b : b / SUB, t = SUB == 1 ? 0 : b - blk * SUB;
            double2 *zz = z + blk * MS + t;
            double2 v[RADIX];
#pragma unroll
            for (int r = 0; r < RADIX; ++r)
                v[r] = (!FIRST || r < RADIX / 2) ? zz[r * SUB] : make_double2(0.0, 0.0);
            if (RADIX == 5)
                dft5(v);
            else if (RADIX == 4)
                dft4(v);
            else if (RADIX == 10)
                dft10<FIRST>(v);
            else
                dft8(v);
            if (SUB > 1 && DIF_GENERATED_TWIDDLES) {
                // W_MS^(t k), k = 1 .. RADIX - 1, as powers of the ONE table entry W_MS^t (tw_powers: pass A issued 102 LDS
                // instructions per wave and iteration, 18 of them these twiddle reads)
                // (one running power, not tw_powers' tree: ten of them live at once cost the 400-point pass A its last
                // registers — 128 and three spills against 108 — and a scratch reload drains the queue of row loads)
                const double2 w1 = tw[t * (M / MS)];
                double2 w = w1;
#pragma unroll
                for (int k = 0; k < RADIX; ++k) {
                    if (k) {
                        v[k] = cmul(v[k], w);
                        if (k + 1 < RADIX)
                            w = cmul(w, w1);
                    }
                    zz[k * SUB] = v[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < RADIX; ++k) {
                    if (k && SUB > 1)
                        v[k] = cmul(v[k], tw[t * k * (M / MS)]);
                    zz[k * SUB] = v[k];
                }
            }
        }
    }
    wave_lds_fence();
}

// slot of X[k] after the stages (10, 10, 4) of a 400-point dif transform: the digits of
// k = ka + 10 (kb + 10 kc) in reverse significance
__device__ __forceinline__ int dif_slot_400(int k)
{
    const int q1 = k / 10, ka = k - 10 * q1;
    const int kc = q1 / 10, kb = q1 - 10 * kc;
    return 40 * ka + 4 * kb + kc;
}

// The last stage (radix 4, sub-transforms of 4 consecutive slots) of a 400-point dif transform WITHOUT the write-back:
// butterfly b = lane + 64 p (< 100) reads slots 4 b .. 4 b + 3 and adds |X|^2 of its four outputs to Pq[4 p + k] — the
// slot 4 b + k holds X[digit-reversed] as after dif_stage<400, 4, 4>.  For kernels that only want the power spectrum
// (the single-pass kernel): eight LDS writes, seven reads and a fence fewer per transform.
__device__ __forceinline__ void dif_last4_power_400(const double2 *z, int lane, double *Pq)
{
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int b = lane + 64 * p;
        if (p == 0 || b < 100) {
            double2 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                v[k] = z[4 * b + k];
            dft4(v);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                Pq[4 * p + k] = fma(v[k].x, v[k].x, fma(v[k].y, v[k].y, Pq[4 * p + k]));
        }
    }
}

// In-place forward M-point transform of z[0..M) (LDS) by one wave.  tw: exp(-2 pi i m / M),
// m < M / 2, in LDS (global twiddle loads would share the vmcnt queue with the streaming loads
// of the callers and make every transform wait for HBM).
template <int M> __device__ __forceinline__ void fft_wave(double2 *z, const double2 *tw, int lane, int n_live)
{
    static_assert(M == 512 || M == 1024, "supported transform lengths");
    stockham_stage<M, 8, 1, true>(z, tw, lane, n_live);
    stockham_stage<M, 8, 8, false>(z, tw, lane, n_live);
    if (M == 512)
        stockham_stage<M, 8, 64, false>(z, tw, lane, n_live);
    else
        stockham_stage<M, M / 64, 64, false>(z, tw, lane, n_live);   // radix 16
}

// Pass A.  grid (pair groups, column ranges, blocks of the trajectory), 512 threads; gridDim.y
// (a power of two <= R2 / 2) splits the R2 columns so that small batches still fill the chip.
// pos: float64 [B * t_block][n_total][3]; the chunk's coordinates are e in [0, n_elem) behind
// particle `first`; coordinate e belongs to pair e / 2 (real part: even e).
// A block walks its columns in order: while one column is transformed the next column's rows
// are already in flight (registers).  t_block <= N / 2: at most R1 / 2 live rows.

template <int R1, int R2>
__global__ __launch_bounds__(THREADS, R1 == 512 ? 4 : 2) void msd_fft_cols_kernel(
    const double *__restrict__ pos, int64_t n_total, int64_t first, int64_t n_elem, int64_t t_block,
    int zero_dims, int p_pad, const double2 *__restrict__ tw_r1, const double2 *__restrict__ twN,
    double2 *__restrict__ Y, int pg_major)
{
    constexpr int ZS = R1 + 1;                  // +1: bank spread of the transposed reads
    constexpr int LOADS = R1 / 2 / 32;          // rows per thread and column (8 or 16)
    constexpr int OUTS = R1 / 64;               // k1 values per thread (8 or 16)
    __shared__ double2 zb[PG][ZS];
    __shared__ double2 s_h[R1 / 2];   // exp(-2 pi i m / R1), m < R1 / 2
    __shared__ double2 s_n[R2];       // exp(-2 pi i m / N),  m < R2
    const int pg = blockIdx.x, b = blockIdx.z;
    const int n2_count = R2 / int(gridDim.y);
    const int n2_begin = blockIdx.y * n2_count;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < R1 / 2; i += THREADS)
        s_h[i] = tw_r1[i];
    for (int i = tid; i < R2; i += THREADS)
        s_n[i] = twN[i];

    const int s = tid & 15, row0 = tid >> 4;
    const int64_t e = int64_t(pg) * 16 + s;
    const bool live = e < n_elem && !((zero_dims >> int(e % 3)) & 1);
    const double *src = pos + (int64_t(b) * t_block * n_total + first) * 3 + e;
    const int64_t row_stride = n_total * 3;
    double *dst = reinterpret_cast<double *>(&zb[s >> 1][0]) + (s & 1);
    const int p = tid & 7, kbase = tid >> 3;
    // (pg_major: Y[block][pair group][k1][n2][pair], see msd_fft_cols400_fused_kernel)
    const int64_t k1_stride = pg_major ? int64_t(R2) * PG : int64_t(p_pad / PG) * R2 * PG;
    const int64_t pg_stride = pg_major ? int64_t(R1) * R2 * PG : int64_t(R2) * PG;
    double2 *out = Y + int64_t(b) * (p_pad / PG) * (int64_t(R1) * R2 * PG) + int64_t(pg) * pg_stride + p;

    // rows n1 = row0 + 32 i of the current column (indexed by unrolled constants only)
    double x[LOADS];
#define MDX_COLS_LOAD(N2)                                                           \
    _Pragma("unroll") for (int i = 0; i < LOADS; ++i)                               \
    {                                                                               \
        const int64_t t = int64_t(row0 + 32 * i) * R2 + (N2);                       \
        x[i] = (live && t < t_block) ? src[t * row_stride] : 0.0;                   \
    }
    MDX_COLS_LOAD(n2_begin)
    __syncthreads();
    for (int n2 = n2_begin; n2 < n2_begin + n2_count; ++n2) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i)
            dst[2 * (row0 + 32 * i)] = x[i];
        __syncthreads();
        {
            const int nxt = min(n2 + 1, R2 - 1);   // the last column reloads itself
            MDX_COLS_LOAD(nxt)
        }
        fft_wave<R1>(zb[wave], s_h, lane, R1 / 2);
        __syncthreads();
        double2 *o = out + int64_t(kbase) * k1_stride + int64_t(n2) * PG;
        if constexpr (R1 >= 1024) {
            // W_N^(n2 k1), k1 = kbase + 64 i: W_N^(n2 kbase) times powers of W_N^(64 n2) — two table products and
            // OUTS - 1 complex products instead of OUTS table products; W_N^m = W_R1^(m / R2) * W_N^(m mod R2), m < N
            // (the 1024-point columns run one block per CU and queue for its LDS pipe: 2^19 30.6 -> 29.4 ms, 2^20
            // 31.2 -> 30.1; the 512-point columns, two blocks per CU, gained nothing from the chain of products)
            const unsigned mb = unsigned(kbase) * unsigned(n2), ms = 64u * unsigned(n2);
            double2 w = cmul(tw_at<R1>(s_h, int(mb / R2)), s_n[mb & (R2 - 1)]);
            const double2 ws = cmul(tw_at<R1>(s_h, int(ms / R2)), s_n[ms & (R2 - 1)]);
#pragma unroll
            for (int i = 0; i < OUTS; ++i) {
                o[int64_t(64 * i) * k1_stride] = cmul(zb[p][kbase + 64 * i], w);
                if (i + 1 < OUTS)
                    w = cmul(w, ws);
            }
        } else {
#pragma unroll
            for (int i = 0; i < OUTS; ++i) {
                const int k1 = kbase + 64 * i;
                // W_N^(n2 k1) = W_R1^(m / R2) * W_N^(m mod R2), m = n2 k1 < N
                const unsigned m = unsigned(k1) * unsigned(n2);
                const double2 w = cmul(tw_at<R1>(s_h, int(m / R2)), s_n[m & (R2 - 1)]);
                o[int64_t(64 * i) * k1_stride] = cmul(zb[p][k1], w);
            }
        }
        __syncthreads();
    }
#undef MDX_COLS_LOAD
}

// Pass A for R1 = 400 with the per-frame sums fused in (the separate sums kernel read the positions
// a second time: 12 GB per 5 000-particle group at C4).
//
// A block owns a SUPER GROUP of SG = 8 pair groups (64 pairs = 128 consecutive coordinates = 1 KB of
// every frame row) and a range of columns; per column it runs the eight pair groups one after the
// other through one stage -> transform -> store iteration, so what a
// block reads of one frame is one run of 1 KB.  While a pair group's rows sit in LDS, threads 0..199
// (one per live row = frame) add up x^2 and the coordinate sums of their row; after the eighth pair
// group the four sums of a frame leave as ONE 32-byte record part[super group][b][n2][row][4]
// (1/32 of the bytes read), and msd_partials_reduce_kernel adds the super groups in a fixed order:
// no atomics, run-to-run reproducible.
//
// Everything inside the loop is branch-free on the memory side: rows past the block's last frame and
// coordinates past the chunk read a valid dummy address and are zeroed when they are staged.  (With
// exec-masked loads the compiler drained vmcnt at every join — and the older kernel also spilled a few
// registers to scratch, whose reloads wait for vmcnt(0): every column waited for its own stores.)
constexpr int SG = 8;              // pair groups per super group
constexpr int SUMS_ROWS = 200;     // live rows of the 400-point first factor

// In: double — the float64 position store of the reference's Onsager (transport.py:932) — or float (round 5): float32
// frames resident in HBM, what a trajectory reader or a GPU MD engine delivers, are widened as they are staged, so a
// plain particle range of them needs no preparation pass (6 GB read + 12 GB written per 5 000-particle group of C4);
// a pair is then 8 bytes per lane and a row piece 64 bytes, everything behind the staging is the same code.
template <int R2, typename In = double>
__global__ __launch_bounds__(THREADS, 4) void msd_fft_cols400_fused_kernel(
    const In *__restrict__ pos, int64_t n_total, int64_t first, int64_t n_elem, int64_t t_block,
    int zero_dims, int p_pad, const double2 *__restrict__ tw_r1, const double2 *__restrict__ twN,
    double2 *__restrict__ Y, double2 *__restrict__ part, int head, int pg_major)
{
    // pg_major: Y is laid out [block][pair group][k1][n2][8 pairs] instead of [block][k1][pair group][n2][8 pairs] — the
    // 400 lines an iteration stores are then 64 KB (R2 = 512) apart inside one pair group's 26 MB instead of a whole
    // k1 row (61 MB at C4) apart: the store side alone runs at 5.5 instead of 4.9 TB/s (scripts/ypiece_bench.hip).
    // head: the chunk is entered `head` (< 16) coordinates BEFORE the first particle's x, so that every
    // 128-byte piece a lane group loads is one cache line (a group that starts 64 bytes off a line touched two
    // lines per piece: 31 % more fetched and 1.6 ms per step at C4, round 3).  n_elem counts from there; the
    // head coordinates belong to the neighbouring particles of the row (valid memory) and are staged as zeros;
    // the dimension of coordinate e is (e - head) % 3 = (e + dshift) % 3.
    const int dshift = (3 - head % 3) % 3;
    constexpr int R1 = 400, ZS = R1 + 1, LIVE = R1 / 2;
    constexpr int LOADS = (LIVE + 63) / 64;     // 4 row rounds of 64 rows (a lane loads a PAIR: 16 bytes)
    constexpr int OUTS = (R1 + 63) / 64;        // 7 line rounds
    static_assert(LIVE == SUMS_ROWS, "row count of the partial sums");
    __shared__ double2 zb[PG][ZS];
    __shared__ double2 s_h[R1];       // exp(-2 pi i m / 400), m < 400
    __shared__ double2 s_n[R2];       // exp(-2 pi i m / N),  m < R2
    __shared__ double2 s_acc[LIVE][2];   // running (x^2, x, y, z) sums of the column's rows
    const int sg = blockIdx.x, b = blockIdx.z;
    const int n_pg = p_pad / PG;
    const int pg0 = sg * SG;
    const int n_q = min(SG, n_pg - pg0);
    // column range of this block: gridDim.y need not divide R2 (it is chosen so that the grid is
    // close to a whole number of rounds of the chip's 512 block slots)
    const int n2_begin = int(int64_t(blockIdx.y) * R2 / gridDim.y);
    const int n2_count = int(int64_t(blockIdx.y + 1) * R2 / gridDim.y) - n2_begin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < R1; i += THREADS)
        s_h[i] = tw_r1[i];
    for (int i = tid; i < R2; i += THREADS)
        s_n[i] = twN[i];

    // loads: pair lp of the pair group (coordinates s, s + 1 = one complex value), rows row0 + 64 i.  16 bytes per
    // lane: 8-byte-per-lane loads run at 0.54-0.70 of the 16-byte rate (MI355X_MICROARCH.md), and the pair is what
    // gets staged anyway
    const int lp = tid & 7, s = 2 * lp, row0 = tid >> 3;
    const In *base = pos + (int64_t(b) * t_block * n_total + first) * 3 - head;
    const int64_t row_stride = n_total * 3;
    const int64_t istr = int64_t(64) * R2 * row_stride;       // rows row0 + 64 i -> + i * istr
    // stores: pair p of line kbase + 64 i
    const int p = tid & 7, kbase = tid >> 3;
    const int64_t k1_stride = pg_major ? int64_t(R2) * PG : int64_t(n_pg) * R2 * PG;
    const int64_t pg_stride = pg_major ? int64_t(R1) * R2 * PG : int64_t(R2) * PG;
    double2 *out = Y + int64_t(b) * n_pg * (int64_t(R1) * R2 * PG) + int64_t(kbase) * k1_stride + p;

    // rows of the current iteration: coordinate e = 16 (pg0 + q) + s, column n2
    const In *cur = base + (int64_t(row0) * R2 + n2_begin) * row_stride + int64_t(pg0) * 16 + s;
    struct alignas(sizeof(In)) Pair { In x, y; };   // two consecutive coordinates; the address is aligned to one of them
    Pair x[LOADS];
    // entry i of (column N2, pair group pg0 + Q) holds live values (else zeros are staged): the coordinates lie
    // in the chunk (the second of a pair may not: the chunk has 3 c coordinates) and their dimensions are kept,
    // the row exists (row0 + 64 i < 200) and its frame (row0 + 64 i) R2 + N2 lies in the block.  32-bit tests
    // against per-thread limits.
    const int e_lim = int(min<int64_t>(n_elem - int64_t(pg0) * 16 - s, 1 << 20));      // 16 Q (+ 1) < e_lim
    // (head < 16: only pair group 0 of the chunk has dead coordinates, those with 16 * 0 + s (+ 1) < head — a
    // wave-uniform test on the pair group and one on s3 = s + dshift, the one per-thread value the loop keeps
    // of `s`: a per-thread limit of its own was one VGPR too many, it spilled, and a scratch reload in this loop
    // drains vmcnt)
    const int s3 = s + dshift, head3 = head + dshift;
    const int t_lim = int(min<int64_t>(t_block - int64_t(row0) * R2, int64_t(1) << 30));   // 64 i R2 + N2 < t_lim
    const int i_lim = row0 < LIVE - 64 * (LOADS - 1) ? LOADS : LOADS - 1;                 // i < i_lim
#define MDX_FUSED_ROW(N2, I) ((I) < i_lim && 64 * R2 * (I) + (N2) < t_lim)
#define MDX_FUSED_OK0(N2, Q, I)                                                              \
    (16 * (Q) < e_lim && (pg0 + (Q) > 0 || s3 >= head3) && !((zero_dims >> ((pg0 + (Q) + s3) % 3)) & 1) && \
     MDX_FUSED_ROW(N2, I))
#define MDX_FUSED_OK1(N2, Q, I)                                                              \
    (16 * (Q) + 1 < e_lim && (pg0 + (Q) > 0 || s3 + 1 >= head3) && !((zero_dims >> ((pg0 + (Q) + s3 + 1) % 3)) & 1) && \
     MDX_FUSED_ROW(N2, I))
    // issue the loads of (column N2, pair group pg0 + Q) from CUR; dead rows read `base`; a pair whose second
    // coordinate lies past the chunk is read one coordinate earlier (its first coordinate arrives in .y), so
    // that no load reaches past the chunk's last coordinate
#define MDX_FUSED_LOAD(N2, Q, CUR)                                                          \
    _Pragma("unroll") for (int i = 0; i < LOADS; ++i)                                       \
    {                                                                                       \
        const bool row_ = 16 * (Q) < e_lim && MDX_FUSED_ROW(N2, i);                         \
        const In *q_ = row_ ? (CUR) + i * istr - (16 * (Q) + 1 < e_lim ? 0 : 1) : base;     \
        x[i] = *reinterpret_cast<const Pair *>(q_);                                         \
    }
    MDX_FUSED_LOAD(n2_begin, 0, cur)
    // the first rows are waited for here, once: the loop is then entered with no load pending and seven stores in
    // flight, and re-entered with four loads FOLLOWED BY seven stores, and the wait at its top is vmcnt(7 + ...)
    // on both paths (left to itself the compiler sank one of these loads below the placeholder stores)
#pragma unroll
    for (int i = 0; i < LOADS; ++i)
        asm volatile("" ::"v"(x[i].x), "v"(x[i].y) : "memory");
    // Seven placeholder stores into the first iteration's own output slots (overwritten there): the
    // loop is then entered with the same queue of memory operations as it is re-entered with — seven
    // loads followed by seven stores — and the compiler's wait for the loads at the top of the loop
    // becomes vmcnt(7 + ...) on both paths instead of draining the previous iteration's stores.
    {
        double2 *o = out + int64_t(pg0) * pg_stride + int64_t(n2_begin) * PG;
#pragma unroll
        for (int i = 0; i < OUTS; ++i) {
            const int k1 = kbase + 64 * i;
            o[int64_t(k1 < R1 ? 64 * i : 64 * (OUTS - 2)) * k1_stride] = make_double2(0.0, 0.0);
        }
    }
    __syncthreads();

    int n2 = n2_begin, q = 0;
    const int n_iter = n2_count * n_q;
    for (int it = 0; it < n_iter; ++it) {
        {
            const bool shifted = !(16 * q + 1 < e_lim);   // the pair was read one coordinate earlier
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                // (rows 200..255 of the last round receive zeros: the first stage does not read rows >= 200
                // and overwrites them; no branch here, or the compiler drains vmcnt at its join)
                const double v0 = double(shifted ? x[i].y : x[i].x);
                zb[lp][row0 + 64 * i] = make_double2(MDX_FUSED_OK0(n2, q, i) ? v0 : 0.0,
                                                     MDX_FUSED_OK1(n2, q, i) ? double(x[i].y) : 0.0);
            }
        }
        __syncthreads();
        const bool wrap = q + 1 == n_q;
        {   // per-frame sums of this pair group's 16 coordinates, every wave taking part: lane pair
            // (2 r, 2 r + 1) of wave w owns row 32 w + r, the even lane pairs 0..3, the odd one 4..7
            const int row = 32 * wave + (lane >> 1), half = lane & 1;
            double c0 = 0.0, c1 = 0.0, c2 = 0.0, d = 0.0;
            if (row < LIVE) {
#pragma unroll
                for (int k = 0; k < PG / 2; ++k) {
                    const double2 v = zb[4 * half + k][row];
                    d = fma(v.x, v.x, fma(v.y, v.y, d));
                    // coordinate 2 pp (+1), pp = 4 half + k, of pair group pg has dimension
                    // (pg + 2 pp (+1)) % 3 = (pg + 2 half + 2 k (+1)) % 3 (8 = 2 mod 3)
                    if ((2 * k) % 3 == 0) c0 += v.x; else if ((2 * k) % 3 == 1) c1 += v.x; else c2 += v.x;
                    if ((2 * k + 1) % 3 == 0) c0 += v.y; else if ((2 * k + 1) % 3 == 1) c1 += v.y; else c2 += v.y;
                }
            }
            // bring (c0, c1, c2) to absolute dimensions: rotate by (pg + 2 half) % 3
            const int rot = (pg0 + q + 2 * half + dshift) % 3;
            double sx = rot == 0 ? c0 : rot == 1 ? c2 : c1;
            double sy = rot == 0 ? c1 : rot == 1 ? c0 : c2;
            double sz = rot == 0 ? c2 : rot == 1 ? c1 : c0;
            // pairs 0..3 first, then 4..7: the even lane adds its odd neighbour's sums
            d += __shfl_xor(d, 1);
            sx += __shfl_xor(sx, 1);
            sy += __shfl_xor(sy, 1);
            sz += __shfl_xor(sz, 1);
            if (half == 0 && row < LIVE) {
                double2 u = s_acc[row][0], v = s_acc[row][1];
                if (q == 0)
                    u = v = make_double2(0.0, 0.0);
                u.x += d;
                u.y += sx;
                v.x += sy;
                v.y += sz;
                if (wrap) {
                    // wave-uniform 64-bit base + a 32-bit lane offset (a per-lane 64-bit index was spilled, and
                    // its scratch reload drains vmcnt)
                    double2 *o = part + ((int64_t(sg) * gridDim.z + b) * R2 + n2) * (2 * LIVE);
                    int r2_ = int(threadIdx.x) & ~1;   // 2 row (row = tid / 2), formed here: a copy kept across
                    asm volatile("" : "+v"(r2_));      // the loop was spilled, and its reload drains vmcnt
                    o[r2_] = u;
                    o[r2_ + 1] = v;
                } else {
                    s_acc[row][0] = u;
                    s_acc[row][1] = v;
                }
            }
        }
        __syncthreads();   // the sums read every pair's rows; the transforms below overwrite them
        // next iteration: the following pair group of this column, else the next column's first
        // (the last iteration reloads itself).  Issued BEFORE the transform so that the rows are in
        // flight while the block computes (between its transform and the next staging a block
        // issues nothing new, and the chip's memory queues ran dry during those phases); the
        // in-place dif stages below keep one butterfly live at a time, which leaves the registers.
        const int q_n = wrap ? 0 : q + 1;
        const int n2_n = wrap ? min(n2 + 1, n2_begin + n2_count - 1) : n2;
        cur += wrap ? (n2_n - n2) * row_stride - int64_t(n_q - 1) * 16 : 16;
        MDX_FUSED_LOAD(n2_n, q_n, cur)
        // three stages (10, 10, 4) instead of (4, 4, 5, 5): the transform phase is bound by the LDS
        // write path (a ds_write_b128 costs 13 cycles of it per wave: in-kernel phase timers show the
        // slowest wave 59 % of an iteration in here), and a stage fewer is 28 instead of 36 writes
        dif_stage<R1, 400, 10, true>(zb[wave], s_h, lane);
        dif_stage<R1, 40, 10, false>(zb[wave], s_h, lane);
        dif_stage<R1, 4, 4, false>(zb[wave], s_h, lane);
        __syncthreads();
        double2 *o = out + int64_t(pg0 + q) * pg_stride + int64_t(n2) * PG;
        {
            // W_N^(n2 k1), k1 = kbase + 64 i: W_N^(n2 kbase) times powers of W_N^(64 n2) — two table products and six
            // complex products instead of seven table products (fourteen 16-byte LDS reads);
            // W_N^m = W_R1^(m / R2) * W_N^(m mod R2), m < N
            const unsigned mb = unsigned(kbase) * unsigned(n2), ms = 64u * unsigned(n2);
            double2 w = cmul(s_h[mb / R2], s_n[mb & (R2 - 1)]);
            const double2 ws = cmul(s_h[ms / R2], s_n[ms & (R2 - 1)]);
#pragma unroll
            for (int i = 0; i < OUTS; ++i) {
                // the last round holds lines for kbase < 16 only: the other threads repeat round 5 (with its twiddle)
                const bool own = i < OUTS - 1 || kbase + 64 * i < R1;
                const int ii = own ? i : OUTS - 2;
                const int k1 = kbase + 64 * ii;
                if (i && own)
                    w = cmul(w, ws);
                o[int64_t(64 * ii) * k1_stride] = cmul(zb[p][dif_slot_400(k1)], w);
            }
        }
        __syncthreads();
        q = q_n;
        n2 = n2_n;
    }
#undef MDX_FUSED_LOAD
#undef MDX_FUSED_OK0
#undef MDX_FUSED_OK1
#undef MDX_FUSED_ROW
}

// D[b][t] += sum over the super groups (in order) of the partial x^2 sums, traj[b][t][k] likewise.
// Records are stored [super group][b][group g][f][4 doubles] with f = column * live + row inside the
// group's `nc` columns; the frame is t = row * r2 + g * nc + column of block b.  One thread per record.
__global__ __launch_bounds__(256) void msd_partials_reduce_kernel(const double2 *__restrict__ part,
                                                                  int n_sg, int r2, int nc, int live,
                                                                  int64_t t_block,
                                                                  double *__restrict__ traj,
                                                                  double *__restrict__ D)
{
    const int64_t idx = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int b = blockIdx.y;
    const int fr = nc * live;
    const int64_t per_b = int64_t(r2 / nc) * fr;
    if (idx >= per_b)
        return;
    const int g = int(idx / fr), f = int(idx % fr);
    const int64_t t = int64_t(f % live) * r2 + int64_t(g) * nc + f / live;
    if (t >= t_block)
        return;
    double d = 0.0, x = 0.0, y = 0.0, z = 0.0;
    for (int sg = 0; sg < n_sg; ++sg) {
        const double2 *r = part + ((int64_t(sg) * gridDim.y + b) * per_b + idx) * 2;
        const double2 u = r[0], v = r[1];
        d += u.x;
        x += u.y;
        y += v.x;
        z += v.y;
    }
    const int64_t fo = int64_t(b) * t_block + t;
    D[fo] += d;
    traj[3 * fo] += x;
    traj[3 * fo + 1] += y;
    traj[3 * fo + 2] += z;
}

// Pass A for the 64-point first factor (n_fft = 2^13 .. 2^16 = 64 x 128 .. 1024) with the per-frame sums fused in, as
// msd_fft_cols400_fused_kernel: a wave transforms NC = 8 neighbouring columns of its pair at once (one butterfly per lane
// and stage), so a block still moves 4 096 values per barrier, and the NC columns of a (k1, pair group) leave as one run
// of NC x 128 B.  A block owns a super group of SG pair groups and a range of column groups; per group of NC columns
// it runs the pair groups one after the other; while a pair group's rows sit in LDS every wave adds
// up x^2 and the coordinate sums of the 256 frames of the iteration (R1 / 2 live rows x NC columns);
// one 32-byte record per frame leaves after the last pair group.  Loads are branch-free (dead
// entries read a valid dummy address and are zeroed when staged), placeholder stores give the loop
// entry the back edge's queue of memory operations, and W_N^m comes from two short tables
// (m = 32 a + b) so that the block's LDS stays below half a CU's with the 8 KB of running sums.
template <int R1, int R2, typename In = double>     // In: double, or float (widened at the staging, as in the 400-point kernel)
__global__ __launch_bounds__(THREADS, 4) void msd_fft_cols_small_fused_kernel(
    const In *__restrict__ pos, int64_t n_total, int64_t first, int64_t n_elem, int64_t t_block,
    int zero_dims, int p_pad, const double2 *__restrict__ tw_r1, const double2 *__restrict__ twN,
    double2 *__restrict__ Y, double2 *__restrict__ part, int pg_major)
{
    static_assert(R1 == 64, "supported first factor");
    constexpr int NC = 512 / R1, ZS = R1 + 1, LIVE = R1 / 2;
    constexpr int PER = 8;                         // columns per thread on either side
    constexpr int FR = LIVE * NC;                  // frames of one iteration: 256
    static_assert(FR == 256, "frames per iteration");
    __shared__ double2 zb[PG][NC][ZS];
    __shared__ double2 s_h[R1 / 2];    // exp(-2 pi i m / R1), m < R1 / 2
    __shared__ double2 s_na[R2 / 32];  // exp(-2 pi i 32 a / N)
    __shared__ double2 s_nb[32];       // exp(-2 pi i b / N)
    __shared__ double2 s_acc[FR][2];   // running (x^2, x, y, z) sums of the iteration's frames
    const int sg = blockIdx.x, b = blockIdx.z;
    const int n_pg = p_pad / PG;
    const int pg0 = sg * SG;
    const int n_q = min(SG, n_pg - pg0);
    constexpr int NG = R2 / NC;                    // column groups
    const int g_begin = int(int64_t(blockIdx.y) * NG / gridDim.y);
    const int g_count = int(int64_t(blockIdx.y + 1) * NG / gridDim.y) - g_begin;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < R1 / 2; i += THREADS)
        s_h[i] = tw_r1[i];
    for (int i = tid; i < R2 / 32; i += THREADS)
        s_na[i] = twN[32 * i];
    if (tid < 32)
        s_nb[tid] = twN[tid];

    // loads: coordinate s of the pair group, live row n1, PER consecutive columns from c_in
    const int s = tid & 15, n1 = (tid >> 4) % LIVE, c_in = (tid >> 4) / LIVE * PER;
    const In *base = pos + (int64_t(b) * t_block * n_total + first) * 3;
    const int64_t row_stride = n_total * 3;
    double *dst = reinterpret_cast<double *>(&zb[s >> 1][c_in][0]) + (s & 1);
    // stores: pair p, line k1, PER consecutive columns from c_out
    const int p = tid & 7, k1 = (tid >> 3) % R1, c_out = (tid >> 3) / R1 * PER;
    // (pg_major: Y[block][pair group][k1][n2][pair], see msd_fft_cols400_fused_kernel)
    const int64_t k1_stride = pg_major ? int64_t(R2) * PG : int64_t(n_pg) * R2 * PG;
    const int64_t pg_stride = pg_major ? int64_t(R1) * R2 * PG : int64_t(R2) * PG;
    double2 *out = Y + int64_t(b) * n_pg * (int64_t(R1) * R2 * PG) + int64_t(k1) * k1_stride + p;

    // frame (n1 R2 + n2 + c_in + i) of coordinate 16 (pg0 + q) + s; 32-bit liveness tests
    const int e_lim = int(min<int64_t>(n_elem - int64_t(pg0) * 16 - s, 1 << 20));                    // 16 Q < e_lim
    const int t_lim = int(min<int64_t>(t_block - int64_t(n1) * R2 - c_in, int64_t(1) << 30));        // N2 + i < t_lim
#define MDX_SF_OK(N2, Q, I) \
    (16 * (Q) < e_lim && !((zero_dims >> ((pg0 + (Q) + s) % 3)) & 1) && (N2) + (I) < t_lim)
    const In *cur = base + (int64_t(n1) * R2 + g_begin * NC + c_in) * row_stride + int64_t(pg0) * 16 + s;
    In x[PER];
#define MDX_SF_LOAD(N2, Q, CUR)                                                              \
    _Pragma("unroll") for (int i = 0; i < PER; ++i)                                          \
    {                                                                                        \
        const In *q_ = MDX_SF_OK(N2, Q, i) ? (CUR) + i * row_stride : base;                  \
        x[i] = *q_;                                                                          \
    }
    MDX_SF_LOAD(g_begin * NC, 0, cur)
    {   // placeholder stores into the first iteration's own slots (see the 400-point kernel)
        double2 *o = out + int64_t(pg0) * pg_stride + int64_t(g_begin * NC + c_out) * PG;
#pragma unroll
        for (int i = 0; i < PER; ++i)
            o[int64_t(i) * PG] = make_double2(0.0, 0.0);
    }
    __syncthreads();

    int n2 = g_begin * NC, q = 0;
    const int n_iter = g_count * n_q;
    for (int it = 0; it < n_iter; ++it) {
#pragma unroll
        for (int i = 0; i < PER; ++i)
            dst[2 * (i * ZS + n1)] = MDX_SF_OK(n2, q, i) ? double(x[i]) : 0.0;
        __syncthreads();
        const bool wrap = q + 1 == n_q;
        {   // per-frame sums of this pair group: lane pair (2 f, 2 f + 1) owns frame f = 32 wave + lane / 2
            // = (column f / LIVE, row f % LIVE); the even lane pairs 0..3, the odd one 4..7
            const int f = 32 * wave + (lane >> 1), half = lane & 1;
            const int col = f / LIVE, row = f % LIVE;
            double c0 = 0.0, c1 = 0.0, c2 = 0.0, d = 0.0;
#pragma unroll
            for (int k = 0; k < PG / 2; ++k) {
                const double2 v = zb[4 * half + k][col][row];
                d = fma(v.x, v.x, fma(v.y, v.y, d));
                if ((2 * k) % 3 == 0) c0 += v.x; else if ((2 * k) % 3 == 1) c1 += v.x; else c2 += v.x;
                if ((2 * k + 1) % 3 == 0) c0 += v.y; else if ((2 * k + 1) % 3 == 1) c1 += v.y; else c2 += v.y;
            }
            const int rot = (pg0 + q + 2 * half) % 3;
            double sx = rot == 0 ? c0 : rot == 1 ? c2 : c1;
            double sy = rot == 0 ? c1 : rot == 1 ? c0 : c2;
            double sz = rot == 0 ? c2 : rot == 1 ? c1 : c0;
            d += __shfl_xor(d, 1);
            sx += __shfl_xor(sx, 1);
            sy += __shfl_xor(sy, 1);
            sz += __shfl_xor(sz, 1);
            if (half == 0) {
                double2 u = s_acc[f][0], v = s_acc[f][1];
                if (q == 0)
                    u = v = make_double2(0.0, 0.0);
                u.x += d;
                u.y += sx;
                v.x += sy;
                v.y += sz;
                if (wrap) {
                    double2 *o = part + (((int64_t(sg) * gridDim.z + b) * NG + n2 / NC) * FR + f) * 2;
                    o[0] = u;
                    o[1] = v;
                } else {
                    s_acc[f][0] = u;
                    s_acc[f][1] = v;
                }
            }
        }
        __syncthreads();   // the sums read every pair's rows; the transforms below overwrite them
        stockham_stage_batch<R1, NC, 8, 1, true>(&zb[wave][0][0], ZS, s_h, lane, LIVE);
        stockham_stage_batch<R1, NC, 8, 8, false>(&zb[wave][0][0], ZS, s_h, lane, R1);
        const int q_n = wrap ? 0 : q + 1;
        const int n2_n = wrap ? min(n2 + NC, (g_begin + g_count - 1) * NC) : n2;
        cur += wrap ? (n2_n - n2) * row_stride - int64_t(n_q - 1) * 16 : 16;
        MDX_SF_LOAD(n2_n, q_n, cur)
        __syncthreads();
        double2 *o = out + int64_t(pg0 + q) * pg_stride + int64_t(n2 + c_out) * PG;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            // W_N^(n2 k1) = W_R1^(m / R2) * W_N^(m mod R2), m = n2 k1 < N; the second from two tables
            const unsigned m = unsigned(k1) * unsigned(n2 + c_out + i);
            const unsigned ml = m & (R2 - 1);
            const double2 w = cmul(cmul(tw_at<R1>(s_h, int(m / R2)), s_na[ml >> 5]), s_nb[ml & 31]);
            o[int64_t(i) * PG] = cmul(zb[p][c_out + i][k1], w);
        }
        __syncthreads();
        q = q_n;
        n2 = n2_n;
    }
#undef MDX_SF_LOAD
#undef MDX_SF_OK
}

// Pass B.  grid (k1 < R1, blocks of the trajectory), 512 threads; thread = k2 (and k2 + 512).
// Pfull[part][b][k1][k2] (+)= sum over the pairs of this launch of |Z_{k1 + R1 k2}|^2; blockIdx.z =
// part: the pair groups are dealt round-robin to gridDim.z parts when R1 x blocks alone would not
// fill the chip (the fold kernel adds the parts in order).
template <int R1, int R2>
__global__ __launch_bounds__(THREADS, R2 == 512 ? 4 : 2) void msd_fft_rows_power_kernel(
    const double2 *__restrict__ Y, int p_pad, const double2 *__restrict__ tw_r2,
    double *__restrict__ Pfull, int accumulate, int pg_major = 0)
{
    constexpr int ZS = R2 + 1;
    constexpr int LOADS = R2 * PG / THREADS;    // 8 or 16 complex values per thread and group
    constexpr int OUTS = R2 / THREADS;          // 1 or 2 spectrum lines per thread
    __shared__ double2 zb[PG][ZS];
    __shared__ double2 s_tw[R2 / 2];
    const int k1 = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < R2 / 2; i += THREADS)
        s_tw[i] = tw_r2[i];
    // Y[b][k1][pair group][n2][pair]: one group's R2 x 8 values are contiguous
    const int n_all = p_pad / PG, n_parts = gridDim.z, part = blockIdx.z;
    const int n_groups = (n_all - part + n_parts - 1) / n_parts;      // groups part, part + n_parts, ...
    // (pg_major: Y[block][pair group][k1][n2][pair], see msd_fft_cols400_fused_kernel)
    const int64_t k1_stride = pg_major ? int64_t(R2) * PG : int64_t(n_all) * R2 * PG;
    const int64_t pg_stride = pg_major ? int64_t(R1) * R2 * PG : int64_t(R2) * PG;
    const int64_t g_stride = int64_t(n_parts) * pg_stride;
    const double2 *src = Y + int64_t(b) * n_all * (int64_t(R1) * R2 * PG) + int64_t(k1) * k1_stride + int64_t(part) * pg_stride + tid;
    if (n_groups <= 0) {   // more parts than pair groups: this part contributes nothing
        if (!accumulate)
#pragma unroll
            for (int i = 0; i < OUTS; ++i)
                Pfull[((int64_t(part) * gridDim.y + b) * R1 + k1) * R2 + tid + THREADS * i] = 0.0;
        return;
    }
    double acc[OUTS];
#pragma unroll
    for (int i = 0; i < OUTS; ++i)
        acc[i] = 0.0;
    // LOADS named registers (an indexed array of them is placed in scratch memory here)
    double2 p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10, p11, p12, p13, p14, p15;
#define MDX_ROWS_EACH(OP)                                                                          \
    OP(0, p0) OP(1, p1) OP(2, p2) OP(3, p3) OP(4, p4) OP(5, p5) OP(6, p6) OP(7, p7)                 \
    if (LOADS > 8) { OP(8, p8) OP(9, p9) OP(10, p10) OP(11, p11) OP(12, p12) OP(13, p13)            \
                     OP(14, p14) OP(15, p15) }
#define MDX_ROWS_LOAD(I, V) V = src[off + THREADS * (I)];
#define MDX_ROWS_PUT(I, V)                       \
    {                                            \
        const int idx = tid + THREADS * (I);     \
        zb[idx & 7][idx >> 3] = V;               \
    }
    {
        const int64_t off = 0;
        MDX_ROWS_EACH(MDX_ROWS_LOAD)
    }
    for (int pg = 0; pg < n_groups; ++pg) {
        MDX_ROWS_EACH(MDX_ROWS_PUT)
        __syncthreads();
        {   // the next group's rows are in flight during this transform (the last iteration
            // reloads its own group: no branch)
            const int64_t off = int64_t(min(pg + 1, n_groups - 1)) * g_stride;
            MDX_ROWS_EACH(MDX_ROWS_LOAD)
        }
        fft_wave<R2>(zb[wave], s_tw, lane, R2);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < OUTS; ++i)
#pragma unroll
            for (int q = 0; q < PG; ++q) {
                const double2 v = zb[q][tid + THREADS * i];
                acc[i] = fma(v.x, v.x, fma(v.y, v.y, acc[i]));
            }
        __syncthreads();
    }
#undef MDX_ROWS_EACH
#undef MDX_ROWS_LOAD
#undef MDX_ROWS_PUT
#pragma unroll
    for (int i = 0; i < OUTS; ++i) {
        double *o = Pfull + ((int64_t(part) * gridDim.y + b) * R1 + k1) * R2 + tid + THREADS * i;
        *o = accumulate ? *o + acc[i] : acc[i];
    }
}

// Pass B for R2 = 512 with the first and the last stage kept out of LDS.  The LDS write path (a
// ds_write_b128 costs 13 cycles of it per wave) is what bounds the general kernel above: per 64 KB
// piece it stages 8 values per thread, runs three in-place stages (8 reads + 8 writes each) and reads
// the result back for the power sums.  Here
//   * a thread's eight loads ARE the inputs of one first-stage butterfly (pair tid & 7, points
//     j + 64 r, j = tid >> 3), so stage 1 runs on the registers the loads land in and only its
//     outputs go to LDS (one barrier; no staging pass);
//   * stage 2 is the wave-private in-place stage;
//   * the outputs of stage 3 stay in registers: lane l of wave w holds X[l + 64 r] of pair w and adds
//     |X|^2 to eight running sums; the eight waves' sums meet once, at the end of the block.
// LDS traffic per piece: 8 writes + (8 + 7) reads + 8 writes + (8 + 7) reads per thread instead of
// 32 writes + 24 + 14 + 8 reads; two barriers per piece instead of three.
template <int R1>
__global__ __launch_bounds__(THREADS, 4) void msd_fft_rows512_power_kernel(
    const double2 *__restrict__ Y, int p_pad, const double2 *__restrict__ tw_r2,
    double *__restrict__ Pfull, int accumulate, int pg_major = 0)
{
    constexpr int R2 = 512, ZS = R2 + 1;
    __shared__ double2 zb[PG][ZS];
    __shared__ double2 s_tw[R2 / 2];
    const int k1 = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < R2 / 2; i += THREADS)
        s_tw[i] = tw_r2[i];
    const int n_all = p_pad / PG, n_parts = gridDim.z, part = blockIdx.z;
    const int n_groups = (n_all - part + n_parts - 1) / n_parts;      // groups part, part + n_parts, ...
    double *pout = Pfull + ((int64_t(part) * gridDim.y + b) * R1 + k1) * R2 + tid;
    if (n_groups <= 0) {   // more parts than pair groups: this part contributes nothing
        if (!accumulate)
            *pout = 0.0;
        return;
    }
    // (pg_major: Y[block][pair group][k1][n2][pair], see msd_fft_cols400_fused_kernel; a (k1, pair group) run is 64 KB either way)
    const int64_t k1_stride = pg_major ? int64_t(R2) * PG : int64_t(n_all) * R2 * PG;
    const int64_t pg_stride = pg_major ? int64_t(R1) * R2 * PG : int64_t(R2) * PG;
    const int64_t g_stride = int64_t(n_parts) * pg_stride;
    const double2 *src = Y + int64_t(b) * n_all * (int64_t(R1) * R2 * PG) + int64_t(k1) * k1_stride + int64_t(part) * pg_stride + tid;
    const int p = tid & 7, j = tid >> 3;
    double2 v[8];
    double acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        acc[r] = 0.0;
        v[r] = src[THREADS * r];
    }
    __syncthreads();   // twiddle table
    for (int pg = 0; pg < n_groups; ++pg) {
        // stage 1 (NS = 1, no twiddles) on the loaded values: butterfly j of pair p -> points 8 j + r
        dft8(v);
#pragma unroll
        for (int r = 0; r < 8; ++r)
            zb[p][8 * j + r] = v[r];
        __syncthreads();
        {   // the next group's rows are in flight during stages 2 and 3 (the last iteration reloads
            // its own group: no branch)
            const int64_t off = int64_t(min(pg + 1, n_groups - 1)) * g_stride;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                v[r] = src[off + THREADS * r];
        }
        stockham_stage<R2, 8, 8, false>(zb[wave], s_tw, lane, R2);
        {   // stage 3 (NS = 64): butterfly l reads points l + 64 r, output r is X[l + 64 r]
            double2 u[8];
#pragma unroll
            for (int r = 0; r < 8; ++r)
                u[r] = zb[wave][lane + 64 * r];
            {   // W_512^(r lane) as powers of W_512^lane
                double2 w[8];
                tw_powers<8>(tw_at<R2>(s_tw, lane), w);
#pragma unroll
                for (int r = 1; r < 8; ++r)
                    u[r] = cmul(u[r], w[r]);
            }
            dft8(u);
#pragma unroll
            for (int r = 0; r < 8; ++r)
                acc[r] = fma(u[r].x, u[r].x, fma(u[r].y, u[r].y, acc[r]));
        }
        __syncthreads();
    }
    // the eight waves' sums (one pair index each), added in wave order
    double *red = reinterpret_cast<double *>(&zb[0][0]);      // 8 x 512 doubles
#pragma unroll
    for (int r = 0; r < 8; ++r)
        red[wave * R2 + lane + 64 * r] = acc[r];
    __syncthreads();
    double total = 0.0;
#pragma unroll
    for (int w = 0; w < PG; ++w)
        total += red[w * R2 + tid];
    *pout = accumulate ? *pout + total : total;
}

// Pass B for R2 = 1024 in the same form (round 5): 400 x 1024 = 409 600, 64 x 1024 = 2^16 and 1024 x 1024 = 2^20
// went through the general kernel above — staging pass, three in-place LDS stages, read-back, three barriers per
// 128 KB piece — at 3.5 - 3.7 TB/s where the 512-point kernel streams Y at 6.0.  Here a thread's SIXTEEN loads (pair
// tid & 7, points j + 64 r, j = tid >> 3) are the inputs of first-stage butterfly j of 1024 = 16 x 8 x 8: the
// radix-16 stage runs on the registers the loads land in, its outputs go to LDS once (one barrier), the radix-8
// stage with NS = 16 is the wave-private in-place one (two butterflies per lane), and the last radix-8 stage
// (NS = 128: butterfly l reads points l + 128 r, output r is X[l + 128 r]) ends in registers — lane l of wave w holds
// X[l + 64 q + 128 r] of pair w, q < 2, and adds |X|^2 to sixteen running sums.  128 KB of LDS per block: one block
// per CU, whose next piece (16 x 16 B per thread) is in flight during stages 2 and 3.
template <int R1>
__global__ __launch_bounds__(THREADS, 2) void msd_fft_rows1024_power_kernel(
    const double2 *__restrict__ Y, int p_pad, const double2 *__restrict__ tw_r2,
    double *__restrict__ Pfull, int accumulate, int pg_major = 0)
{
    constexpr int R2 = 1024, ZS = R2 + 1;
    __shared__ double2 zb[PG][ZS];
    __shared__ double2 s_tw[R2 / 2];
    const int k1 = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < R2 / 2; i += THREADS)
        s_tw[i] = tw_r2[i];
    const int n_all = p_pad / PG, n_parts = gridDim.z, part = blockIdx.z;
    const int n_groups = (n_all - part + n_parts - 1) / n_parts;      // groups part, part + n_parts, ...
    double *pout = Pfull + ((int64_t(part) * gridDim.y + b) * R1 + k1) * R2 + tid;
    if (n_groups <= 0) {   // more parts than pair groups: this part contributes nothing
        if (!accumulate) {
            pout[0] = 0.0;
            pout[THREADS] = 0.0;
        }
        return;
    }
    const int64_t k1_stride = pg_major ? int64_t(R2) * PG : int64_t(n_all) * R2 * PG;
    const int64_t pg_stride = pg_major ? int64_t(R1) * R2 * PG : int64_t(R2) * PG;
    const int64_t g_stride = int64_t(n_parts) * pg_stride;
    const double2 *src = Y + int64_t(b) * n_all * (int64_t(R1) * R2 * PG) + int64_t(k1) * k1_stride + int64_t(part) * pg_stride + tid;
    const int p = tid & 7, j = tid >> 3;
    double2 v[16];
    double acc[2][8];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        v[r] = src[THREADS * r];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 8; ++r)
            acc[q][r] = 0.0;
    __syncthreads();   // twiddle table
    for (int pg = 0; pg < n_groups; ++pg) {
        // stage 1 (radix 16, NS = 1, no twiddles) on the loaded values: butterfly j of pair p -> points 16 j + r
        dft16(v);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            zb[p][16 * j + r] = v[r];
        __syncthreads();
        {   // the next group's rows are in flight during stages 2 and 3 (the last iteration reloads its own group)
            const int64_t off = int64_t(min(pg + 1, n_groups - 1)) * g_stride;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                v[r] = src[off + THREADS * r];
        }
        stockham_stage<R2, 8, 16, false>(zb[wave], s_tw, lane, R2);
#pragma unroll
        for (int q = 0; q < 2; ++q) {   // stage 3 (radix 8, NS = 128): butterfly l = lane + 64 q
            const int l = lane + 64 * q;
            double2 u[8];
#pragma unroll
            for (int r = 0; r < 8; ++r)
                u[r] = zb[wave][l + 128 * r];
            {   // W_1024^(r l) as powers of W_1024^l
                double2 w[8];
                tw_powers<8>(tw_at<R2>(s_tw, l), w);
#pragma unroll
                for (int r = 1; r < 8; ++r)
                    u[r] = cmul(u[r], w[r]);
            }
            dft8(u);
#pragma unroll
            for (int r = 0; r < 8; ++r)
                acc[q][r] = fma(u[r].x, u[r].x, fma(u[r].y, u[r].y, acc[q][r]));
        }
        __syncthreads();
    }
    // the eight waves' sums (one pair index each), added in wave order
    double *red = reinterpret_cast<double *>(&zb[0][0]);      // 8 x 1024 doubles
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 8; ++r)
            red[wave * R2 + lane + 64 * q + 128 * r] = acc[q][r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        double total = 0.0;
#pragma unroll
        for (int w = 0; w < PG; ++w)
            total += red[w * R2 + tid + THREADS * i];
        pout[THREADS * i] = accumulate ? pout[THREADS * i] + total : total;
    }
}

// Pass B for 8-, 16-, 32- and 64-point rows (n_fft = 3 200 = 400 x 8, 6 400 = 400 x 16, 12 800 = 400 x 32, 25 600 = 400 x 64:
// blocks of 801 .. 1 600, 1 601 .. 3 200, 4 097 .. 6 400 and 8 193 .. 12 800 frames — C4 with eight blocks is the last, whose half-transformed
// block is then as large as with one block; padded to 2^15 it was 28 % larger).
// A (k1, pair group) run of Y is R2 n2 x 8 pairs = R2 x 128 B, consecutive pair groups are contiguous.  Wave g of
// a block streams pair groups g, g + 8, ... of the block's share on its own: lane (pair p = lane & 7, j = lane >> 3)
// loads n2 = j + 8 r, r < F = R2 / 8 — a wave instruction reads 1 KB contiguous, and the F values ARE butterfly j of the
// first radix-F stage.  Stage 1 on those registers, outputs times W_R2^(j ka) (F - 1 per-thread constants) into the
// wave's own 8 x R2 slots, stage 2 (radix 8; lanes j < F) back into registers: lane (p, ka) holds X[ka + F kb] of pair
// p and adds |X|^2 to eight running sums.  Nothing in the loop is shared between waves: no block barrier until the
// sums meet at the end.
template <int F> __device__ __forceinline__ void dft_first(double2 *a)
{
    if constexpr (F == 1) {
        (void)a;
    } else if constexpr (F == 8) {
        dft8(a);
    } else if constexpr (F == 4) {
        const double2 t0 = cadd(a[0], a[2]), t1 = csub(a[0], a[2]);
        const double2 t2 = cadd(a[1], a[3]), t3 = mul_mi(csub(a[1], a[3]));
        a[0] = cadd(t0, t2);
        a[1] = cadd(t1, t3);
        a[2] = csub(t0, t2);
        a[3] = csub(t1, t3);
    } else {
        const double2 t0 = cadd(a[0], a[1]), t1 = csub(a[0], a[1]);
        a[0] = t0;
        a[1] = t1;
    }
}

template <int R1, int R2>
__global__ __launch_bounds__(THREADS, 4) void msd_fft_rows_short_power_kernel(
    const double2 *__restrict__ Y, int p_pad, const double2 *__restrict__ tw_r2,
    double *__restrict__ Pfull, int accumulate, int pg_major = 0)
{
    static_assert(R2 == 8 || R2 == 16 || R2 == 32 || R2 == 64, "row lengths of this kernel");
    constexpr int F = R2 / 8, ZS = R2 + 1;
    __shared__ double2 zb[PG][PG][ZS];            // [wave][pair][point]
    const int k1 = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_all = p_pad / PG, n_parts = gridDim.z, part = blockIdx.z;
    const int n_groups = (n_all - part + n_parts - 1) / n_parts;      // groups part, part + n_parts, ... of the block
    double *pout = Pfull + ((int64_t(part) * gridDim.y + b) * R1 + k1) * R2 + tid;
    const int n_mine = (n_groups - wave + PG - 1) / PG;               // ... of which this wave takes wave, wave + 8, ...
    const int p = lane & 7, j = lane >> 3;
    // W_R2^(j r), r = 1 .. F - 1, from the half table exp(-2 pi i m / R2), m < R2 / 2
    double2 w[F];
    w[0] = make_double2(1.0, 0.0);
#pragma unroll
    for (int r = 1; r < F; ++r)
        w[r] = tw_at<R2>(tw_r2, j * r);
    double acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
        acc[r] = 0.0;
    if (n_mine > 0) {
        // (pg_major: Y[block][pair group][k1][n2][pair], see msd_fft_cols400_fused_kernel)
        const int64_t k1_stride = pg_major ? int64_t(R2) * PG : int64_t(n_all) * R2 * PG;
        const int64_t pg_stride = pg_major ? int64_t(R1) * R2 * PG : int64_t(R2) * PG;
        const int64_t g_stride = int64_t(n_parts) * PG * pg_stride;     // this wave's next pair group
        const double2 *src = Y + int64_t(b) * n_all * (int64_t(R1) * R2 * PG) + int64_t(k1) * k1_stride +
                             (part + int64_t(wave) * n_parts) * pg_stride + lane;
        double2 v[F];
#pragma unroll
        for (int r = 0; r < F; ++r)
            v[r] = src[64 * r];
        double2 *z = &zb[wave][p][0];
        for (int it = 0; it < n_mine; ++it) {
            dft_first<F>(v);
#pragma unroll
            for (int r = 0; r < F; ++r)
                z[8 * r + j] = r ? cmul(v[r], w[r]) : v[r];
            {   // the next pair group is in flight during stage 2 (the last iteration reloads its own: no branch)
                const int64_t off = int64_t(min(it + 1, n_mine - 1)) * g_stride;
#pragma unroll
                for (int r = 0; r < F; ++r)
                    v[r] = src[off + 64 * r];
            }
            wave_lds_fence();
            if (F == 8 || j < F) {
                double2 u[8];
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    u[r] = z[8 * j + r];        // lane (p, ka = j): the eight first-stage outputs ka of butterflies r
                dft8(u);
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    acc[r] = fma(u[r].x, u[r].x, fma(u[r].y, u[r].y, acc[r]));
            }
            wave_lds_fence();
        }
    }
    // the sums of the 8 waves x 8 pairs, added in a fixed order: red[wave * 8 + pair][k2], k2 = ka + F kb
    __syncthreads();
    double *red = reinterpret_cast<double *>(&zb[0][0][0]);      // 64 x R2 doubles
    if (F == 8 || j < F) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
            red[(wave * PG + p) * R2 + j + F * r] = acc[r];
    }
    __syncthreads();
    if (tid < R2) {
        double total = 0.0;
        for (int i = 0; i < PG * PG; ++i)
            total += red[i * R2 + tid];
        *pout = accumulate ? *pout + total : total;
    }
}

// Pass B for 2- and 4-point rows (n_fft = 800 = 400 x 2, 1 600 = 400 x 4: blocks of 201 .. 400 and 401 .. 800 frames).
// A (k1, pair group) run of Y is R2 lines of 128 B; thread (pair p = tid & 7, slot = tid >> 3) takes the pair groups
// slot, slot + 64, ... of the block's share, loads its pair's R2 values (a wave instruction reads eight whole lines),
// transforms them in registers and adds |X|^2 to R2 running sums.  The sums of the block meet in a fixed order:
// xor-shuffles inside a wave, then the eight waves one after the other.
template <int R1, int R2>
__global__ __launch_bounds__(THREADS, 4) void msd_fft_rows_tiny_power_kernel(
    const double2 *__restrict__ Y, int p_pad, double *__restrict__ Pfull, int accumulate)
{
    static_assert(R2 == 2 || R2 == 4, "row lengths of this kernel");
    __shared__ double red[PG][R2];
    const int k1 = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_all = p_pad / PG, n_parts = gridDim.z, part = blockIdx.z;
    const int n_groups = (n_all - part + n_parts - 1) / n_parts;      // groups part, part + n_parts, ... of the block
    const int p = tid & 7, slot = tid >> 3;
    const double2 *src = Y + ((int64_t(b) * R1 + k1) * n_all + part) * (R2 * PG) + p;
    double acc[R2];
#pragma unroll
    for (int r = 0; r < R2; ++r)
        acc[r] = 0.0;
    for (int g = slot; g < n_groups; g += THREADS / PG) {
        const double2 *q = src + int64_t(g) * n_parts * (R2 * PG);
        double2 v[R2];
#pragma unroll
        for (int r = 0; r < R2; ++r)
            v[r] = q[r * PG];
        dft_first<R2>(v);
#pragma unroll
        for (int r = 0; r < R2; ++r)
            acc[r] = fma(v[r].x, v[r].x, fma(v[r].y, v[r].y, acc[r]));
    }
#pragma unroll
    for (int r = 0; r < R2; ++r) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
            acc[r] += __shfl_xor(acc[r], m);
        if (lane == 0)
            red[wave][r] = acc[r];
    }
    __syncthreads();
    if (tid < R2) {
        double total = 0.0;
        for (int w = 0; w < PG; ++w)
            total += red[w][tid];
        double *pout = Pfull + ((int64_t(part) * gridDim.y + b) * R1 + k1) * R2 + tid;
        *pout = accumulate ? *pout + total : total;
    }
}

// Pass B for 128- and 256-point rows (n_fft = 51 200 = 400 x 128 and 102 400 = 400 x 256: blocks of 12 801 .. 25 600
// and 25 601 .. 51 200 frames, which the power-of-two shapes padded by up to 2 x and 204 800 points by up to 3 x).
// The structure of msd_fft_rows512_power_kernel with a short first stage: a block streams 64 KB pieces of Y — now
// NPG = 512 / R2 whole pair groups — and thread (pair p = tid & 7, j = tid >> 3) loads piece elements tid + 512 r:
// pair group r / F, points j + 64 (r % F) with F = R2 / 64.  Those ARE the inputs of NPG radix-F butterflies of the
// first Stockham stage (F = 2 or 4, no twiddles), done on the registers; stage 2 (radix 8, NS = F) runs in LDS, a
// wave owning TPW = NPG transforms of its pair index side by side (R2 / 8 butterflies each: 64 lanes in all); stage 3
// (radix 8, NS = R2 / 8) ends in registers: lane (c, l) of wave w holds X[l + (R2 / 8) r] of transform (w TPW + c) and adds
// |X|^2 to eight running sums.
template <int R1, int R2>
__global__ __launch_bounds__(THREADS, 4) void msd_fft_rows_mid_power_kernel(
    const double2 *__restrict__ Y, int p_pad, const double2 *__restrict__ tw_r2,
    double *__restrict__ Pfull, int accumulate, int pg_major = 0)
{
    static_assert(R2 == 128 || R2 == 256, "row lengths of this kernel");
    constexpr int F = R2 / 64;              // first radix
    constexpr int NPG = 512 / R2;            // pair groups per 64 KB piece
    constexpr int NT = NPG * PG;             // transforms per piece
    constexpr int TPW = NT / PG;             // transforms per wave (= NPG)
    constexpr int T8 = R2 / 8;               // butterflies of a radix-8 stage
    constexpr int ZS = R2 + 1;
    __shared__ double2 zb[NT][ZS];
    __shared__ double2 s_tw[R2 / 2];
    const int k1 = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < R2 / 2; i += THREADS)
        s_tw[i] = tw_r2[i];
    const int n_all = p_pad / PG / NPG;      // pieces per (b, k1) row of Y (p_pad is a multiple of NPG pair groups)
    const int n_parts = gridDim.z, part = blockIdx.z;
    const int n_pieces = (n_all - part + n_parts - 1) / n_parts;      // pieces part, part + n_parts, ...
    double *pout = Pfull + ((int64_t(part) * gridDim.y + b) * R1 + k1) * R2 + tid;
    if (n_pieces <= 0) {
        if (!accumulate && tid < R2)
            *pout = 0.0;
        return;
    }
    // (pg_major: Y[block][pair group][k1][n2][pair], see msd_fft_cols400_fused_kernel: the NPG pair groups of a piece are
    // then pg_stride apart; element tid + 512 r of a piece is element tid + 512 (r % F) of its pair group r / F)
    const int64_t k1_stride = pg_major ? int64_t(R2) * PG : int64_t(n_all) * NPG * R2 * PG;
    const int64_t pg_stride = pg_major ? int64_t(R1) * R2 * PG : int64_t(R2) * PG;
    const int64_t g_stride = int64_t(n_parts) * NPG * pg_stride;
    const double2 *src = Y + int64_t(b) * n_all * NPG * (int64_t(R1) * R2 * PG) + int64_t(k1) * k1_stride +
                         int64_t(part) * NPG * pg_stride + tid;
    const int p = tid & 7, j = tid >> 3;
    double2 v[8];
    double acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        acc[r] = 0.0;
        v[r] = src[(r / F) * pg_stride + (r % F) * THREADS];
    }
    __syncthreads();   // twiddle table
    const int c = lane / T8, l = lane % T8;                 // stage 3: transform c of this wave, butterfly l
    for (int pc = 0; pc < n_pieces; ++pc) {
        // stage 1 on the loaded values: pair group g, butterfly j of pair p -> points F j + k
#pragma unroll
        for (int g = 0; g < NPG; ++g) {
            double2 *z = &zb[g * PG + p][F * j];
            if (F == 2) {
                z[0] = cadd(v[2 * g], v[2 * g + 1]);
                z[1] = csub(v[2 * g], v[2 * g + 1]);
            } else {
                double2 a[4] = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                dft4(a);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    z[k] = a[k];
            }
        }
        __syncthreads();
        {   // the next piece is in flight during stages 2 and 3 (the last iteration reloads its own: no branch)
            const int64_t off = int64_t(min(pc + 1, n_pieces - 1)) * g_stride;
#pragma unroll
            for (int r = 0; r < 8; ++r)
                v[r] = src[off + (r / F) * pg_stride + (r % F) * THREADS];
        }
        // wave w owns pair index w of every pair group of the piece: transforms g * PG + w.  They are not adjacent
        // rows of zb, so the batch stage takes its stride between them: PG rows
        stockham_stage_batch<R2, TPW, 8, F, false>(&zb[wave][0], PG * ZS, s_tw, lane, R2);
        {   // stage 3 (NS = R2 / 8): butterfly l reads points l + T8 r, output r is X[l + T8 r]
            const double2 *z = &zb[c * PG + wave][0];
            double2 u[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                u[r] = z[l + T8 * r];
                if (r)
                    u[r] = cmul(u[r], tw_at<R2>(s_tw, r * l));
            }
            dft8(u);
#pragma unroll
            for (int r = 0; r < 8; ++r)
                acc[r] = fma(u[r].x, u[r].x, fma(u[r].y, u[r].y, acc[r]));
        }
        __syncthreads();
    }
    // the NT transforms' sums, added in a fixed order: red[transform][k2], k2 = l + T8 r
    double *red = reinterpret_cast<double *>(&zb[0][0]);      // NT x R2 doubles = 32 KB
#pragma unroll
    for (int r = 0; r < 8; ++r)
        red[(c * PG + wave) * R2 + l + T8 * r] = acc[r];
    __syncthreads();
    if (tid < R2) {
        double total = 0.0;
        for (int t = 0; t < NT; ++t)
            total += red[t * R2 + tid];
        *pout = accumulate ? *pout + total : total;
    }
}

// P[b][k] += sum over the parts of (Pfull[part][b][k] + Pfull[part][b][N - k]) / 2 for the half
// spectrum k <= N/2, with Pfull stored as [k1][k2], k = k1 + R1 k2.
__global__ __launch_bounds__(256) void msd_power_fold_kernel(const double *__restrict__ Pfull,
                                                            int r1, int r2, int n_parts, int64_t nc,
                                                            double *__restrict__ P)
{
    const int64_t k = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (k >= nc)
        return;
    const int64_t n = int64_t(r1) * r2;
    const int64_t km = (n - k) % n;
    double s = 0.0;
    for (int part = 0; part < n_parts; ++part) {
        const double *pf = Pfull + (int64_t(part) * gridDim.y + b) * n;
        s += 0.5 * (pf[(k % r1) * r2 + k / r1] + pf[(km % r1) * r2 + km / r1]);
    }
    P[int64_t(b) * nc + k] += s;
}


// ------------------------------------------------------------------------------ single pass (N = 400, 800, 1 600)
// Blocks of at most 800 frames: the whole transform of a pair fits in LDS, so `Y` never exists — the positions are
// read ONCE from HBM and nothing else moves (the two-pass pipeline moves 5.1 x that; blocks of <= 200 frames went
// through rocFFT at ~17 x).  N = 400 R2 (R2 = 1, 2, 4), at most LIVE = 200 R2 live frames.  Decimation in frequency
// over the zero-padded input: X[R2 j + r] = DFT_400(y_r)[j] with
//     y_r[n] = (sum_{q < R2 / 2} x[n + 400 q] (-i)^(q r)) W_N^(r n),   n < 400
// (q >= R2 / 2: the padding), so a pair costs R2 in-place 400-point transforms (the dif stages 10, 10, 4 of pass A)
// by ONE wave, and |X|^2 is added up in registers over all the pair groups a block walks — slot lane + 64 i of
// sub-transform r; the bit-reversal is undone once, when the block writes its sums.
//
// grid (splits, blocks of the trajectory): block (s, b) takes pair groups s, s + splits, ... of trajectory block b.
// Per pair group: 512 threads load the LIVE rows of 128 B (8 pairs) into registers (the NEXT pair group's rows are
// issued before the current one is transformed), stage them pair-major in LDS, every wave adds up the per-frame
// sums (x^2 and the coordinate sums: registers, one record per frame at the end), then wave w transforms pair w.
//   R2 = 1: in place on the staged rows (first stage skips the zero half);
//   R2 = 2: the wave keeps its pair's rows in registers, transforms in place (r = 0), re-stages x W_800^n (r = 1);
//   R2 = 4: y_r is formed from the staged rows into a wave-private work buffer, four times.
// Pfull[split][b][k1][k2] (k = k1 + 400 k2) as the row kernels leave it: msd_power_fold_kernel folds the splits.
constexpr int single_live(int r2) { return 200 * r2; }

template <int R2, typename In = double>       // In: double, or float (widened at the staging: see msd_fft_cols400_fused_kernel)
__global__ __launch_bounds__(THREADS, R2 == 4 ? 1 : 2) void msd_fft_single400_kernel(
    const In *__restrict__ pos, int64_t n_total, int64_t first, int64_t n_elem, int64_t t_block,
    int zero_dims, int p_pad, const double2 *__restrict__ tw_r1, const double2 *__restrict__ twN,
    double *__restrict__ Pfull, double2 *__restrict__ part, int head)
{
    static_assert(R2 == 1 || R2 == 2 || R2 == 4, "single-pass lengths");
    constexpr int R1 = 400, LIVE = 200 * R2;
    constexpr int LOADS = (LIVE + 63) / 64;          // row rounds of 64 rows (a lane loads a PAIR: 16 bytes)
    constexpr int SL = (R1 + 63) / 64;               // 7 slot rounds of a 400-point transform
    constexpr int XS = (R2 == 4 ? LIVE : R1) + 1;    // staged row of a pair (+ 1: the 8 pairs of a frame on different banks)
    constexpr int PASSES = (LIVE + THREADS - 1) / THREADS;   // per-frame sums: a thread owns frames tid + 512 k
    __shared__ double2 zb[PG][XS];
    __shared__ double2 zw[R2 == 4 ? PG : 1][R2 == 4 ? R1 + 1 : 1];
    __shared__ double2 s_h[R1];                      // exp(-2 pi i m / 400), m < 400
    const int b = blockIdx.y, n_pg = p_pad / PG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < R1; i += THREADS)
        s_h[i] = tw_r1[i];
    const int dshift = (3 - head % 3) % 3;           // dimension of chunk coordinate e: (e + dshift) % 3
    const int lp = tid & 7, s = 2 * lp, row0 = tid >> 3;
    const In *base = pos + (int64_t(b) * t_block * n_total + first) * 3 - head;
    const int64_t row_stride = n_total * 3;
    const int t_lim = int(min<int64_t>(t_block, LIVE));
    struct alignas(sizeof(In)) Pair { In x, y; };
    Pair x[LOADS];
    // dead rows re-read the block's last live row and coordinates past the chunk the chunk's first pair (valid
    // memory; zeros are staged for them); a pair whose second coordinate lies past the chunk is read one coordinate
    // earlier (its first one arrives in .y).  The row offsets are formed per load from an opaque copy of row0: hoisted
    // out of the loop they were 26 registers for R2 = 4, spilled, and reloaded from scratch in front of every load.
#define MDX_SINGLE_LOAD(PG_, I0, I1)                                                                 \
    {                                                                                                \
        const int64_t e0_ = int64_t(16) * (PG_) + s;                                                 \
        const In *c_ = e0_ < n_elem ? base + e0_ - (e0_ + 1 < n_elem ? 0 : 1) : base;                \
        int r0_ = row0;                                                                              \
        asm volatile("" : "+v"(r0_));                                                                \
        _Pragma("unroll") for (int i = (I0); i < (I1); ++i)                                          \
        {                                                                                            \
            const int row_ = min(r0_ + 64 * i, t_lim - 1);                                           \
            x[i] = *reinterpret_cast<const Pair *>(c_ + int64_t(row_) * row_stride);                 \
        }                                                                                            \
    }
    // rounds [0, EARLY) of the next pair group are issued early — R2 < 4: all of them, right after the staging, so that
    // they are in flight while the current pair group is transformed; R2 = 4: none (the last sub-transform alone needs
    // ~250 registers beside |X|^2 and the per-frame sums: any row kept beside it spilled — 7 rounds 28, 4 rounds 12, 3
    // rounds 5 registers — and a scratch reload waits for every load issued before it), its rows are loaded at the
    // top of their own iteration
    constexpr int EARLY = R2 == 4 ? 0 : LOADS;
    double P[R2][8];        // |X|^2 sums: P[r][4 p + k] <-> slot 4 (lane + 64 p) + k of sub-transform r (dif_last4_power_400)
#pragma unroll
    for (int r = 0; r < R2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i)
            P[r][i] = 0.0;
    double sd[PASSES], sx[PASSES], sy[PASSES], sz[PASSES];
#pragma unroll
    for (int k = 0; k < PASSES; ++k)
        sd[k] = sx[k] = sy[k] = sz[k] = 0.0;
    // W_N^n = W_400^(n / R2) W_N^(n % R2)
    const double2 wn1 = twN[R2 > 1 ? 1 : 0], wn2 = twN[R2 > 2 ? 2 : 0], wn3 = twN[R2 > 2 ? 3 : 0];

    int pg = blockIdx.x;
    if (pg < n_pg)
        MDX_SINGLE_LOAD(pg, 0, EARLY)
    __syncthreads();
    for (; pg < n_pg; pg += gridDim.x) {
        if (EARLY < LOADS)
            MDX_SINGLE_LOAD(pg, EARLY, LOADS)
        {   // stage the rows pair-major
            const int64_t e0 = int64_t(16) * pg + s;
            const bool shifted = !(e0 + 1 < n_elem);
            const bool ok0 = e0 < n_elem && e0 >= head && !((zero_dims >> int((e0 + dshift) % 3)) & 1);
            const bool ok1 = e0 + 1 < n_elem && e0 + 1 >= head && !((zero_dims >> int((e0 + 1 + dshift) % 3)) & 1);
#pragma unroll
            for (int i = 0; i < LOADS; ++i) {
                const int row = row0 + 64 * i;
                if (LIVE % 64 == 0 || row < LIVE) {
                    const bool live = row < t_lim;
                    const double v0 = double(shifted ? x[i].y : x[i].x);
                    zb[lp][row] = make_double2(ok0 && live ? v0 : 0.0, ok1 && live ? double(x[i].y) : 0.0);
                }
            }
        }
        // the next pair group's rows are in flight while this one is transformed
        if (R2 < 4 && pg + int(gridDim.x) < n_pg)
            MDX_SINGLE_LOAD(pg + int(gridDim.x), 0, LOADS)
        __syncthreads();
        {   // per-frame sums of the pair group's 16 coordinates: thread t owns frames t, t + 512, ... (consecutive
            // threads read consecutive rows of one pair: no bank conflict); coordinate 2 j (+ 1) of pair group pg has
            // dimension (pg + 2 j (+ 1) + dshift) % 3 (16 = 1 mod 3)
            const int rot = int((int64_t(pg) + dshift) % 3);
#pragma unroll
            for (int k = 0; k < PASSES; ++k) {
                const int row = THREADS * k + tid;
                if (LIVE % THREADS == 0 || row < LIVE) {
                    double c0 = 0.0, c1 = 0.0, c2 = 0.0, d = 0.0;
#pragma unroll
                    for (int j = 0; j < PG; ++j) {
                        const double2 v = zb[j][row];
                        d = fma(v.x, v.x, fma(v.y, v.y, d));
                        if ((2 * j) % 3 == 0) c0 += v.x; else if ((2 * j) % 3 == 1) c1 += v.x; else c2 += v.x;
                        if ((2 * j + 1) % 3 == 0) c0 += v.y; else if ((2 * j + 1) % 3 == 1) c1 += v.y; else c2 += v.y;
                    }
                    sd[k] += d;
                    sx[k] += rot == 0 ? c0 : rot == 1 ? c2 : c1;
                    sy[k] += rot == 0 ? c1 : rot == 1 ? c0 : c2;
                    sz[k] += rot == 0 ? c2 : rot == 1 ? c1 : c0;
                }
            }
        }
        __syncthreads();   // the sums read every pair's rows; the transforms below overwrite them
        double2 *z = zb[wave];
        if (R2 == 1) {
            dif_stage<R1, 400, 10, true>(z, s_h, lane);
            dif_stage<R1, 40, 10, false>(z, s_h, lane);
            dif_last4_power_400(z, lane, P[0]);
        } else if (R2 == 2) {
            double2 xr[SL];
#pragma unroll
            for (int i = 0; i < SL; ++i)
                xr[i] = (i < SL - 1 || lane + 64 * i < R1) ? z[lane + 64 * i] : make_double2(0.0, 0.0);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                if (r == 1) {
                    // x W_800^n, n = lane + 64 i: W_800^lane (one table entry) times powers of W_800^64 = W_400^32
                    const double2 wl = s_h[lane >> 1];
                    double2 w = (lane & 1) ? cmul(wl, wn1) : wl;
                    const double2 ws = s_h[32];
#pragma unroll
                    for (int i = 0; i < SL; ++i) {
                        if (i < SL - 1 || lane + 64 * i < R1)
                            z[lane + 64 * i] = cmul(xr[i], w);
                        if (i + 1 < SL)
                            w = cmul(w, ws);
                    }
                    wave_lds_fence();
                }
                dif_stage<R1, 400, 10, false>(z, s_h, lane);
                dif_stage<R1, 40, 10, false>(z, s_h, lane);
                dif_last4_power_400(z, lane, P[r]);
            }
        } else {
            double2 *w = zw[R2 == 4 ? wave : 0];
#pragma unroll
            for (int r = 0; r < R2; ++r) {
                if (R2 == 4 && r == R2 - 1 && pg + int(gridDim.x) < n_pg)
                    MDX_SINGLE_LOAD(pg + int(gridDim.x), 0, EARLY)
                // (not unrolled: seven slots' operands and twiddles in flight at once were ~110 registers on top
                // of the ~140 the kernel keeps — rows in flight, |X|^2 and per-frame sums — and spilled)
#pragma unroll 1
                for (int i = 0; i < SL; ++i)
                    if (i < SL - 1 || lane + 64 * i < R1) {
                        const int n = lane + 64 * i;
                        const double2 a = z[n], c = z[n + R1];
                        // a + (-i)^r c
                        const double2 t = r == 0   ? cadd(a, c)
                                          : r == 1 ? cadd(a, mul_mi(c))
                                          : r == 2 ? csub(a, c)
                                                   : csub(a, mul_mi(c));
                        if (r == 0) {
                            w[n] = t;
                        } else {
                            const int m = n & 3;
                            const double2 w0 = s_h[n >> 2];
                            const double2 w1 = m == 0 ? w0 : cmul(w0, m == 1 ? wn1 : m == 2 ? wn2 : wn3);
                            const double2 w2 = cmul(w1, w1);
                            w[n] = cmul(t, r == 1 ? w1 : r == 2 ? w2 : cmul(w2, w1));
                        }
                    }
                wave_lds_fence();
                __builtin_amdgcn_sched_barrier(0);
                dif_stage<R1, 400, 10, false>(w, s_h, lane);
                dif_stage<R1, 40, 10, false>(w, s_h, lane);
                dif_last4_power_400(w, lane, P[r]);
                __builtin_amdgcn_sched_barrier(0);
                wave_lds_fence();          // the next sub-transform's work buffer is this one
            }
        }
        __syncthreads();   // the next staging overwrites the rows
    }
#undef MDX_SINGLE_LOAD
    // per-frame records of this block: [split][b][frame][(x^2, x), (y, z)]
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
        const int row = THREADS * k + tid;
        if (LIVE % THREADS == 0 || row < LIVE) {
            double2 *o = part + ((int64_t(blockIdx.x) * gridDim.y + b) * LIVE + row) * 2;
            o[0] = make_double2(sd[k], sx[k]);
            o[1] = make_double2(sy[k], sz[k]);
        }
    }
    // |X|^2 over the block's pairs: the eight waves' sums meet in LDS (the staging rows are free now)
    double *red = reinterpret_cast<double *>(&zb[0][0]);      // [wave][r][slot]: 8 x R2 x 400 doubles <= sizeof(zb)
    static_assert(sizeof(double) * PG * R2 * R1 <= sizeof(double2) * PG * XS, "reduction buffer");
#pragma unroll
    for (int r = 0; r < R2; ++r)
#pragma unroll
        for (int pq = 0; pq < 2; ++pq) {
            if (pq == 0 || lane + 64 * pq < 100) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    red[(wave * R2 + r) * R1 + 4 * (lane + 64 * pq) + k] = P[r][4 * pq + k];
            }
        }
    __syncthreads();
    double *pf = Pfull + (int64_t(blockIdx.x) * gridDim.y + b) * (int64_t(R1) * R2);
    for (int idx = tid; idx < R2 * R1; idx += THREADS) {
        const int r = idx / R1, slot = idx - r * R1;
        double total = 0.0;
#pragma unroll
        for (int wv = 0; wv < PG; ++wv)
            total += red[(wv * R2 + r) * R1 + slot];
        // slot 40 ka + 4 kb + kc holds X_r[j], j = ka + 10 kb + 100 kc (dif_slot_400); k = R2 j + r = k1 + 400 k2
        const int ka = slot / 40, rem = slot - 40 * ka, kb = rem >> 2, kc = rem & 3;
        const int k = R2 * (ka + 10 * kb + 100 * kc) + r;
        pf[(k % R1) * R2 + k / R1] = total;
    }
}


// ------------------------------------------------------------------------------ host side

struct Shape {
    int r1 = 0, r2 = 0;   // 0: no own transform for this length
    int64_t n() const { return int64_t(r1) * r2; }
};

inline Shape shape_for(int64_t n_fft)
{
    Shape s;
    if (n_fft == (int64_t(1) << 13))
        s.r1 = 64, s.r2 = 128;
    else if (n_fft == (int64_t(1) << 14))
        s.r1 = 64, s.r2 = 256;
    else if (n_fft == (int64_t(1) << 15))
        s.r1 = 64, s.r2 = 512;
    else if (n_fft == (int64_t(1) << 16))
        s.r1 = 64, s.r2 = 1024;
    else if (n_fft == 400)
        s.r1 = 400, s.r2 = 1;
    else if (n_fft == 800)
        s.r1 = 400, s.r2 = 2;
    else if (n_fft == 1600)
        s.r1 = 400, s.r2 = 4;
    else if (n_fft == 3200)
        s.r1 = 400, s.r2 = 8;
    else if (n_fft == 6400)
        s.r1 = 400, s.r2 = 16;
    else if (n_fft == 12800)
        s.r1 = 400, s.r2 = 32;
    else if (n_fft == 25600)
        s.r1 = 400, s.r2 = 64;
    else if (n_fft == 51200)
        s.r1 = 400, s.r2 = 128;
    else if (n_fft == 102400)
        s.r1 = 400, s.r2 = 256;
    else if (n_fft == 204800)
        s.r1 = 400, s.r2 = 512;
    else if (n_fft == 409600)
        s.r1 = 400, s.r2 = 1024;
    else if (n_fft == (int64_t(1) << 18))
        s.r1 = 512, s.r2 = 512;
    else if (n_fft == (int64_t(1) << 19))
        s.r1 = 1024, s.r2 = 512;
    else if (n_fft == (int64_t(1) << 20))
        s.r1 = 1024, s.r2 = 1024;
    return s;
}

// entries of the first-factor twiddle table the kernels expect
inline int tw_r1_len(int r1) { return r1 == 400 ? 400 : r1 / 2; }

// Parts of pass B (blockIdx.z): R1 x blocks x parts >= 2048 blocks = four rounds of the 512 block
// slots of the chip (with fewer, a block count that is not a multiple of 512 costs a whole extra
// round: 800 blocks took as long as 1024); Pfull holds one copy per part.
constexpr int ROWS_PARTS_MAX = 32;
constexpr int BLOCK_SLOTS = 512;   // two 512-thread blocks per CU (their LDS) x 256 CUs

// A factor f in [lo, hi] for a grid of base * f equal blocks: the smallest one that gives at least
// min_rounds (four) rounds of the chip's block slots and wastes under 3 % of the last round (a grid of 3.7 rounds
// takes as long as one of 4; with fewer than ~4 rounds the ramp-up and the tail weigh too much), else
// the one that wastes least.
inline int slots_split(int64_t base, int lo, int hi, int min_rounds = 4)
{
    int best = lo;
    double best_eff = -1.0;
    for (int f = lo; f <= hi; ++f) {
        const int64_t blocks = base * f;
        const int64_t rounds = (blocks + BLOCK_SLOTS - 1) / BLOCK_SLOTS;
        const double eff = double(blocks) / double(rounds * BLOCK_SLOTS);
        if (rounds >= min_rounds && eff >= 0.97)
            return f;
        if (eff * (rounds >= min_rounds ? 1.0 : 0.9) > best_eff) {
            best_eff = eff * (rounds >= min_rounds ? 1.0 : 0.9);
            best = f;
        }
    }
    return best;
}

inline int rows_parts(const Shape &sh, int n_blocks)
{
    const int64_t base = int64_t(sh.r1) * n_blocks;
    if (base >= 8 * BLOCK_SLOTS)
        return 1;
    return slots_split(base, 1, ROWS_PARTS_MAX);
}

// Pass B of the 128- and 256-point rows takes 64 KB pieces of 4 / 2 whole pair groups: the padded pair count of a
// launch is a multiple of this many pair groups (the padding pairs are zeros and add nothing to the spectrum).
inline int pair_group_multiple(const Shape &sh) { return (sh.r2 == 128 || sh.r2 == 256) ? 512 / sh.r2 : 1; }

// Shapes whose pass A carries the per-frame sums (x^2 and the coordinate sums of every frame) itself:
// the caller then skips its own sums kernel and hands `part`, `traj`, `dsq` to launch().
inline bool fuses_sums(const Shape &sh) { return sh.r1 == 400 || sh.r1 == 64; }
// pass A kernels that can enter a chunk a few coordinates early to make its 128-byte pieces whole cache lines
inline bool aligns_head(const Shape &sh) { return sh.r1 == 400; }   // (16-point factors: trajectories too short to matter)
// pass A kernels that read float32 positions where they lie (launch(..., pos32)): the 400-point family with two passes
// (the single-pass kernel of the 400-point first factor included), the 64-point family: every shape with fused sums
inline bool cols_read_f32(const Shape &sh) { return sh.r1 == 400 || sh.r1 == 64; }
inline int fused_super_groups(int p_pad) { return (p_pad / PG + SG - 1) / SG; }
// bytes of the partial-sum records of one launch: [super group][block][R2][200 rows][4 doubles]
inline size_t fused_part_bytes(const Shape &sh, int p_pad, int n_blocks)
{
    // records per block: 400-point factor r2 x 200 rows; short factors (r2 / nc groups) x 256 frames = r2 x r1 / 2
    const size_t per_b = sh.r1 == 400 ? size_t(sh.r2) * SUMS_ROWS : size_t(sh.r2) * (sh.r1 / 2);
    return size_t(fused_super_groups(p_pad)) * n_blocks * per_b * 32;
}

// splits of the single-pass grid: >= 2 048 blocks where the pair groups allow it
inline int single_splits(int n_blocks) { return std::max(1, std::min(64, (2048 + n_blocks - 1) / n_blocks)); }
inline bool single_pass(const Shape &sh) { return sh.r1 == 400 && sh.r2 <= 4; }
inline size_t single_part_bytes(const Shape &sh, int n_blocks)
{
    return size_t(single_splits(n_blocks)) * n_blocks * single_live(sh.r2) * 32;
}

// one launch of the single-pass pipeline for a chunk; returns the number of Pfull parts written
inline int launch_single(const Shape &sh, hipStream_t stream, const double *pos, int64_t n_total, int64_t first,
                         int64_t n_elem, int64_t t_block, int n_blocks, int zero_dims, int p_pad,
                         const double2 *tw_r1, const double2 *twN, double *Pfull, double2 *part, double *traj,
                         double *dsq, int head, const float *pos32 = nullptr)
{
    const int splits = std::min(single_splits(n_blocks), p_pad / PG);
    const dim3 grid((unsigned)splits, (unsigned)n_blocks);
#define MDX_MSDFFT_SINGLE(R2_)                                                                                   \
    if (pos32)                                                                                                   \
        hipLaunchKernelGGL((msd_fft_single400_kernel<R2_, float>), grid, dim3(THREADS), 0, stream, pos32, n_total, first, \
                           n_elem, t_block, zero_dims, p_pad, tw_r1, twN, Pfull, part, head);                    \
    else                                                                                                         \
        hipLaunchKernelGGL((msd_fft_single400_kernel<R2_>), grid, dim3(THREADS), 0, stream, pos, n_total, first, n_elem, \
                           t_block, zero_dims, p_pad, tw_r1, twN, Pfull, part, head)
    if (sh.r2 == 1) {
        MDX_MSDFFT_SINGLE(1);
    } else if (sh.r2 == 2) {
        MDX_MSDFFT_SINGLE(2);
    } else {
        MDX_MSDFFT_SINGLE(4);
    }
#undef MDX_MSDFFT_SINGLE
    const int live = single_live(sh.r2);
    hipLaunchKernelGGL(msd_partials_reduce_kernel, dim3((unsigned)((live + 255) / 256), (unsigned)n_blocks), dim3(256),
                       0, stream, part, splits, 1, 1, live, t_block, traj, dsq);
    return splits;
}

// pass B of shape R1 x R2: the register-staged kernel for 512-point rows, the general one for 1024
template <int R1, int R2>
inline void launch_rows(dim3 gb, hipStream_t stream, const double2 *Y, int p_pad, const double2 *tw_r2,
                        double *Pfull, int accumulate, int pg_major = 0)
{
    if constexpr (R2 == 512)
        hipLaunchKernelGGL((msd_fft_rows512_power_kernel<R1>), gb, dim3(THREADS), 0, stream, Y, p_pad, tw_r2,
                           Pfull, accumulate, pg_major);
    else if constexpr (R2 == 128 || R2 == 256)
        hipLaunchKernelGGL((msd_fft_rows_mid_power_kernel<R1, R2>), gb, dim3(THREADS), 0, stream, Y, p_pad, tw_r2,
                           Pfull, accumulate, pg_major);
    else if constexpr (R2 == 1024)
        hipLaunchKernelGGL((msd_fft_rows1024_power_kernel<R1>), gb, dim3(THREADS), 0, stream, Y, p_pad, tw_r2,
                           Pfull, accumulate, pg_major);
    else
        hipLaunchKernelGGL((msd_fft_rows_power_kernel<R1, R2>), gb, dim3(THREADS), 0, stream, Y, p_pad, tw_r2,
                           Pfull, accumulate, pg_major);
}

// tw_r1 / tw_r2: half tables exp(-2 pi i m / R), m < R / 2; twN: exp(-2 pi i m / N), m < R2
// One batch of coordinates -> Pfull; accumulate != 0 adds to what earlier batches left there.
inline void launch(const Shape &sh, hipStream_t stream, const double *pos, int64_t n_total,
                   int64_t first, int64_t n_elem, int64_t t_block, int n_blocks, int zero_dims,
                   int p_pad, const double2 *tw_r1, const double2 *tw_r2, const double2 *twN,
                   double2 *Y, double *Pfull, int accumulate, double2 *part = nullptr,
                   double *traj = nullptr, double *dsq = nullptr, int head = 0, const float *pos32 = nullptr)
{
    // pos32: the positions as float32 (400-point first factor with fused sums only: cols400_reads_f32); `pos` is unused then
    // >= ~1024 blocks of pass A where the batch allows it
    int split = 4;
    while (split < sh.r2 / 2 && int64_t(p_pad / PG) * split * n_blocks < 1024)
        split *= 2;
    const dim3 ga((unsigned)(p_pad / PG), (unsigned)split, (unsigned)n_blocks);
    const dim3 gb((unsigned)sh.r1, (unsigned)n_blocks, (unsigned)rows_parts(sh, n_blocks));
#define MDX_MSDFFT_LAUNCH(A, B)                                                                    \
    hipLaunchKernelGGL((msd_fft_cols_kernel<A, B>), ga, dim3(THREADS), 0, stream, pos, n_total, first, \
                       n_elem, t_block, zero_dims, p_pad, tw_r1, twN, Y, 1);                           \
    launch_rows<A, B>(gb, stream, Y, p_pad, tw_r2, Pfull, accumulate, 1)
    if (sh.r1 == 64 && part) {
        // per-frame sums fused into pass A (short first factors)
        const int n_sg = fused_super_groups(p_pad);
        const int nc = 512 / sh.r1, ng = sh.r2 / nc;
        const int fsplit = std::min(ng, slots_split(int64_t(n_sg) * n_blocks, 1, 64));
        const dim3 gf((unsigned)n_sg, (unsigned)fsplit, (unsigned)n_blocks);
#define MDX_MSDFFT_SMALL_FUSED(A, B)                                                                              \
    if (pos32)                                                                                                    \
        hipLaunchKernelGGL((msd_fft_cols_small_fused_kernel<A, B, float>), gf, dim3(THREADS), 0, stream, pos32, n_total, \
                           first, n_elem, t_block, zero_dims, p_pad, tw_r1, twN, Y, part, pgm);                   \
    else                                                                                                          \
        hipLaunchKernelGGL((msd_fft_cols_small_fused_kernel<A, B>), gf, dim3(THREADS), 0, stream, pos, n_total, first, \
                           n_elem, t_block, zero_dims, p_pad, tw_r1, twN, Y, part, pgm);                         \
    hipLaunchKernelGGL(msd_partials_reduce_kernel, dim3((unsigned)((int64_t(B) * (A / 2) + 255) / 256),          \
                       (unsigned)n_blocks), dim3(256), 0, stream, part, n_sg, B, 512 / A, A / 2, t_block, traj, dsq); \
    launch_rows<A, B>(gb, stream, Y, p_pad, tw_r2, Pfull, accumulate, pgm)
        // (the pair-group-major layout of the 400-point family gains nothing here — this pass A already stores runs of
        // NC x 128 B = 1 KB: C4 size in 25 / 13 / 7 blocks 25.6 / 27.1 / 28.8 ms k1-major, 26.1 / 27.4 / 28.6 pair-group-major)
        const int pgm = 0;
        if (sh.r2 == 128) {
            MDX_MSDFFT_SMALL_FUSED(64, 128);
        } else if (sh.r2 == 256) {
            MDX_MSDFFT_SMALL_FUSED(64, 256);
        } else if (sh.r2 == 512) {
            MDX_MSDFFT_SMALL_FUSED(64, 512);
        } else {
            MDX_MSDFFT_SMALL_FUSED(64, 1024);
        }
#undef MDX_MSDFFT_SMALL_FUSED
    } else if (sh.r1 == 400 && part) {
        // per-frame sums fused into pass A: super groups of SG pair groups, >= ~1024 blocks
        const int n_sg = fused_super_groups(p_pad);
        // (twelve rounds of the chip's block slots: C4 with one block 27.1 -> 26.2 ms of kernels per step against the
        // four rounds of round 2 — blocks in more different phases share a CU, and the tail is a twelfth)
        const int fsplit = slots_split(int64_t(n_sg) * n_blocks, std::min(8, sh.r2), std::min(64, sh.r2), 12);
        // Y[block][pair group][k1][n2][pair] where a (k1, pair group) run is at least 4 KB (measured: 1 and 2 KB runs cost pass B
        // 0.3 ms more than the layout saves pass A) and pass B takes the layout
        const int pg_major = sh.r2 >= 32 ? 1 : 0;
#define MDX_MSDFFT_COLS400(R2_)                                                                                       \
    if (pos32)                                                                                                       \
        hipLaunchKernelGGL((msd_fft_cols400_fused_kernel<R2_, float>), dim3((unsigned)n_sg, (unsigned)fsplit,        \
                           (unsigned)n_blocks), dim3(THREADS), 0, stream, pos32, n_total, first, n_elem, t_block,    \
                           zero_dims, p_pad, tw_r1, twN, Y, part, head, pg_major);                                   \
    else                                                                                                             \
        hipLaunchKernelGGL((msd_fft_cols400_fused_kernel<R2_>), dim3((unsigned)n_sg, (unsigned)fsplit, (unsigned)n_blocks), \
                           dim3(THREADS), 0, stream, pos, n_total, first, n_elem, t_block, zero_dims, p_pad, tw_r1, twN, Y, \
                           part, head, pg_major)
        if (sh.r2 == 2) {
            MDX_MSDFFT_COLS400(2);
        } else if (sh.r2 == 4) {
            MDX_MSDFFT_COLS400(4);
        } else if (sh.r2 == 8) {
            MDX_MSDFFT_COLS400(8);
        } else if (sh.r2 == 16) {
            MDX_MSDFFT_COLS400(16);
        } else if (sh.r2 == 32) {
            MDX_MSDFFT_COLS400(32);
        } else if (sh.r2 == 64) {
            MDX_MSDFFT_COLS400(64);
        } else if (sh.r2 == 128) {
            MDX_MSDFFT_COLS400(128);
        } else if (sh.r2 == 256) {
            MDX_MSDFFT_COLS400(256);
        } else if (sh.r2 == 512) {
            MDX_MSDFFT_COLS400(512);
        } else {
            MDX_MSDFFT_COLS400(1024);     // 80.5 KB of LDS: still two blocks per CU
        }
#undef MDX_MSDFFT_COLS400
        hipLaunchKernelGGL(msd_partials_reduce_kernel,
                           dim3((unsigned)((int64_t(sh.r2) * SUMS_ROWS + 255) / 256), (unsigned)n_blocks), dim3(256), 0,
                           stream, part, n_sg, sh.r2, 1, SUMS_ROWS, t_block, traj, dsq);
        if (sh.r2 == 2)
            hipLaunchKernelGGL((msd_fft_rows_tiny_power_kernel<400, 2>), gb, dim3(THREADS), 0, stream, Y, p_pad, Pfull,
                               accumulate);
        else if (sh.r2 == 4)
            hipLaunchKernelGGL((msd_fft_rows_tiny_power_kernel<400, 4>), gb, dim3(THREADS), 0, stream, Y, p_pad, Pfull,
                               accumulate);
        else if (sh.r2 == 8)
            hipLaunchKernelGGL((msd_fft_rows_short_power_kernel<400, 8>), gb, dim3(THREADS), 0, stream, Y, p_pad, tw_r2,
                               Pfull, accumulate, pg_major);
        else if (sh.r2 == 16)
            hipLaunchKernelGGL((msd_fft_rows_short_power_kernel<400, 16>), gb, dim3(THREADS), 0, stream, Y, p_pad, tw_r2,
                               Pfull, accumulate, pg_major);
        else if (sh.r2 == 32)
            hipLaunchKernelGGL((msd_fft_rows_short_power_kernel<400, 32>), gb, dim3(THREADS), 0, stream, Y, p_pad, tw_r2,
                               Pfull, accumulate, pg_major);
        else if (sh.r2 == 64)
            hipLaunchKernelGGL((msd_fft_rows_short_power_kernel<400, 64>), gb, dim3(THREADS), 0, stream, Y, p_pad, tw_r2,
                               Pfull, accumulate, pg_major);
        else if (sh.r2 == 128)
            hipLaunchKernelGGL((msd_fft_rows_mid_power_kernel<400, 128>), gb, dim3(THREADS), 0, stream, Y, p_pad, tw_r2,
                               Pfull, accumulate, pg_major);
        else if (sh.r2 == 256)
            hipLaunchKernelGGL((msd_fft_rows_mid_power_kernel<400, 256>), gb, dim3(THREADS), 0, stream, Y, p_pad, tw_r2,
                               Pfull, accumulate, pg_major);
        else if (sh.r2 == 512)
            hipLaunchKernelGGL((msd_fft_rows512_power_kernel<400>), gb, dim3(THREADS), 0, stream, Y, p_pad,
                               tw_r2, Pfull, accumulate, pg_major);
        else
            hipLaunchKernelGGL((msd_fft_rows1024_power_kernel<400>), gb, dim3(THREADS), 0, stream, Y, p_pad,
                               tw_r2, Pfull, accumulate, pg_major);
    } else if (sh.r1 == 512 && sh.r2 == 512) {
        MDX_MSDFFT_LAUNCH(512, 512);
    } else if (sh.r1 == 1024 && sh.r2 == 512) {
        MDX_MSDFFT_LAUNCH(1024, 512);
    } else {
        MDX_MSDFFT_LAUNCH(1024, 1024);
    }
#undef MDX_MSDFFT_LAUNCH
}

inline void launch_fold(const Shape &sh, hipStream_t stream, const double *Pfull, int n_blocks,
                        int64_t nc, double *P)
{
    const int n_parts = rows_parts(sh, n_blocks);
    hipLaunchKernelGGL(msd_power_fold_kernel, dim3((unsigned)((nc + 255) / 256), (unsigned)n_blocks),
                       dim3(256), 0, stream, Pfull, sh.r1, sh.r2, n_parts, nc, P);
}

}  // namespace msdfft
