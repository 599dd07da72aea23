// mdx_msd.hip — FFT time correlation / MSD on gfx950 (MI355X) with rocFFT.
//
// Carries reference src/mdhelper/algorithm/correlation.py:
//   correlation_fft :17-226   f = rfft(x, n=N_fft); corr = irfft(f f*)[:N_t]
//                             with N_fft = 2 * next_fast_len(N_t, real=True)
//   msd_fft         :461-668  MSD_m = S_m - 2 A_m, S_m by the cumsum recurrence
// as Onsager._conclude calls them (src/mdhelper/analysis/transport.py:1016-1059):
// per-particle self MSDs summed over the particles of a group (:1036-1039) and the
// summed trajectories of each group for the collective / cross terms (:1034, :1044-1052).
//
// Engine (mdx_msd_*): HBM-bound.  The reference materialises a spectrum, a product
// and an inverse per particle (3 x 48 GB at 10k particles x 100k frames); by linearity
//     sum_p irfft(|F_p|^2) = irfft(sum_p |F_p|^2)
// so the engine keeps ONE power spectrum per (group, block), accumulated over
// particles and xyz, and does one inverse transform per (group, block) at the end.
// Per pushed chunk of particles:
//   1. msd_gather_kernel   [T][N][3] -> zero-padded series [b][a][k][N_fft]  (LDS transpose)
//   2. msd_sums_kernel     D_t = sum_{a,k} x^2  and  sum_a r_a(t)          (fixed order)
//   3. rocFFT R2C, batch = 3 * particles * blocks, double precision
//   4. msd_power_kernel    P[b][f] += sum_{a,k} |F|^2                      (fixed order)
// Everything is fp64: S_m - 2 A_m cancels catastrophically at small lags.
#include "mdx_common.hpp"
#include "mdx_internal.hpp"
#include "mdx_traj.hpp"
#include "mdx_msd_fft.hpp"

#include <rocfft/rocfft.h>

#include <map>
#include <mutex>
#include <tuple>

using namespace mdx;

namespace {

#define MDX_FFT(expr)                                                             \
    do {                                                                          \
        rocfft_status _s = (expr);                                                \
        if (_s != rocfft_status_success)                                          \
            return fail(MDX_ERR_ROCFFT, "%s failed with rocfft_status %d", #expr, (int)_s); \
    } while (0)

int rocfft_ready()
{
    static std::once_flag once;
    static rocfft_status status = rocfft_status_success;
    std::call_once(once, [] { status = rocfft_setup(); });
    if (status != rocfft_status_success)
        return fail(MDX_ERR_ROCFFT, "rocfft_setup failed with rocfft_status %d", (int)status);
    return MDX_OK;
}

// scipy.fft.next_fast_len(n, real=True): smallest 2^a 3^b 5^c >= n
int64_t next_fast_len_real(int64_t n)
{
    if (n <= 6)
        return std::max<int64_t>(n, 1);
    int64_t best = INT64_MAX;
    for (int64_t p5 = 1; p5 < 2 * n; p5 *= 5)
        for (int64_t p35 = p5; p35 < 2 * n; p35 *= 3) {
            int64_t v = p35;
            while (v < n)
                v *= 2;
            best = std::min(best, v);
        }
    return best;
}

struct FftPlan {
    rocfft_plan plan = nullptr;
    size_t work_bytes = 0;
};

// Plans are kept for the life of the process, per (device, length, direction, batch): creating the plans of a
// 204 800-point transform costs ~10 - 25 ms each time (seconds the first time, while rocFFT compiles its
// kernels) — per analysis object, when every Onsager(...).run() and every msd_fft() call made its own.  A plan
// is not tied to a stream (the execution info is), so handles on different streams share it; rocFFT plans
// may be executed concurrently.
struct PlanStore {
    std::mutex m;
    std::map<std::tuple<int, int64_t, int, int64_t>, FftPlan> plans;
};
PlanStore &plan_store()
{
    static PlanStore *p = new PlanStore();   // never destroyed: rocFFT may be gone at exit
    return *p;
}

struct FftCache {
    rocfft_execution_info info = nullptr;
    DeviceBuffer work;
    int64_t n_fft = 0;

    int get(int inverse, int64_t batch, FftPlan *out)
    {
        int dev = 0;
        MDX_HIP(hipGetDevice(&dev));
        PlanStore &store = plan_store();
        std::lock_guard<std::mutex> lk(store.m);
        auto key = std::make_tuple(dev, n_fft, inverse, batch);
        auto it = store.plans.find(key);
        if (it == store.plans.end()) {
            FftPlan p;
            size_t len = (size_t)n_fft;
            rocfft_plan_description desc = nullptr;
            MDX_FFT(rocfft_plan_description_create(&desc));
            const size_t nc = (size_t)(n_fft / 2 + 1);
            rocfft_status s;
            if (!inverse)
                s = rocfft_plan_description_set_data_layout(
                    desc, rocfft_array_type_real, rocfft_array_type_hermitian_interleaved, nullptr,
                    nullptr, 1, nullptr, len, 1, nullptr, nc);
            else
                s = rocfft_plan_description_set_data_layout(
                    desc, rocfft_array_type_hermitian_interleaved, rocfft_array_type_real, nullptr,
                    nullptr, 1, nullptr, nc, 1, nullptr, len);
            if (s == rocfft_status_success)
                s = rocfft_plan_create(&p.plan, rocfft_placement_notinplace,
                                       inverse ? rocfft_transform_type_real_inverse
                                               : rocfft_transform_type_real_forward,
                                       rocfft_precision_double, 1, &len, (size_t)batch, desc);
            rocfft_plan_description_destroy(desc);
            if (s != rocfft_status_success)
                return fail(MDX_ERR_ROCFFT, "rocfft_plan_create(n=%lld, batch=%lld) failed (%d)",
                            (long long)n_fft, (long long)batch, (int)s);
            MDX_FFT(rocfft_plan_get_work_buffer_size(p.plan, &p.work_bytes));
            it = store.plans.emplace(key, p).first;
        }
        *out = it->second;
        return MDX_OK;
    }

    int exec(int inverse, int64_t batch, void *in, void *out, hipStream_t stream)
    {
        FftPlan p;
        MDX_TRY(get(inverse, batch, &p));
        if (!info)
            MDX_FFT(rocfft_execution_info_create(&info));
        MDX_FFT(rocfft_execution_info_set_stream(info, stream));
        if (p.work_bytes) {
            if (p.work_bytes > work.bytes) {
                // the previous work buffer may still be in use on the stream
                MDX_HIP(hipStreamSynchronize(stream));
                MDX_TRY(work.ensure(p.work_bytes));
            }
            MDX_FFT(rocfft_execution_info_set_work_buffer(info, work.ptr, p.work_bytes));
        }
        void *ib[1] = {in}, *ob[1] = {out};
        MDX_FFT(rocfft_execute(p.plan, ib, ob, info));
        return MDX_OK;
    }

    // the caller has synchronised the streams the transforms ran on
    void destroy()
    {
        if (info)
            rocfft_execution_info_destroy(info);
        info = nullptr;
        work.recycle();
    }
};

// ------------------------------------------------------------------- kernels

constexpr int GT = 64;   // gather tile: 64 timesteps x 64 (particle, xyz) elements

// X[((b*count + a)*3 + k)][t] = pos[(b*T_b + t)][first + a][k]  (t < T_b), 0 for t >= T_b
__global__ __launch_bounds__(256) void msd_gather_kernel(const double *__restrict__ pos,
                                                         int64_t n_total, int64_t first,
                                                         int64_t count, int64_t t_block,
                                                         int64_t n_fft, int zero_dims,
                                                         double *__restrict__ X)
{
    __shared__ double tile[GT][GT + 1];
    const int b = blockIdx.z;
    const int64_t e0 = int64_t(blockIdx.y) * GT;   // element = a*3 + k within the chunk
    const int64_t t0 = int64_t(blockIdx.x) * GT;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t n_elem = count * 3;
    if (t0 < t_block) {
        for (int r = ty; r < GT; r += 4) {
            int64_t t = t0 + r, e = e0 + tx;
            double v = 0.0;
            if (t < t_block && e < n_elem) {
                v = pos[((int64_t(b) * t_block + t) * n_total + first) * 3 + e];
                if ((zero_dims >> int(e % 3)) & 1)
                    v = 0.0;
            }
            tile[r][tx] = v;
        }
        __syncthreads();
        for (int r = ty; r < GT; r += 4) {
            int64_t e = e0 + r, t = t0 + tx;
            if (e < n_elem && t < n_fft)
                X[(int64_t(b) * n_elem + e) * n_fft + t] = tile[tx][r];
        }
    } else {
        for (int r = ty; r < GT; r += 4) {
            int64_t e = e0 + r, t = t0 + tx;
            if (e < n_elem && t < n_fft)
                X[(int64_t(b) * n_elem + e) * n_fft + t] = 0.0;
        }
    }
}

// per frame row: traj[row][k] += sum_a x_{a,k};  D[row] += sum_{a,k} x^2.  192 threads; a thread reads PAIRS of
// coordinates (16 bytes per lane: 8-byte loads run at 0.54 - 0.70 of that rate), 384 coordinates per round of the
// block, so the dimensions of a thread's two coordinates never change (384 = 0 mod 3); a row that starts 8 bytes off
// a 16-byte boundary gives its first coordinate to thread 0 alone, and the last odd coordinate goes the same way.
__global__ __launch_bounds__(192) void msd_sums_kernel(const double *__restrict__ pos,
                                                       int64_t n_total, int64_t first, int64_t count,
                                                       int zero_dims, double *__restrict__ traj,
                                                       double *__restrict__ D)
{
    __shared__ double s_sum[3][192], s_sq[192];
    const int64_t row = blockIdx.x;
    const int tid = threadIdx.x;
    const double *p = pos + (row * n_total + first) * 3;
    const int64_t n_elem = count * 3;
    const int head = int((reinterpret_cast<uintptr_t>(p) >> 3) & 1);      // coordinates before the first aligned pair
    double a[3] = {0.0, 0.0, 0.0}, q = 0.0;
    const int e0 = head + 2 * tid;                                        // this thread's first coordinate of a round
    const int d0 = e0 % 3, d1 = (e0 + 1) % 3;
    const bool dead0 = (zero_dims >> d0) & 1, dead1 = (zero_dims >> d1) & 1;
    double a0 = 0.0, a1 = 0.0;
    for (int64_t e = e0; e + 1 < n_elem; e += 384) {
        const double2 v = *reinterpret_cast<const double2 *>(p + e);
        const double x = dead0 ? 0.0 : v.x, y = dead1 ? 0.0 : v.y;
        a0 += x;
        a1 += y;
        q = fma(x, x, fma(y, y, q));
    }
    a[d0] += a0;
    a[d1] += a1;
    if (tid == 0) {
        // the coordinates no pair covers: the one before the first aligned pair, the one behind the last whole pair
        if (head && n_elem > 0 && !((zero_dims >> 0) & 1)) {
            a[0] += p[0];
            q = fma(p[0], p[0], q);
        }
        const int64_t covered = head + ((n_elem - head) / 2) * 2;
        if (covered < n_elem && covered >= head) {
            const int dk = int(covered % 3);
            if (!((zero_dims >> dk) & 1)) {
                a[dk] += p[covered];
                q = fma(p[covered], p[covered], q);
            }
        }
    }
    s_sum[0][tid] = a[0];
    s_sum[1][tid] = a[1];
    s_sum[2][tid] = a[2];
    s_sq[tid] = q;
    __syncthreads();
    if (tid < 3) {
        double sa = 0.0;
        for (int i = 0; i < 192; ++i)
            sa += s_sum[tid][i];
        traj[row * 3 + tid] += sa;
    }
    if (tid == 64) {
        double sq = 0.0;
        for (int i = 0; i < 192; ++i)
            sq += s_sq[i];
        D[row] += sq;
    }
}

// P[b][f] += sum over the chunk's series of |F[b][series][f]|^2
__global__ __launch_bounds__(256) void msd_power_kernel(const double2 *__restrict__ F,
                                                        int64_t n_series, int64_t nc,
                                                        double *__restrict__ P)
{
    const int64_t f = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (f >= nc)
        return;
    const double2 *p = F + int64_t(b) * n_series * nc + f;
    double acc = 0.0;
    for (int64_t s = 0; s < n_series; ++s) {
        double2 v = p[s * nc];
        acc = fma(v.x, v.x, fma(v.y, v.y, acc));
    }
    P[int64_t(b) * nc + f] += acc;
}

__global__ void msd_real_to_complex_kernel(const double *__restrict__ P, int64_t n,
                                           double2 *__restrict__ C)
{
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n)
        C[i] = make_double2(P[i], 0.0);
}

// MSD_m = S_m - 2 A_m of one (group, block) row on the device (correlation.py:621-648, summed over the particles):
//   S_m (T_b - m) = 2 sum_k D_k - sum_{k=1..m} (D_{k-1} + D_{T_b-k}),   A_m = acf[m] / n_fft,
// out[m] = (2 total - run_m) / (T_b - m) - acf_factor A_m / (T_b - m)   (acf_factor 2; 1 for the cross displacements of
// mdx_msd_cross, whose spectrum product carries the 2).  One block per row.  `total` is a block-wide sum; the running
// sum is a scan over tiles of 4 096 lags: the tile's terms and its acf values are loaded with consecutive threads on
// consecutive addresses and parked in LDS, thread t then owns the four lags 4 t .. 4 t + 3 of the tile — its partial
// sum goes through a block-wide exclusive scan (wave shuffles, then the sixteen wave totals) —, the finished values go
// back through LDS and are stored the way they were loaded.  (The first form of this kernel gave every thread one
// long segment of the row: every load of a wave touched 64 different lines, and at C4 size — 2 rows of 100 000 lags —
// it took longer than the host loop it had replaced: result() 0.77 -> 1.1 - 1.4 ms.)
constexpr int FINISH_THREADS = 1024;
constexpr int FINISH_PER = 4;                                   // lags per thread and tile
constexpr int FINISH_TILE = FINISH_THREADS * FINISH_PER;
__global__ __launch_bounds__(FINISH_THREADS) void msd_finish_kernel(const double *__restrict__ acf, int64_t acf_stride,
                                                                   const double *__restrict__ D, int64_t t_block,
                                                                   double inv_n, double *__restrict__ out,
                                                                   double acf_factor = 2.0)
{
    __shared__ double sv[FINISH_TILE];        // terms D_{m-1} + D_{T-m}, then the finished values
    __shared__ double sa[FINISH_TILE];        // acf[m]
    __shared__ double wsum[FINISH_THREADS / 64];
    __shared__ double s_total, s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const double *d = D + row * t_block;
    const double *a = acf + row * acf_stride;
    double *o = out + row * t_block;
    // total of D
    double part = 0.0;
    for (int64_t m = tid; m < t_block; m += FINISH_THREADS)
        part += d[m];
    for (int off = 32; off > 0; off >>= 1)
        part += __shfl_down(part, off, 64);
    if (lane == 0)
        wsum[wave] = part;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < FINISH_THREADS / 64; ++w)
            t += wsum[w];
        s_total = t;
        s_carry = 0.0;
    }
    __syncthreads();
    const double total = s_total;
    for (int64_t m0 = 0; m0 < t_block; m0 += FINISH_TILE) {
#pragma unroll
        for (int j = 0; j < FINISH_PER; ++j) {
            const int64_t m = m0 + j * FINISH_THREADS + tid;
            const bool in = m < t_block;
            sv[j * FINISH_THREADS + tid] = (in && m > 0) ? d[m - 1] + d[t_block - m] : 0.0;
            sa[j * FINISH_THREADS + tid] = in ? a[m] : 0.0;
        }
        __syncthreads();
        double v[FINISH_PER], mine = 0.0;
#pragma unroll
        for (int i = 0; i < FINISH_PER; ++i) {
            mine += sv[FINISH_PER * tid + i];
            v[i] = mine;                                  // inclusive within the thread
        }
        // exclusive scan of `mine` over the block
        double incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const double up = __shfl_up(incl, off, 64);
            if (lane >= off)
                incl += up;
        }
        if (lane == 63)
            wsum[wave] = incl;
        __syncthreads();
        double before = s_carry;
        for (int w = 0; w < wave; ++w)
            before += wsum[w];
        before += incl - mine;
#pragma unroll
        for (int i = 0; i < FINISH_PER; ++i) {
            const int64_t m = m0 + FINISH_PER * tid + i;
            const double w = double(t_block - m);
            const double run = before + v[i];
            sv[FINISH_PER * tid + i] = (2.0 * total - run) / w - acf_factor * (sa[FINISH_PER * tid + i] * inv_n) / w;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < FINISH_PER; ++j) {
            const int64_t m = m0 + j * FINISH_THREADS + tid;
            if (m < t_block)
                o[m] = sv[j * FINISH_THREADS + tid];
        }
        if (tid == FINISH_THREADS - 1)
            s_carry = before + mine;                      // everything up to the end of this tile
        __syncthreads();
    }
}

// D[row][t] = r_i(t) . r_j(t) of the summed trajectories of a (pair, block) row (mdx_msd_cross): the per-frame
// products the S_m recurrence of a cross displacement starts from (correlation.py:621-648 with r1 != r2)
__global__ __launch_bounds__(256) void cross_dot_kernel(const double *__restrict__ traj, const int *__restrict__ pairs,
                                                        int n_blocks, int64_t t_block, int64_t row0,
                                                        double *__restrict__ D)
{
    const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (t >= t_block)
        return;
    const int64_t row = row0 + blockIdx.y;
    const int p = int(row / n_blocks), b = int(row - int64_t(p) * n_blocks);
    const double *a = traj + ((int64_t(pairs[2 * p]) * n_blocks + b) * t_block + t) * 3;
    const double *c = traj + ((int64_t(pairs[2 * p + 1]) * n_blocks + b) * t_block + t) * 3;
    D[int64_t(blockIdx.y) * t_block + t] = __dadd_rn(__dadd_rn(__dmul_rn(a[0], c[0]), __dmul_rn(a[1], c[1])),
                                                     __dmul_rn(a[2], c[2]));
}

// [n_series][n_t] -> zero-padded [n_series][n_fft]
__global__ void corr_pad_kernel(const double *__restrict__ in, int64_t n_series, int64_t n_t,
                                int64_t n_fft, double *__restrict__ out)
{
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_series * n_fft)
        return;
    int64_t s = i / n_fft, t = i - s * n_fft;
    out[i] = t < n_t ? in[s * n_t + t] : 0.0;
}

// C = conj(A) * B   (B == nullptr: |A|^2)
__global__ void corr_product_kernel(const double2 *__restrict__ A, const double2 *__restrict__ B,
                                    int64_t n, double2 *__restrict__ C)
{
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    double2 a = A[i];
    if (B) {
        double2 b = B[i];
        C[i] = make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
    } else {
        C[i] = make_double2(a.x * a.x + a.y * a.y, 0.0);
    }
}

// summed trajectories [G][B * T_b][3] -> zero-padded series X[((g * B + b) * 3 + k)][n_fft]
__global__ void cross_pad_kernel(const double *__restrict__ traj, int64_t n_rows /* G * B */, int64_t t_block,
                                 int64_t n_fft, double *__restrict__ X)
{
    const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n_rows * 3 * n_fft)
        return;
    const int64_t s = i / n_fft, t = i - s * n_fft;
    const int64_t row = s / 3;
    const int k = int(s - 3 * row);
    X[i] = t < t_block ? traj[(row * t_block + t) * 3 + k] : 0.0;
}

// Q[p * B + b][f] = sum_k 2 Re(conj(F_i) F_j): the spectrum of corr(a, b)(m) + corr(b, a)(m), summed over xyz
// (rows [row0, row0 + gridDim.y) of the n_pairs * B (pair, block) rows; Q holds this batch only)
__global__ void cross_product_kernel(const double2 *__restrict__ F, const int *__restrict__ pairs, int n_blocks,
                                     int64_t nc, int64_t row0, double2 *__restrict__ Q)
{
    const int64_t f = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (f >= nc)
        return;
    const int64_t row = row0 + blockIdx.y;
    const int p = int(row / n_blocks), b = int(row - int64_t(p) * n_blocks);
    const int gi = pairs[2 * p], gj = pairs[2 * p + 1];
    const double2 *A = F + (int64_t(gi) * n_blocks + b) * 3 * nc + f;
    const double2 *Bv = F + (int64_t(gj) * n_blocks + b) * 3 * nc + f;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double2 a = A[k * nc], c = Bv[k * nc];
        acc += a.x * c.x + a.y * c.y;
    }
    Q[int64_t(blockIdx.y) * nc + f] = make_double2(2.0 * acc, 0.0);
}

}  // namespace

struct mdx_msd {
    int dev = 0;
    hipStream_t stream = nullptr;
    int64_t t_block = 0, n_fft = 0, nc = 0;
    int n_blocks = 0, n_groups = 0;
    FftCache fft;
    // accumulators: power [G][B][nc] + D [G][B*T_b] contiguous (one all-reduce), traj [G][B*T_b][3]
    DeviceBuffer d_acc, d_traj, d_series, d_spec, d_stage, d_inv_in, d_inv_out, d_finish;
    DeviceBuffer d_f32, d_index, d_prev, d_image;   // trajectory-file path: frames, unwrap state
    DeviceBuffer d_cross_f;                         // spectra of the summed trajectories (mdx_msd_cross)
    DeviceBuffer d_xstart, d_seg;                   // ... of the segment-parallel walk (msd_launch_unwrap)
    DeviceBuffer d_masses, d_com_x, d_shift;        // system centre of mass per frame
    // trajectory-file path with groupings="residues"/"segments": rows sorted molecule by molecule
    std::vector<int64_t> mol_offsets;               // CSR over the rows of a push_traj call
    std::vector<int> images0;                       // initial image flags [n_sel][3] of the next unwrapped pushes
    double mol_mass = 0.0;                          // total mass of the grouping's molecules
    DeviceBuffer d_mol_offsets, d_mol_masses, d_mol_total, d_mol_com;
    bool own_fft = false;                           // n_fft is one of the shapes of mdx_msd_fft.hpp
    bool fused_sums = false;                        // pass A of this shape also forms the per-frame sums
    bool single = false;                            // <= 800 frames per block: one pass, no Y (msd_fft_single400_kernel)
    DeviceBuffer d_part;                            // its partial-sum records
    msdfft::Shape shape;
    DeviceBuffer d_tw, d_pfull;                     // twiddle tables [2][512], full-spectrum sums [B][N]
    StreamTimer timer;
    int64_t bytes_moved = 0;
    double *power(int g) { return d_acc.as<double>() + int64_t(g) * n_blocks * nc; }
    double *dsq(int g)
    {
        return d_acc.as<double>() + int64_t(n_groups) * n_blocks * nc + int64_t(g) * n_blocks * t_block;
    }
    double *traj(int g) { return d_traj.as<double>() + int64_t(g) * n_blocks * t_block * 3; }
    int64_t acc_len() const { return int64_t(n_groups) * n_blocks * (nc + t_block); }
    int64_t traj_len() const { return int64_t(n_groups) * n_blocks * t_block * 3; }
};

// d_pos32: the positions as float32 instead (d_pos unused): only where pass A reads them itself
// (msdfft::cols_read_f32 — the 400-point family with two passes); other shapes answer MDX_ERR_UNSUPPORTED and the
// caller widens the frames first (mdx_msd_push_frames_device)
static int msd_push_device(mdx_msd *h, int group, const double *d_pos, int64_t n_total, int64_t first,
                           int64_t count, int zero_dims, const float *d_pos32 = nullptr)
{
    if (d_pos32 && !(h->own_fft && (h->single || h->fused_sums) && msdfft::cols_read_f32(h->shape)))
        return fail(MDX_ERR_UNSUPPORTED, "float32 positions are read in place by the transforms with a 400- or 64-point first factor only "
                    "(this engine: n_fft = %lld)", (long long)h->n_fft);
    if (count == 0)
        return MDX_OK;
    const int B = h->n_blocks;
    // chunk the particles so that series + spectrum stay within ~40% of free HBM.  The driver is asked for the
    // free memory only when the buffers the handle already holds do not take the whole push as one chunk:
    // hipMemGetInfo costs the host 1 - 2 ms, and right after a synchronisation point (every analysis starts with
    // one) the device idles for as long.
    const int64_t per_atom = h->single    ? 0                    // nothing but the accumulators
                             : h->own_fft ? 3 * B * h->n_fft * 8   // Y: one complex per two reals
                                          : 3 * B * (h->n_fft * 8 + h->nc * 16);
    int64_t chunk = count;
    if (!h->single && size_t(count + 6) * per_atom > h->d_series.bytes + h->d_spec.bytes) {
        size_t free_b = 0, total_b = 0;
        MDX_HIP(hipMemGetInfo(&free_b, &total_b));
        free_b += h->d_series.bytes + h->d_spec.bytes + cached_device_bytes(h->dev);
        chunk = std::max<int64_t>(1, int64_t(double(free_b) * 0.4) / per_atom);
    }
    chunk = std::min<int64_t>(chunk, count);
    // a handful of equal chunks keeps the rocFFT plan cache small
    const int64_t n_chunks = ceil_div(count, chunk);
    chunk = ceil_div(count, n_chunks);
    if (h->single) {
        MDX_TRY(h->d_part.ensure(msdfft::single_part_bytes(h->shape, B)));
    } else if (h->own_fft) {
        const int64_t pmul = int64_t(msdfft::PG) * msdfft::pair_group_multiple(h->shape);
        const int64_t p_pad_max = ceil_div(ceil_div(chunk * 3 + 15, 2), pmul) * pmul;   // (+ 15: head)
        MDX_TRY(h->d_spec.ensure(size_t(B) * h->n_fft * p_pad_max * 16));
        if (h->fused_sums)
            MDX_TRY(h->d_part.ensure(msdfft::fused_part_bytes(h->shape, (int)p_pad_max, B)));
    } else {
        MDX_TRY(h->d_series.ensure(size_t(chunk) * 3 * B * h->n_fft * 8));
        MDX_TRY(h->d_spec.ensure(size_t(chunk) * 3 * B * h->nc * 16));
    }
    hipEvent_t ev = h->timer.begin();
    for (int64_t a0 = 0; a0 < count; a0 += chunk) {
        const int64_t c = std::min(chunk, count - a0);
        const int64_t n_elem = c * 3;
        if (!h->fused_sums && !h->single)
            hipLaunchKernelGGL(msd_sums_kernel, dim3((unsigned)(B * h->t_block)), dim3(192), 0, h->stream,
                               d_pos, n_total, first + a0, c, zero_dims, h->traj(group), h->dsq(group));
        if (h->own_fft) {
            // tables: half table of W_R1, half table of W_R2, W_N^m for m < R2
            const double2 *tw_r1 = h->d_tw.as<double2>(), *tw_r2 = tw_r1 + msdfft::tw_r1_len(h->shape.r1),
                          *twN = tw_r2 + h->shape.r2 / 2;
            // (batches of particles whose half-transformed block fits the 256 MB memory-side cache were measured in
            // rounds 1 and 2: small launches lose more than the cache returns — NOTES.md — so a chunk is one launch)
            // rows whose length is a multiple of 128 bytes (and positions from hipMalloc): a chunk that starts
            // in the middle of a line is entered `head` coordinates early, so that pass A's 128-byte pieces
            // are whole lines (msd_fft_cols400_fused_kernel); the head is staged as zeros
            const int head = ((h->single || (h->fused_sums && msdfft::aligns_head(h->shape))) && (n_total * 3) % 16 == 0 &&
                              (reinterpret_cast<uintptr_t>(d_pos32 ? static_cast<const void *>(d_pos32)
                                                                   : static_cast<const void *>(d_pos)) & 127u) == 0)
                                 ? int(((first + a0) * 3) % 16)
                                 : 0;
            const int64_t ne = c * 3 + head;
            const int64_t pmul = int64_t(msdfft::PG) * msdfft::pair_group_multiple(h->shape);
            const int p_pad = (int)(ceil_div(ceil_div(ne, 2), pmul) * pmul);
            if (h->single) {
                // one pass: positions -> |F|^2 sums and per-frame sums, nothing written in between
                const int parts = msdfft::launch_single(h->shape, h->stream, d_pos, n_total, first + a0, ne, h->t_block, B,
                                                        zero_dims, p_pad, tw_r1, twN, h->d_pfull.as<double>(),
                                                        h->d_part.as<double2>(), h->traj(group), h->dsq(group), head,
                                                        d_pos32);
                hipLaunchKernelGGL(msdfft::msd_power_fold_kernel, dim3((unsigned)ceil_div(h->nc, 256), (unsigned)B),
                                   dim3(256), 0, h->stream, h->d_pfull.as<double>(), h->shape.r1, h->shape.r2, parts,
                                   h->nc, h->power(group));
                h->bytes_moved += c * 3 * B * h->t_block * (d_pos32 ? 4 : 8);      // the positions, once
                continue;
            }
            msdfft::launch(h->shape, h->stream, d_pos, n_total, first + a0, ne, h->t_block, B, zero_dims, p_pad,
                           tw_r1, tw_r2, twN, h->d_spec.as<double2>(), h->d_pfull.as<double>(), 0,
                           h->fused_sums ? h->d_part.as<double2>() : nullptr, h->traj(group), h->dsq(group), head, d_pos32);
            msdfft::launch_fold(h->shape, h->stream, h->d_pfull.as<double>(), B, h->nc, h->power(group));
            // positions read once (twice where the sums are a kernel of their own), Y written and read once
            h->bytes_moved += c * 3 * B * ((h->fused_sums ? 1 : 2) * h->t_block * (d_pos32 ? 4 : 8) + 2 * h->n_fft * 8);
            continue;
        }
        dim3 g1((unsigned)ceil_div(h->n_fft, GT), (unsigned)ceil_div(n_elem, GT), (unsigned)B);
        hipLaunchKernelGGL(msd_gather_kernel, g1, dim3(256), 0, h->stream, d_pos, n_total,
                           first + a0, c, h->t_block, h->n_fft, zero_dims, h->d_series.as<double>());
        MDX_TRY(h->fft.exec(0, n_elem * B, h->d_series.ptr, h->d_spec.ptr, h->stream));
        hipLaunchKernelGGL(msd_power_kernel, dim3((unsigned)ceil_div(h->nc, 256), (unsigned)B),
                           dim3(256), 0, h->stream, h->d_spec.as<double2>(), n_elem, h->nc,
                           h->power(group));
        // algorithmic traffic of this chunk: positions read twice, series written + read,
        // spectrum written + read
        h->bytes_moved += c * 3 * B * (2 * h->t_block * 8 + 2 * h->n_fft * 8 + 2 * h->nc * 16);
    }
    h->timer.end(ev);
    MDX_HIP(hipGetLastError());
    return MDX_OK;
}

// Frame preparation on the device: float32 frames -> the float64
// [frame][particle][xyz] block the correlation kernels read, optionally unwrapped across the
// periodic boundaries.  Restates `unwrap` (reference src/mdhelper/algorithm/topology.py:366-376)
// per coordinate: d = x - x_old; |d| >= L/2 -> image -= sign(d); x_old = x; out = x + image * L,
// with the same float64 operations (one multiply, one add, no contraction), so the unwrapped
// trajectory is the one the reference builds.  One thread owns one coordinate and walks the
// frames of the block in order; the state (x_old, image) persists between blocks.
// In: float (what an MDAnalysis reader and the trajectory files deliver) or double (in-memory
// float64 trajectories).
// Source: `in` holds frames of `in_stride` coordinates; the block's coordinate e = 3 a + k is read at
// 3 rows[a] + k when `rows` is given (frames resident in HBM, a particle selection) and at e otherwise
// (a staged block of the selection, in_stride = n_coord).
//
// The only state the walk carries from frame to frame is the integer image flag (x_old is the RAW
// previous coordinate), so the frames of a block are cut into segments of SEG frames that run in
// parallel: (1) msd_image_delta_kernel — net flag change of every (segment, coordinate); (2)
// msd_image_scan_kernel — running sum over the segments, from the flag the block starts with; (3)
// msd_unwrap_widen_kernel — every segment walks its frames from its own starting flag.  Without unwrapping
// only (3) runs.  The operations on a coordinate are the serial walk's, in its order.
constexpr int UNWRAP_SEG = 128;

template <typename In>
__device__ inline int image_step(In x, In x_old, double half)
{
    const double d = __dsub_rn((double)x, (double)x_old);
    return fabs(d) >= half ? (d < 0.0) - (d > 0.0) : 0;
}

template <typename In>
__global__ __launch_bounds__(256) void msd_image_delta_kernel(
    const In *__restrict__ in, int64_t in_stride, const int *__restrict__ rows, int64_t n_coord,
    int64_t n_frames, int first_block, double lx, double ly, double lz, const In *__restrict__ prev,
    int *__restrict__ delta /* [n_seg][n_coord] */)
{
    const int64_t e = blockIdx.x * int64_t(256) + threadIdx.x;
    if (e >= n_coord)
        return;
    const int k = int(e % 3);
    const int64_t src = rows ? int64_t(rows[e / 3]) * 3 + k : e;
    const double half = 0.5 * (k == 0 ? lx : (k == 1 ? ly : lz));
    const int64_t f_lo = int64_t(blockIdx.y) * UNWRAP_SEG;
    const int64_t f_hi = f_lo + UNWRAP_SEG < n_frames ? f_lo + UNWRAP_SEG : n_frames;
    In x_old = f_lo > 0 ? in[(f_lo - 1) * in_stride + src] : (first_block ? in[src] : prev[e]);
    int acc = 0;
    for (int64_t f = f_lo; f < f_hi; ++f) {
        const In x = in[f * in_stride + src];
        acc += image_step(x, x_old, half);
        x_old = x;
    }
    delta[int64_t(blockIdx.y) * n_coord + e] = acc;
}

// delta[s][e] <- flag segment s starts with; image[e] <- flag after the block; x_start[e] <- the raw
// coordinate before the block's first frame; prev[e] <- the block's last raw coordinate
template <typename In>
__global__ __launch_bounds__(256) void msd_image_scan_kernel(
    const In *__restrict__ in, int64_t in_stride, const int *__restrict__ rows, int64_t n_coord,
    int64_t n_frames, int n_seg, int first_block, In *__restrict__ prev, In *__restrict__ x_start,
    int *__restrict__ image, int *__restrict__ delta)
{
    const int64_t e = blockIdx.x * int64_t(256) + threadIdx.x;
    if (e >= n_coord)
        return;
    const int64_t src = rows ? int64_t(rows[e / 3]) * 3 + int(e % 3) : e;
    // first_block: 1 = the walk starts here without image flags; 2 = it starts here from the flags in
    // `image` (molecules made whole in the first analysed frame, transport.py:936-941)
    int img = first_block == 1 ? 0 : image[e];
    for (int s = 0; s < n_seg; ++s) {
        const int t = delta[int64_t(s) * n_coord + e];
        delta[int64_t(s) * n_coord + e] = img;
        img += t;
    }
    image[e] = img;
    x_start[e] = first_block ? in[src] : prev[e];
    prev[e] = in[(n_frames - 1) * in_stride + src];
}

template <typename In>
__global__ __launch_bounds__(256) void msd_unwrap_widen_kernel(
    const In *__restrict__ in, int64_t in_stride, const int *__restrict__ rows, int64_t n_coord,
    int64_t n_frames, int unwrap, double lx, double ly, double lz, const In *__restrict__ x_start,
    const int *__restrict__ seg_image /* [n_seg][n_coord] */, double *__restrict__ out,
    const double *__restrict__ shift /* [n_frames][3] or nullptr */)
{
    const int64_t e = blockIdx.x * int64_t(256) + threadIdx.x;
    if (e >= n_coord)
        return;
    const int k = int(e % 3);
    const int64_t src = rows ? int64_t(rows[e / 3]) * 3 + k : e;
    const double L = k == 0 ? lx : (k == 1 ? ly : lz);
    const double half = 0.5 * L;
    const int64_t f_lo = int64_t(blockIdx.y) * UNWRAP_SEG;
    const int64_t f_hi = f_lo + UNWRAP_SEG < n_frames ? f_lo + UNWRAP_SEG : n_frames;
    In x_old = 0;
    int img = 0;
    if (unwrap) {
        x_old = f_lo > 0 ? in[(f_lo - 1) * in_stride + src] : x_start[e];
        img = seg_image[int64_t(blockIdx.y) * n_coord + e];
    }
    for (int64_t f = f_lo; f < f_hi; ++f) {
        const In x = in[f * in_stride + src];
        double v = (double)x;
        if (unwrap) {
            img += image_step(x, x_old, half);
            x_old = x;
            v = __dadd_rn(v, __dmul_rn((double)img, L));
        }
        if (shift)   // system centre of mass of this frame (transport.py:993-1014)
            v = __dsub_rn(v, shift[3 * f + k]);
        out[f * n_coord + e] = v;
    }
}

// out[f][k] = sum_a m_a x[f][a][k] / sum_a m_a; wrap != 0: coordinates outside [0, L] are first
// brought back with x -= floor(x / L) L (algorithm/topology.py `wrap`).  One block per frame.
__global__ __launch_bounds__(256) void msd_frame_com_kernel(const double *__restrict__ x, int64_t n,
                                                           const double *__restrict__ masses,
                                                           double inv_total, int wrap, double lx,
                                                           double ly, double lz, double *__restrict__ out)
{
    __shared__ double red[3][256];
    const int64_t f = blockIdx.x;
    const int tid = threadIdx.x;
    const double L[3] = {lx, ly, lz};
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t a = tid; a < n; a += 256) {
        const double m = masses[a];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            double v = x[(f * n + a) * 3 + k];
            if (wrap && (v < 0.0 || v > L[k]))
                v -= floor(v / L[k]) * L[k];
            acc[k] = fma(m, v, acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
        red[k][tid] = acc[k];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                red[k][tid] += red[k][tid + off];
        __syncthreads();
    }
    if (tid < 3)
        out[3 * f + tid] = red[tid][0] * inv_total;
}

// System centre of mass of every listed frame (transport.py:993-1014), positions unwrapped first
// when requested — the state walk needs every listed particle, so the frames are staged whole.
// Where the float32 frames of the frame-preparation path come from: rows [a0, a0 + c) of the
// selection for frames [f0, f0 + nf) of the analysed series -> d_out float32[nf][c][3], queued on
// the engine's stream.
struct FrameSource {
    virtual ~FrameSource() = default;
    virtual int elem() const { return 4; }   // bytes per staged coordinate: float32, or float64 frames
    virtual int prepare(mdx_msd *h, int64_t n_sel) = 0;
    virtual int stage(mdx_msd *h, int64_t a0, int64_t c, int64_t f0, int64_t nf, void *d_out) = 0;
    // frames the unwrap kernel can read where they lie (HBM-resident trajectories): pointer to frame f0,
    // coordinates per frame, rows of the selection from a0 on (device int32, or nullptr: rows a0, a0+1, ...
    // are folded into the pointer).  nullptr: the block has to be staged.
    virtual const void *direct(mdx_msd *, int64_t, int64_t, int64_t *, const int **) { return nullptr; }
    virtual bool resident() const { return false; }
};

// frames per unwrap launch: ~64 MB of staged frames, or (resident sources: nothing is staged) as many as the
// grid's second dimension takes
static int64_t msd_frame_block(const FrameSource &src, int64_t n_frames, int64_t n_rows)
{
    const int64_t most = src.resident() ? int64_t(32768) * UNWRAP_SEG
                                        : (int64_t(64) << 20) / (int64_t(3) * src.elem() * n_rows);
    return std::max<int64_t>(1, std::min(n_frames, most));
}

static int msd_unwrap_buffers(mdx_msd *h, const FrameSource &src, int64_t n_rows, int64_t block)
{
    const int64_t row_bytes = int64_t(3) * src.elem();
    MDX_TRY(h->d_prev.ensure(size_t(n_rows) * row_bytes));
    MDX_TRY(h->d_xstart.ensure(size_t(n_rows) * row_bytes));
    MDX_TRY(h->d_image.ensure(size_t(n_rows) * 12));
    MDX_TRY(h->d_seg.ensure(size_t(ceil_div(block, UNWRAP_SEG)) * n_rows * 12));
    if (!src.resident())
        MDX_TRY(h->d_f32.ensure(size_t(block) * n_rows * row_bytes));
    return MDX_OK;
}

// unwrap + widen (+ shift) of one staged block of frames, in the block's element type
static void msd_launch_unwrap(mdx_msd *h, FrameSource &src, int64_t n_coord, int64_t nf, int first,
                              int unwrap, const double *dims, double *d_out, const double *d_shift,
                              int64_t a0, int64_t f0);

// a trajectory file: listed frames, listed particles (gathered by the unpack kernel)
struct TrajFrames final : FrameSource {
    Trajectory *t;
    const int64_t *frames;
    const int32_t *index;
    TrajFrames(Trajectory *t_, const int64_t *f, const int32_t *i) : t(t_), frames(f), index(i) {}
    int prepare(mdx_msd *h, int64_t n_sel) override
    {
        std::vector<int32_t> iota;
        if (!index) {
            iota.resize(size_t(n_sel));
            for (int64_t i = 0; i < n_sel; ++i)
                iota[size_t(i)] = (int32_t)i;
        }
        MDX_HIP(hipStreamSynchronize(h->stream));
        MDX_TRY(h->d_index.ensure(size_t(4) * n_sel));
        MDX_HIP(hipMemcpy(h->d_index.ptr, index ? index : iota.data(), size_t(4) * n_sel,
                          hipMemcpyHostToDevice));
        return MDX_OK;
    }
    int stage(mdx_msd *h, int64_t a0, int64_t c, int64_t f0, int64_t nf, void *d_out) override
    {
        TrajSelection sel{h->d_index.as<int>() + a0, c, static_cast<float *>(d_out)};
        return t->stage_async(h->dev, h->stream, frames + f0, nf, &sel, 1);
    }
};

// host memory: float32 or float64 [n_frames][n_sel][3], the selection already gathered by the caller
template <typename In> struct HostFrames final : FrameSource {
    const In *pos;
    int64_t n_sel;
    HostFrames(const In *p, int64_t n) : pos(p), n_sel(n) {}
    int elem() const override { return (int)sizeof(In); }
    int prepare(mdx_msd *h, int64_t) override
    {
        MDX_HIP(hipStreamSynchronize(h->stream));
        return MDX_OK;
    }
    int stage(mdx_msd *h, int64_t a0, int64_t c, int64_t f0, int64_t nf, void *d_out) override
    {
        // (one 2-D DMA where the rows lie — page-locked memory, or pageable rows of >= 4 KB through the runtime's copy —
        // or the pinned ring's gather for short rows: HostStager::upload_rows)
        const size_t row = 3 * sizeof(In);
        return device_stager(h->dev).upload_rows(h->dev, h->stream, d_out, pos + (f0 * n_sel + a0) * 3, row * c,
                                                 row * n_sel, (size_t)nf);
    }
};

// frames resident in HBM: float32 or float64 [n_frames][n_total][3]; selection = listed rows (host int32,
// uploaded once) or the first n_sel rows.  Nothing is staged: the unwrap kernel gathers.
struct DeviceFrames final : FrameSource {
    const void *d_pos;
    int elem_bytes;
    int64_t n_total;
    const int32_t *index;
    DeviceFrames(const void *p, int eb, int64_t nt, const int32_t *i) : d_pos(p), elem_bytes(eb), n_total(nt), index(i) {}
    int elem() const override { return elem_bytes; }
    int prepare(mdx_msd *h, int64_t n_sel) override
    {
        MDX_HIP(hipStreamSynchronize(h->stream));
        if (index) {
            MDX_TRY(h->d_index.ensure(size_t(4) * n_sel));
            MDX_HIP(hipMemcpy(h->d_index.ptr, index, size_t(4) * n_sel, hipMemcpyHostToDevice));
        }
        return MDX_OK;
    }
    int stage(mdx_msd *, int64_t, int64_t, int64_t, int64_t, void *) override { return MDX_OK; }
    bool resident() const override { return true; }
    const void *direct(mdx_msd *h, int64_t a0, int64_t f0, int64_t *stride, const int **rows) override
    {
        *stride = n_total * 3;
        const char *base = static_cast<const char *>(d_pos) + size_t(f0) * n_total * 3 * elem_bytes;
        if (index) {
            *rows = h->d_index.as<int>() + a0;
            return base;
        }
        *rows = nullptr;
        return base + size_t(a0) * 3 * elem_bytes;
    }
};

static void msd_launch_unwrap(mdx_msd *h, FrameSource &src, int64_t n_coord, int64_t nf, int first,
                              int unwrap, const double *dims, double *d_out, const double *d_shift,
                              int64_t a0, int64_t f0)
{
    const dim3 grid((unsigned)ceil_div(n_coord, 256));
    const double lx = dims ? dims[0] : 0.0, ly = dims ? dims[1] : 0.0, lz = dims ? dims[2] : 0.0;
    if (first && unwrap && !h->images0.empty()) {
        // initial image flags of rows [a0, a0 + n_coord / 3) (mdx_msd_set_initial_images)
        (void)hipMemcpyAsync(h->d_image.ptr, h->images0.data() + 3 * a0, size_t(4) * n_coord,
                             hipMemcpyHostToDevice, h->stream);
        first = 2;
    }
    // frames read where they lie (HBM-resident source) or from the staged block
    int64_t stride = n_coord;
    const int *rows = nullptr;
    const void *in = src.direct(h, a0, f0, &stride, &rows);
    if (!in) {
        in = h->d_f32.ptr;
        stride = n_coord;
        rows = nullptr;
    }
    const int n_seg = (int)ceil_div(nf, UNWRAP_SEG);
    const dim3 grid2(grid.x, (unsigned)n_seg);
    auto run = [&](auto zero) {
        using In = decltype(zero);
        const In *p = static_cast<const In *>(in);
        if (unwrap) {
            hipLaunchKernelGGL(msd_image_delta_kernel<In>, grid2, dim3(256), 0, h->stream, p, stride, rows, n_coord,
                               nf, first, lx, ly, lz, h->d_prev.as<In>(), h->d_seg.as<int>());
            hipLaunchKernelGGL(msd_image_scan_kernel<In>, grid, dim3(256), 0, h->stream, p, stride, rows, n_coord, nf,
                               n_seg, first, h->d_prev.as<In>(), h->d_xstart.as<In>(), h->d_image.as<int>(),
                               h->d_seg.as<int>());
        }
        hipLaunchKernelGGL(msd_unwrap_widen_kernel<In>, grid2, dim3(256), 0, h->stream, p, stride, rows, n_coord, nf,
                           unwrap, lx, ly, lz, h->d_xstart.as<In>(), h->d_seg.as<int>(), d_out, d_shift);
    };
    if (src.elem() == 8)
        run(double(0));
    else
        run(float(0));
}

__global__ void msd_molecule_com_kernel(const double *__restrict__ x, int64_t c, int64_t a0,
                                        const int64_t *__restrict__ offsets,
                                        const double *__restrict__ masses,
                                        const double *__restrict__ total, int64_t m0, int64_t n_mol,
                                        const double *__restrict__ shift, double *__restrict__ out);

// With a grouping declared (mdx_msd_set_grouping) the selection's rows are particles of molecules and
// the result is the centre of mass of the molecules' (optionally wrapped) CENTRES, each weighted
// with its molecule's mass — Onsager(center=True, center_atom=False) with residue / segment
// groupings (transport.py:1004-1014: wrap(frame) acts on the centres); `masses` is then ignored
// in favour of the grouping's.
static int msd_system_com_frames(mdx_msd *h, FrameSource &src, int64_t n_frames, int64_t n_sel,
                                 const double *masses, int unwrap, const double *dims, int wrap,
                                 double *out)
{
    const bool molecules = !h->mol_offsets.empty();
    const int64_t n_mol = molecules ? (int64_t)h->mol_offsets.size() - 1 : 0;
    if (molecules)
        MDX_REQUIRE(h->mol_offsets.back() == n_sel,
                    "%lld particles given, the grouping was defined for %lld", (long long)n_sel,
                    (long long)h->mol_offsets.back());
    double total = 0.0;
    for (int64_t i = 0; i < n_sel; ++i)
        total += masses[i];
    MDX_REQUIRE(total > 0.0, "the selection has no mass");
    MDX_REQUIRE(!unwrap || h->images0.empty() || (int64_t)h->images0.size() == 3 * n_sel,
                "initial image flags were set for %lld particles, %lld are staged",
                (long long)h->images0.size() / 3, (long long)n_sel);
    MDX_TRY(src.prepare(h, n_sel));
    MDX_TRY(h->d_masses.ensure(size_t(8) * n_sel));
    MDX_HIP(hipMemcpy(h->d_masses.ptr, masses, size_t(8) * n_sel, hipMemcpyHostToDevice));
    const int64_t block = msd_frame_block(src, n_frames, n_sel);
    MDX_TRY(msd_unwrap_buffers(h, src, n_sel, block));
    MDX_TRY(h->d_com_x.ensure(size_t(block) * n_sel * 24));
    MDX_TRY(h->d_shift.ensure(size_t(24) * n_frames));
    for (int64_t f0 = 0; f0 < n_frames; f0 += block) {
        const int64_t nf = std::min(block, n_frames - f0);
        MDX_TRY(src.stage(h, 0, n_sel, f0, nf, h->d_f32.ptr));
        msd_launch_unwrap(h, src, 3 * n_sel, nf, f0 == 0 ? 1 : 0, unwrap, dims, h->d_com_x.as<double>(),
                          nullptr, 0, f0);
        if (molecules) {
            MDX_TRY(h->d_mol_com.ensure(size_t(nf) * n_mol * 24));
            hipLaunchKernelGGL(msd_molecule_com_kernel, dim3((unsigned)ceil_div(n_mol * 3, 256), (unsigned)nf),
                               dim3(256), 0, h->stream, h->d_com_x.as<double>(), n_sel, int64_t(0),
                               h->d_mol_offsets.as<int64_t>(), h->d_mol_masses.as<double>(),
                               h->d_mol_total.as<double>(), int64_t(0), n_mol, (const double *)nullptr,
                               h->d_mol_com.as<double>());
            hipLaunchKernelGGL(msd_frame_com_kernel, dim3((unsigned)nf), dim3(256), 0, h->stream,
                               h->d_mol_com.as<double>(), n_mol, h->d_mol_total.as<double>(), 1.0 / h->mol_mass,
                               wrap, dims ? dims[0] : 1.0, dims ? dims[1] : 1.0, dims ? dims[2] : 1.0,
                               h->d_shift.as<double>() + 3 * f0);
        } else {
            hipLaunchKernelGGL(msd_frame_com_kernel, dim3((unsigned)nf), dim3(256), 0, h->stream,
                               h->d_com_x.as<double>(), n_sel, h->d_masses.as<double>(), 1.0 / total, wrap,
                               dims ? dims[0] : 1.0, dims ? dims[1] : 1.0, dims ? dims[2] : 1.0,
                               h->d_shift.as<double>() + 3 * f0);
        }
        MDX_HIP(hipGetLastError());
    }
    MDX_HIP(hipMemcpyAsync(out, h->d_shift.ptr, size_t(24) * n_frames, hipMemcpyDeviceToHost, h->stream));
    MDX_HIP(hipStreamSynchronize(h->stream));
    return MDX_OK;
}

// Centres of mass of molecules [m0, m0 + n_mol) from the unwrapped float64 block x[frame][c][3] of
// their particles (rows a0 .. a0 + c of the selection): sum_a m_a x_a in row order with separate
// multiply and add, one division — numpy.bincount(weights=m * x) / bincount(weights=m), the
// reference's center_of_mass (algorithm/molecule.py:300-306) — then the frame's shift.
__global__ void msd_molecule_com_kernel(
    const double *__restrict__ x, int64_t c, int64_t a0, const int64_t *__restrict__ offsets,
    const double *__restrict__ masses, const double *__restrict__ total, int64_t m0, int64_t n_mol,
    const double *__restrict__ shift, double *__restrict__ out)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;   // (molecule, k)
    const int64_t f = blockIdx.y;
    if (i >= n_mol * 3)
        return;
    const int64_t m = i / 3;
    const int k = int(i - 3 * m);
    const double *row = x + f * c * 3 + k;
    double acc = 0.0;
    for (int64_t a = offsets[m0 + m]; a < offsets[m0 + m + 1]; ++a)
        acc = __dadd_rn(acc, __dmul_rn(masses[a], row[3 * (a - a0)]));
    double v = acc / total[m0 + m];
    if (shift)
        v = __dsub_rn(v, shift[3 * f + k]);
    out[(f * n_mol + m) * 3 + k] = v;
}

// The first n_blocks * t_block listed frames of a trajectory file -> one group of the engine.
static int msd_push_frames(mdx_msd *h, int group, FrameSource &src, int64_t n_sel, int unwrap,
                           const double *dims, int zero_dims, const double *shift)
{
    const int64_t T = int64_t(h->n_blocks) * h->t_block;
    const bool molecules = !h->mol_offsets.empty();
    const int64_t n_mol = molecules ? (int64_t)h->mol_offsets.size() - 1 : 0;
    if (molecules)
        MDX_REQUIRE(h->mol_offsets.back() == n_sel,
                    "%lld particles given, the grouping was defined for %lld", (long long)n_sel,
                    (long long)h->mol_offsets.back());
    // particle chunks: the float64 block of a chunk within ~30 % of the free HBM (with molecules:
    // their centres need a second, smaller block)
    size_t free_b = 0, total_b = 0;
    MDX_HIP(hipMemGetInfo(&free_b, &total_b));
    free_b += h->d_stage.bytes + h->d_mol_com.bytes + cached_device_bytes(h->dev);
    int64_t chunk = std::max<int64_t>(1, int64_t(double(free_b) * (molecules ? 0.2 : 0.3)) / (T * 24));
    chunk = std::min(chunk, n_sel);
    chunk = ceil_div(n_sel, ceil_div(n_sel, chunk));
    // chunk boundaries in rows: whole molecules only
    std::vector<int64_t> cuts{0};
    if (molecules) {
        int64_t largest = 0;
        for (int64_t m = 0; m < n_mol; ++m)
            largest = std::max(largest, h->mol_offsets[size_t(m) + 1] - h->mol_offsets[size_t(m)]);
        chunk = std::max(chunk, largest);
        for (int64_t m = 0; m < n_mol; ++m)
            if (h->mol_offsets[size_t(m) + 1] - cuts.back() > chunk)
                cuts.push_back(h->mol_offsets[size_t(m)]);
        cuts.push_back(n_sel);
    } else {
        for (int64_t a0 = chunk; a0 < n_sel; a0 += chunk)
            cuts.push_back(a0);
        cuts.push_back(n_sel);
    }
    MDX_REQUIRE(!unwrap || h->images0.empty() || (int64_t)h->images0.size() == 3 * n_sel,
                "initial image flags were set for %lld particles, %lld are staged",
                (long long)h->images0.size() / 3, (long long)n_sel);
    MDX_TRY(src.prepare(h, n_sel));
    MDX_TRY(h->d_stage.ensure(size_t(T) * chunk * 24));
    const int64_t block = msd_frame_block(src, T, chunk);
    MDX_TRY(msd_unwrap_buffers(h, src, chunk, block));
    if (shift) {
        MDX_TRY(h->d_shift.ensure(size_t(24) * T));
        MDX_HIP(hipMemcpy(h->d_shift.ptr, shift, size_t(24) * T, hipMemcpyHostToDevice));
    }
    // with molecules the shift is subtracted from the centres, as the reference orders it
    // (transport.py:983-1014), not from the particles
    const bool shift_rows = shift && !molecules;
    int64_t m_lo = 0;
    for (size_t ci = 0; ci + 1 < cuts.size(); ++ci) {
        const int64_t a0 = cuts[ci], c = cuts[ci + 1] - a0;
        if (c == 0)
            continue;
        for (int64_t f0 = 0; f0 < T; f0 += block) {
            const int64_t nf = std::min(block, T - f0);
            MDX_TRY(src.stage(h, a0, c, f0, nf, h->d_f32.ptr));
            msd_launch_unwrap(h, src, 3 * c, nf, f0 == 0 ? 1 : 0, unwrap, dims,
                              h->d_stage.as<double>() + f0 * c * 3,
                              shift_rows ? h->d_shift.as<double>() + 3 * f0 : nullptr, a0, f0);
            MDX_HIP(hipGetLastError());
        }
        if (!molecules) {
            MDX_TRY(msd_push_device(h, group, h->d_stage.as<double>(), c, 0, c, zero_dims));
            continue;
        }
        int64_t m_hi = m_lo;
        while (m_hi < n_mol && h->mol_offsets[size_t(m_hi) + 1] <= a0 + c)
            ++m_hi;
        const int64_t nm = m_hi - m_lo;
        MDX_TRY(h->d_mol_com.ensure(size_t(T) * nm * 24));
        hipLaunchKernelGGL(msd_molecule_com_kernel, dim3((unsigned)ceil_div(nm * 3, 256), (unsigned)T),
                           dim3(256), 0, h->stream, h->d_stage.as<double>(), c, a0,
                           h->d_mol_offsets.as<int64_t>(), h->d_mol_masses.as<double>(),
                           h->d_mol_total.as<double>(), m_lo, nm,
                           shift ? h->d_shift.as<double>() : (const double *)nullptr,
                           h->d_mol_com.as<double>());
        MDX_HIP(hipGetLastError());
        MDX_TRY(msd_push_device(h, group, h->d_mol_com.as<double>(), nm, 0, nm, zero_dims));
        m_lo = m_hi;
    }
    return MDX_OK;
}

extern "C" {

int mdx_msd_create(mdx_msd_t *out, int dev, int64_t n_frames_block, int n_blocks, int n_groups)
{
    MDX_REQUIRE(out, "NULL argument");
    MDX_REQUIRE(n_frames_block >= 1 && n_frames_block < (int64_t(1) << 30), "n_frames_block out of range");
    MDX_REQUIRE(n_blocks >= 1 && n_blocks <= 65535, "n_blocks out of range");
    MDX_REQUIRE(n_groups >= 1 && n_groups <= 4096, "n_groups out of range");
    MDX_TRY(set_device(dev));
    MDX_TRY(rocfft_ready());
    mdx_msd *h = new mdx_msd();
    h->dev = dev;
    h->t_block = n_frames_block;
    h->n_blocks = n_blocks;
    h->n_groups = n_groups;
    // Any length >= 2 N_t - 1 gives the same linear correlation up to rounding.  The reference pads to
    // 2 * next_fast_len(N_t) (correlation.py:176-178).  Here:
    //   * blocks of more than 200 frames take the shortest length of the engine's own two-pass transform
    //     (mdx_msd_fft.hpp) that covers 2 N_t: 800 = 400 x 2, 1 600 = 400 x 4, 3 200 = 400 x 8, 6 400 = 400 x 16, 2^13, 12 800 = 400 x 32, 2^14, 25 600 = 400 x 64, 2^15, 51 200 = 400 x 128, 2^16,
    //     102 400 = 400 x 256, 204 800 = 400 x 512, 2^18, 409 600 = 400 x 1024, 2^19, 2^20 (2^17 is served by 204 800 / 2^18).  It never materialises the padding and moves
    //     ~4.3 MB per series at 2^18 whatever N_t is, where the rocFFT pipeline moves ~17.7 MB;
    //   * everything else goes through rocFFT at the reference's length, or at the next power of two when that is
    //     at most 1.5 x longer (fewer rocFFT passes: measured 15 % faster end to end at N_t = 1e5).
    // MDX_MSD_ROCFFT=1 (test hook: tests/test_gpu_engines.py compares the two pipelines) keeps rocFFT throughout.
    h->n_fft = 2 * next_fast_len_real(n_frames_block);
    {
        int64_t p = 1;
        while (p < 2 * n_frames_block)
            p <<= 1;
        int64_t own_len = 0;
        if (!getenv("MDX_MSD_ROCFFT")) {
            // the shortest own length that covers 2 N_t
            static const int64_t lengths[] = {400, 800, 1600, 3200, 6400, int64_t(1) << 13, 12800, int64_t(1) << 14, 25600, int64_t(1) << 15, 51200,
                                              int64_t(1) << 16, 102400, 204800, int64_t(1) << 18, 409600, int64_t(1) << 19,
                                              int64_t(1) << 20};
            for (int64_t len : lengths)
                if (len >= 2 * n_frames_block) {
                    own_len = len;
                    break;
                }
        }
        if (own_len && msdfft::shape_for(own_len).r1)
            h->n_fft = own_len;
        else if (2 * p <= 3 * h->n_fft)
            h->n_fft = p;
    }
    h->nc = h->n_fft / 2 + 1;
    h->fft.n_fft = h->n_fft;
    int rc = MDX_OK;
    do {
        if ((rc = stream_acquire(&h->stream)) != MDX_OK) break;
        h->timer.stream = h->stream;
        if ((rc = h->d_acc.ensure(size_t(8) * h->acc_len())) != MDX_OK) break;
        if ((rc = h->d_traj.ensure(size_t(8) * h->traj_len())) != MDX_OK) break;
        // the shapes of mdx_msd_fft.hpp run through the engine's own two-pass kernels (MDX_MSD_ROCFFT=1: rocFFT)
        h->shape = msdfft::shape_for(h->n_fft);
        h->own_fft = h->shape.r1 != 0 && !getenv("MDX_MSD_ROCFFT");
        h->fused_sums = h->own_fft && msdfft::fuses_sums(h->shape);
        // blocks of <= 800 frames: the single-pass kernel (MDX_MSD_TWO_PASS=1, test hook: the two-pass pipeline for
        // 800 and 1 600 points; 400 points has no two-pass form)
        h->single = h->own_fft && msdfft::single_pass(h->shape) && (h->shape.r2 == 1 || !getenv("MDX_MSD_TWO_PASS"));
        if (h->own_fft) {
            const int r1 = h->shape.r1, r2 = h->shape.r2;
            std::vector<double> tw;
            const double two_pi = 6.283185307179586476925286766559;
            auto push = [&](int count, double period) {
                for (int m = 0; m < count; ++m) {
                    tw.push_back(std::cos(two_pi * m / period));
                    tw.push_back(-std::sin(two_pi * m / period));
                }
            };
            push(msdfft::tw_r1_len(r1), r1);   // half table; the whole one for 400 points
            push(r2 / 2, r2);
            push(r2, (double)h->n_fft);
            if ((rc = h->d_tw.ensure(tw.size() * 8)) != MDX_OK) break;
            if (hipMemcpy(h->d_tw.ptr, tw.data(), tw.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
                rc = fail(MDX_ERR_HIP, "twiddle table upload failed");
                break;
            }
            if ((rc = h->d_pfull.ensure(size_t(8) * n_blocks * h->n_fft *
                                        (h->single ? msdfft::single_splits(n_blocks)
                                                   : msdfft::rows_parts(h->shape, n_blocks)))) != MDX_OK) break;
        }
    } while (0);
    if (rc != MDX_OK) {
        mdx_msd_destroy(h);
        return rc;
    }
    *out = h;
    return mdx_msd_reset(h);
}

int mdx_msd_destroy(mdx_msd_t h)
{
    if (!h)
        return MDX_OK;
    (void)hipSetDevice(h->dev);
    if (h->stream)
        (void)hipStreamSynchronize(h->stream);
    h->timer.destroy();
    h->fft.destroy();
    // (the stream has been synchronised: the blocks and the stream go back to the per-device pools)
    for (DeviceBuffer *b : {&h->d_acc, &h->d_traj, &h->d_series, &h->d_spec, &h->d_stage, &h->d_finish,
                            &h->d_inv_in, &h->d_inv_out, &h->d_f32, &h->d_index, &h->d_prev, &h->d_xstart, &h->d_seg,
                            &h->d_image, &h->d_tw, &h->d_pfull, &h->d_part, &h->d_masses, &h->d_com_x,
                            &h->d_shift, &h->d_mol_offsets, &h->d_mol_masses, &h->d_mol_total,
                            &h->d_mol_com, &h->d_cross_f})
        b->recycle();
    if (h->stream)
        stream_release(h->stream);
    delete h;
    return MDX_OK;
}

int mdx_msd_reset(mdx_msd_t h)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipMemsetAsync(h->d_acc.ptr, 0, size_t(8) * h->acc_len(), h->stream));
    MDX_HIP(hipMemsetAsync(h->d_traj.ptr, 0, size_t(8) * h->traj_len(), h->stream));
    MDX_HIP(hipStreamSynchronize(h->stream));
    h->timer.reset();
    h->bytes_moved = 0;
    return MDX_OK;
}

int mdx_msd_n_fft(mdx_msd_t h, int64_t *n_fft)
{
    MDX_REQUIRE(h && n_fft, "NULL argument");
    *n_fft = h->n_fft;
    return MDX_OK;
}

int mdx_msd_transform(mdx_msd_t h, int *own, int *r1, int *r2)
{
    MDX_REQUIRE(h && own && r1 && r2, "NULL argument");
    *own = h->own_fft ? 1 : 0;
    *r1 = h->own_fft ? h->shape.r1 : 0;
    *r2 = h->own_fft ? h->shape.r2 : 0;
    return MDX_OK;
}

int mdx_msd_push_device(mdx_msd_t h, int group, const double *d_pos, int64_t n_total, int64_t first,
                        int64_t count, int zero_dims)
{
    MDX_REQUIRE(h && d_pos, "NULL argument");
    MDX_REQUIRE(group >= 0 && group < h->n_groups, "group %d out of range", group);
    MDX_REQUIRE(first >= 0 && count >= 0 && first + count <= n_total, "particle range out of bounds");
    MDX_REQUIRE(zero_dims >= 0 && zero_dims < 8, "zero_dims is a 3-bit mask");
    MDX_TRY(set_device(h->dev));
    return msd_push_device(h, group, d_pos, n_total, first, count, zero_dims);
}

int mdx_msd_push_device_f32(mdx_msd_t h, int group, const float *d_pos, int64_t n_total, int64_t first,
                            int64_t count, int zero_dims)
{
    MDX_REQUIRE(h && d_pos, "NULL argument");
    MDX_REQUIRE(group >= 0 && group < h->n_groups, "group %d out of range", group);
    MDX_REQUIRE(first >= 0 && count >= 0 && first + count <= n_total, "particle range out of bounds");
    MDX_REQUIRE(zero_dims >= 0 && zero_dims < 8, "zero_dims is a 3-bit mask");
    MDX_TRY(set_device(h->dev));
    return msd_push_device(h, group, nullptr, n_total, first, count, zero_dims, d_pos);
}

int mdx_msd_push(mdx_msd_t h, int group, const double *pos, int64_t n_total, int64_t first,
                 int64_t count, int zero_dims)
{
    MDX_REQUIRE(h && pos, "NULL argument");
    MDX_REQUIRE(group >= 0 && group < h->n_groups, "group %d out of range", group);
    MDX_REQUIRE(first >= 0 && count >= 0 && first + count <= n_total, "particle range out of bounds");
    MDX_REQUIRE(zero_dims >= 0 && zero_dims < 8, "zero_dims is a 3-bit mask");
    MDX_TRY(set_device(h->dev));
    // stage [T][count_chunk][3] slabs: rows are strided on the host (n_total particles per frame)
    const int64_t T = int64_t(h->n_blocks) * h->t_block;
    const int64_t budget = int64_t(2) << 30;
    int64_t chunk = std::max<int64_t>(1, budget / (T * 24));
    chunk = std::min(chunk, count);
    for (int64_t a0 = 0; a0 < count; a0 += chunk) {
        const int64_t c = std::min(chunk, count - a0);
        MDX_TRY(h->d_stage.ensure(size_t(T) * c * 24));
        MDX_TRY(device_stager(h->dev).upload_rows(h->dev, h->stream, h->d_stage.ptr, pos + (first + a0) * 3,
                                                  size_t(c) * 24, size_t(n_total) * 24, (size_t)T));
        MDX_TRY(msd_push_device(h, group, h->d_stage.as<double>(), c, 0, c, zero_dims));
        MDX_HIP(hipStreamSynchronize(h->stream));
    }
    return MDX_OK;
}

int mdx_msd_set_grouping(mdx_msd_t h, int64_t n_molecules, const int64_t *offsets, const double *masses)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    if (n_molecules <= 0 && h->mol_offsets.empty())
        return MDX_OK;               // no grouping before, none now: nothing the queued pushes read changes
    MDX_HIP(hipStreamSynchronize(h->stream));
    h->mol_offsets.clear();
    if (n_molecules <= 0)
        return MDX_OK;
    MDX_REQUIRE(offsets && masses, "NULL argument");
    MDX_REQUIRE(offsets[0] == 0, "offsets must start at 0");
    std::vector<double> total((size_t)n_molecules);
    for (int64_t g = 0; g < n_molecules; ++g) {
        MDX_REQUIRE(offsets[g + 1] > offsets[g], "molecule %lld is empty", (long long)g);
        double m = 0.0;
        for (int64_t a = offsets[g]; a < offsets[g + 1]; ++a)
            m += masses[a];   // sequential, as numpy.bincount sums the weights
        MDX_REQUIRE(m > 0.0, "molecule %lld has no mass", (long long)g);
        total[(size_t)g] = m;
    }
    const int64_t n_atoms = offsets[n_molecules];
    MDX_TRY(h->d_mol_offsets.ensure(size_t(8) * (n_molecules + 1)));
    MDX_TRY(h->d_mol_masses.ensure(size_t(8) * n_atoms));
    MDX_TRY(h->d_mol_total.ensure(size_t(8) * n_molecules));
    MDX_HIP(hipMemcpy(h->d_mol_offsets.ptr, offsets, size_t(8) * (n_molecules + 1), hipMemcpyHostToDevice));
    MDX_HIP(hipMemcpy(h->d_mol_masses.ptr, masses, size_t(8) * n_atoms, hipMemcpyHostToDevice));
    MDX_HIP(hipMemcpy(h->d_mol_total.ptr, total.data(), size_t(8) * n_molecules, hipMemcpyHostToDevice));
    h->mol_offsets.assign(offsets, offsets + n_molecules + 1);
    h->mol_mass = 0.0;
    for (double m : total)
        h->mol_mass += m;
    return MDX_OK;
}

int mdx_msd_system_com_traj(mdx_msd_t h, mdx_traj_t traj, const int64_t *frames, int64_t n_frames,
                            const int32_t *index, int64_t n_index, const double *masses, int unwrap,
                            const double *dims, int wrap, double *out)
{
    MDX_REQUIRE(h && traj && frames && masses && out, "NULL argument");
    MDX_REQUIRE(n_frames >= 0, "negative frame count");
    MDX_REQUIRE((!unwrap && !wrap) || (dims && dims[0] > 0 && dims[1] > 0 && dims[2] > 0),
                "unwrapping / wrapping needs positive box dimensions");
    MDX_TRY(set_device(h->dev));
    Trajectory *t = mdx_traj_internal(traj);
    const int64_t n = index ? n_index : (n_index > 0 ? n_index : t->n_atoms);
    MDX_REQUIRE(n > 0 && (index || n <= t->n_atoms), "bad selection");
    for (int64_t i = 0; index && i < n; ++i)
        if (index[i] < 0 || index[i] >= t->n_atoms)
            return fail(MDX_ERR_INVALID_VALUE, "particle index %d out of range [0, %lld)", index[i],
                        (long long)t->n_atoms);
    if (n_frames == 0)
        return MDX_OK;
    TrajFrames src(t, frames, index);
    return msd_system_com_frames(h, src, n_frames, n, masses, unwrap, dims, wrap, out);
}

int mdx_msd_system_com_f32(mdx_msd_t h, const float *pos, int64_t n_frames, int64_t n_sel,
                           const double *masses, int unwrap, const double *dims, int wrap, double *out)
{
    MDX_REQUIRE(h && pos && masses && out, "NULL argument");
    MDX_REQUIRE(n_frames >= 0 && n_sel > 0, "bad size");
    MDX_REQUIRE((!unwrap && !wrap) || (dims && dims[0] > 0 && dims[1] > 0 && dims[2] > 0),
                "unwrapping / wrapping needs positive box dimensions");
    MDX_TRY(set_device(h->dev));
    if (n_frames == 0)
        return MDX_OK;
    HostFrames<float> src(pos, n_sel);
    return msd_system_com_frames(h, src, n_frames, n_sel, masses, unwrap, dims, wrap, out);
}

int mdx_msd_system_com_f64(mdx_msd_t h, const double *pos, int64_t n_frames, int64_t n_sel,
                           const double *masses, int unwrap, const double *dims, int wrap, double *out)
{
    MDX_REQUIRE(h && pos && masses && out, "NULL argument");
    MDX_REQUIRE(n_frames >= 0 && n_sel > 0, "bad size");
    MDX_REQUIRE((!unwrap && !wrap) || (dims && dims[0] > 0 && dims[1] > 0 && dims[2] > 0),
                "unwrapping / wrapping needs positive box dimensions");
    MDX_TRY(set_device(h->dev));
    if (n_frames == 0)
        return MDX_OK;
    HostFrames<double> src(pos, n_sel);
    return msd_system_com_frames(h, src, n_frames, n_sel, masses, unwrap, dims, wrap, out);
}

int mdx_msd_set_initial_images(mdx_msd_t h, const int32_t *images, int64_t n_sel)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_REQUIRE(n_sel >= 0 && (n_sel == 0 || images), "bad image flags");
    MDX_TRY(set_device(h->dev));
    // an earlier call's copy out of the vector may still be queued
    MDX_HIP(hipStreamSynchronize(h->stream));
    h->images0.assign(images, images + 3 * n_sel);
    return MDX_OK;
}

int mdx_msd_push_f32(mdx_msd_t h, int group, const float *pos, int64_t n_frames, int64_t n_sel,
                     int unwrap, const double *dims, int zero_dims, const double *shift)
{
    MDX_REQUIRE(h && pos, "NULL argument");
    MDX_REQUIRE(group >= 0 && group < h->n_groups, "group %d out of range", group);
    MDX_REQUIRE(zero_dims >= 0 && zero_dims < 8, "zero_dims is a 3-bit mask");
    const int64_t T = int64_t(h->n_blocks) * h->t_block;
    MDX_REQUIRE(n_frames >= T, "%lld frames given, the engine needs %lld", (long long)n_frames,
                (long long)T);
    MDX_REQUIRE(!unwrap || (dims && dims[0] > 0 && dims[1] > 0 && dims[2] > 0),
                "unwrapping needs positive box dimensions");
    MDX_REQUIRE(n_sel >= 0, "negative particle count");
    MDX_TRY(set_device(h->dev));
    if (n_sel == 0)
        return MDX_OK;
    HostFrames<float> src(pos, n_sel);
    return msd_push_frames(h, group, src, n_sel, unwrap, dims, zero_dims, shift);
}

int mdx_msd_push_f64(mdx_msd_t h, int group, const double *pos, int64_t n_frames, int64_t n_sel,
                     int unwrap, const double *dims, int zero_dims, const double *shift)
{
    MDX_REQUIRE(h && pos, "NULL argument");
    MDX_REQUIRE(group >= 0 && group < h->n_groups, "group %d out of range", group);
    MDX_REQUIRE(zero_dims >= 0 && zero_dims < 8, "zero_dims is a 3-bit mask");
    const int64_t T = int64_t(h->n_blocks) * h->t_block;
    MDX_REQUIRE(n_frames >= T, "%lld frames given, the engine needs %lld", (long long)n_frames,
                (long long)T);
    MDX_REQUIRE(!unwrap || (dims && dims[0] > 0 && dims[1] > 0 && dims[2] > 0),
                "unwrapping needs positive box dimensions");
    MDX_REQUIRE(n_sel >= 0, "negative particle count");
    MDX_TRY(set_device(h->dev));
    if (n_sel == 0)
        return MDX_OK;
    HostFrames<double> src(pos, n_sel);
    return msd_push_frames(h, group, src, n_sel, unwrap, dims, zero_dims, shift);
}

int mdx_msd_push_frames_device(mdx_msd_t h, int group, const void *d_pos, int elem_bytes, int64_t n_frames,
                               int64_t n_total, const int32_t *index, int64_t n_index, int unwrap,
                               const double *dims, int zero_dims, const double *shift)
{
    MDX_REQUIRE(h && d_pos, "NULL argument");
    MDX_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "elem_bytes is 4 (float32) or 8 (float64)");
    MDX_REQUIRE(group >= 0 && group < h->n_groups, "group %d out of range", group);
    MDX_REQUIRE(zero_dims >= 0 && zero_dims < 8, "zero_dims is a 3-bit mask");
    const int64_t T = int64_t(h->n_blocks) * h->t_block;
    MDX_REQUIRE(n_frames >= T, "%lld frames given, the engine needs %lld", (long long)n_frames,
                (long long)T);
    MDX_REQUIRE(!unwrap || (dims && dims[0] > 0 && dims[1] > 0 && dims[2] > 0),
                "unwrapping needs positive box dimensions");
    MDX_REQUIRE(n_total > 0 && n_total < (int64_t(1) << 31) / 3, "n_total out of range");
    const int64_t n = index ? n_index : (n_index > 0 ? n_index : n_total);
    MDX_REQUIRE(n >= 0 && (index || n <= n_total), "selection larger than the frames");
    for (int64_t i = 0; index && i < n; ++i)
        if (index[i] < 0 || index[i] >= n_total)
            return fail(MDX_ERR_INVALID_VALUE, "particle index %d out of range [0, %lld)", index[i],
                        (long long)n_total);
    MDX_TRY(set_device(h->dev));
    if (n == 0)
        return MDX_OK;
    DeviceFrames src(d_pos, elem_bytes, n_total, index);
    return msd_push_frames(h, group, src, n, unwrap, dims, zero_dims, shift);
}

int mdx_msd_system_com_device(mdx_msd_t h, const void *d_pos, int elem_bytes, int64_t n_frames,
                              int64_t n_total, const int32_t *index, int64_t n_index, const double *masses,
                              int unwrap, const double *dims, int wrap, double *out)
{
    MDX_REQUIRE(h && d_pos && masses && out, "NULL argument");
    MDX_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "elem_bytes is 4 (float32) or 8 (float64)");
    MDX_REQUIRE(n_frames >= 0, "negative frame count");
    MDX_REQUIRE((!unwrap && !wrap) || (dims && dims[0] > 0 && dims[1] > 0 && dims[2] > 0),
                "unwrapping / wrapping needs positive box dimensions");
    MDX_REQUIRE(n_total > 0 && n_total < (int64_t(1) << 31) / 3, "n_total out of range");
    const int64_t n = index ? n_index : (n_index > 0 ? n_index : n_total);
    MDX_REQUIRE(n > 0 && (index || n <= n_total), "bad selection");
    for (int64_t i = 0; index && i < n; ++i)
        if (index[i] < 0 || index[i] >= n_total)
            return fail(MDX_ERR_INVALID_VALUE, "particle index %d out of range [0, %lld)", index[i],
                        (long long)n_total);
    MDX_TRY(set_device(h->dev));
    if (n_frames == 0)
        return MDX_OK;
    DeviceFrames src(d_pos, elem_bytes, n_total, index);
    return msd_system_com_frames(h, src, n_frames, n, masses, unwrap, dims, wrap, out);
}

int mdx_msd_push_traj(mdx_msd_t h, int group, mdx_traj_t traj, const int64_t *frames,
                      int64_t n_frames, const int32_t *index, int64_t n_index, int unwrap,
                      const double *dims, int zero_dims, const double *shift)
{
    MDX_REQUIRE(h && traj && frames, "NULL argument");
    MDX_REQUIRE(group >= 0 && group < h->n_groups, "group %d out of range", group);
    MDX_REQUIRE(zero_dims >= 0 && zero_dims < 8, "zero_dims is a 3-bit mask");
    const int64_t T = int64_t(h->n_blocks) * h->t_block;
    MDX_REQUIRE(n_frames >= T, "%lld frames listed, the engine needs %lld", (long long)n_frames,
                (long long)T);
    MDX_REQUIRE(!unwrap || (dims && dims[0] > 0 && dims[1] > 0 && dims[2] > 0),
                "unwrapping needs positive box dimensions");
    MDX_TRY(set_device(h->dev));
    Trajectory *t = mdx_traj_internal(traj);
    const int64_t n = index ? n_index : (n_index > 0 ? n_index : t->n_atoms);
    MDX_REQUIRE(index || n <= t->n_atoms, "selection larger than the trajectory");
    if (n == 0)
        return MDX_OK;
    for (int64_t i = 0; index && i < n; ++i)
        if (index[i] < 0 || index[i] >= t->n_atoms)
            return fail(MDX_ERR_INVALID_VALUE, "particle index %d out of range [0, %lld)", index[i],
                        (long long)t->n_atoms);
    TrajFrames src(t, frames, index);
    return msd_push_frames(h, group, src, n, unwrap, dims, zero_dims, shift);
}

static int msd_result(mdx_msd_t h, double *msd_self_sum, double *acf_sum, double *sum_traj);

int mdx_msd_result(mdx_msd_t h, double *msd_self_sum, double *sum_traj)
{
    MDX_REQUIRE(h, "NULL handle");
    return msd_result(h, msd_self_sum, nullptr, sum_traj);
}

int mdx_msd_result_acf(mdx_msd_t h, double *acf_sum)
{
    MDX_REQUIRE(h && acf_sum, "NULL argument");
    return msd_result(h, nullptr, acf_sum, nullptr);
}

// Cross displacements of the groups' summed trajectories, MSD_m = S_m - 2 A_m with r1 = sum of group i, r2 = sum of
// group j (reference correlation.py:621-668 as transport.py:1034, 1052 call it): everything from the
// accumulators that are already in HBM — one batched forward transform of the G * B * 3 series, one spectrum
// product and one inverse per (pair, block), the S_m recurrence on the host.
int mdx_msd_cross(mdx_msd_t h, const int32_t *pairs, int64_t n_pairs, double *out)
{
    MDX_REQUIRE(h && pairs && out, "NULL argument");
    MDX_REQUIRE(n_pairs >= 1 && n_pairs < (int64_t(1) << 28), "pair count out of range");
    for (int64_t p = 0; p < 2 * n_pairs; ++p)
        MDX_REQUIRE(pairs[p] >= 0 && pairs[p] < h->n_groups, "group %d out of range", pairs[p]);
    MDX_TRY(set_device(h->dev));
    const int B = h->n_blocks;
    const int64_t Tb = h->t_block, GB = int64_t(h->n_groups) * B, PB = n_pairs * B;
    // (pair, block) rows go through the products and the inverse transforms in batches: the grid's y extent
    // (65 535) and ~1 GiB of inverse-transform buffers bound a batch, not the number of pairs (36 groups at 100
    // blocks, 12 at 1 000 used to be refused)
    const int64_t rows_max = std::max<int64_t>(1, std::min<int64_t>(65535, (int64_t(1) << 30) / (24 * h->n_fft)));
    const int64_t rows_batch = std::min(PB, rows_max);
    MDX_TRY(h->d_series.ensure(size_t(8) * GB * 3 * h->n_fft));
    MDX_TRY(h->d_inv_in.ensure(size_t(16) * std::max(GB * 3, rows_batch) * h->nc));
    MDX_TRY(h->d_inv_out.ensure(size_t(8) * std::max(GB, rows_batch) * h->n_fft));
    MDX_TRY(h->d_cross_f.ensure(size_t(16) * GB * 3 * h->nc));
    MDX_TRY(h->d_index.ensure(size_t(8) * n_pairs));
    MDX_HIP(hipMemcpyAsync(h->d_index.ptr, pairs, size_t(8) * n_pairs, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(cross_pad_kernel, dim3((unsigned)ceil_div(GB * 3 * h->n_fft, 256)), dim3(256), 0, h->stream,
                       h->d_traj.as<double>(), GB, Tb, h->n_fft, h->d_series.as<double>());
    MDX_TRY(h->fft.exec(0, GB * 3, h->d_series.ptr, h->d_cross_f.ptr, h->stream));
    // per batch: spectrum products, inverse transforms, the per-frame products r_i . r_j and the S_m recurrence —
    // all on the device (the host used to copy the correlations and the summed trajectories back through the
    // runtime's pageable path and walk them); the finished rows come back in one copy through the pinned ring
    MDX_TRY(h->d_finish.ensure(size_t(8) * PB * Tb));
    MDX_TRY(h->d_stage.ensure(size_t(8) * rows_batch * Tb));
    const double inv_n = 1.0 / double(h->n_fft);
    for (int64_t r0 = 0; r0 < PB; r0 += rows_batch) {
        const int64_t nr = std::min(rows_batch, PB - r0);
        hipLaunchKernelGGL(cross_product_kernel, dim3((unsigned)ceil_div(h->nc, 256), (unsigned)nr), dim3(256), 0,
                           h->stream, h->d_cross_f.as<double2>(), h->d_index.as<int>(), B, h->nc, r0,
                           h->d_inv_in.as<double2>());
        MDX_TRY(h->fft.exec(1, nr, h->d_inv_in.ptr, h->d_inv_out.ptr, h->stream));
        hipLaunchKernelGGL(cross_dot_kernel, dim3((unsigned)ceil_div(Tb, 256), (unsigned)nr), dim3(256), 0, h->stream,
                           h->d_traj.as<double>(), h->d_index.as<int>(), B, Tb, r0, h->d_stage.as<double>());
        hipLaunchKernelGGL(msd_finish_kernel, dim3((unsigned)nr), dim3(FINISH_THREADS), 0, h->stream,
                           h->d_inv_out.as<double>(), h->n_fft, h->d_stage.as<double>(), Tb, inv_n,
                           h->d_finish.as<double>() + r0 * Tb, 1.0);
        MDX_HIP(hipGetLastError());
    }
    MDX_TRY(device_stager(h->dev).download(h->dev, h->stream, out, h->d_finish.ptr, size_t(8) * PB * Tb));
    return MDX_OK;
}

static int msd_result(mdx_msd_t h, double *msd_self_sum, double *acf_sum, double *sum_traj)
{
    MDX_TRY(set_device(h->dev));
    const int64_t GB = int64_t(h->n_groups) * h->n_blocks;
    const int64_t Tb = h->t_block;
    if (sum_traj) {
        MDX_HIP(hipStreamSynchronize(h->stream));
        MDX_TRY(mdx_memcpy_d2h(h->dev, sum_traj, h->d_traj.ptr, size_t(8) * h->traj_len()));
    }
    if (!msd_self_sum && !acf_sum) {
        h->timer.collect();
        return MDX_OK;
    }
    MDX_TRY(h->d_inv_in.ensure(size_t(16) * GB * h->nc));
    MDX_TRY(h->d_inv_out.ensure(size_t(8) * GB * h->n_fft));
    hipEvent_t ev = h->timer.begin();
    hipLaunchKernelGGL(msd_real_to_complex_kernel, dim3((unsigned)ceil_div(GB * h->nc, 256)),
                       dim3(256), 0, h->stream, h->d_acc.as<double>(), GB * h->nc,
                       h->d_inv_in.as<double2>());
    MDX_TRY(h->fft.exec(1, GB, h->d_inv_in.ptr, h->d_inv_out.ptr, h->stream));
    h->timer.end(ev);
    const double inv_n = 1.0 / double(h->n_fft);
    if (!acf_sum) {
        // the S_m recurrence on the device; the finished rows come back in one copy through the pinned ring
        MDX_TRY(h->d_finish.ensure(size_t(8) * GB * Tb));
        hipEvent_t ev2 = h->timer.begin();
        hipLaunchKernelGGL(msd_finish_kernel, dim3((unsigned)GB), dim3(FINISH_THREADS), 0, h->stream,
                           h->d_inv_out.as<double>(), h->n_fft, h->dsq(0), Tb, inv_n, h->d_finish.as<double>());
        h->timer.end(ev2);
        MDX_HIP(hipGetLastError());
        MDX_TRY(device_stager(h->dev).download(h->dev, h->stream, msd_self_sum, h->d_finish.ptr, size_t(8) * GB * Tb));
        h->timer.collect();
        return MDX_OK;
    }
    // (mdx_msd_result_acf) sum over particles and dimensions of sum_k x(k) x(k+m), not normalised: the live lags of
    // every row are packed on the device and come back in one copy through the pinned ring, like the finished rows
    MDX_TRY(h->d_finish.ensure(size_t(8) * GB * Tb));
    MDX_HIP(hipMemcpy2DAsync(h->d_finish.ptr, size_t(Tb) * 8, h->d_inv_out.ptr, size_t(h->n_fft) * 8, size_t(Tb) * 8,
                             (size_t)GB, hipMemcpyDeviceToDevice, h->stream));
    MDX_TRY(device_stager(h->dev).download(h->dev, h->stream, acf_sum, h->d_finish.ptr, size_t(8) * GB * Tb));
    h->timer.collect();
    for (int64_t i = 0; i < GB * Tb; ++i)
        acf_sum[i] *= inv_n;
    return MDX_OK;
}

int mdx_msd_stats(mdx_msd_t h, int64_t *launches, double *kernel_ms, int64_t *bytes_moved)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    h->timer.collect();
    if (launches) *launches = h->timer.launches;
    if (kernel_ms) *kernel_ms = h->timer.total_ms;
    if (bytes_moved) *bytes_moved = h->bytes_moved;
    return MDX_OK;
}

int mdx_msd_enable_timing(mdx_msd_t h, int on)
{
    MDX_REQUIRE(h, "NULL handle");
    h->timer.enabled = on != 0;
    return MDX_OK;
}

int mdx_msd_internal_buffers(mdx_msd_t h, double **d_a, int64_t *na, double **d_b, int64_t *nb,
                             hipStream_t *stream)
{
    MDX_TRY(set_device(h->dev));
    *d_a = h->d_acc.as<double>();
    *na = h->acc_len();
    *d_b = h->d_traj.as<double>();
    *nb = h->traj_len();
    *stream = h->stream;
    return MDX_OK;
}

int mdx_correlate(int dev, const double *a, const double *b, int64_t n_series, int64_t n_t,
                  double *out, double *neg_out)
{
    MDX_REQUIRE(a && out, "NULL argument");
    MDX_REQUIRE(n_series >= 1 && n_t >= 1 && n_t < (int64_t(1) << 30), "bad size");
    MDX_TRY(set_device(dev));
    MDX_TRY(rocfft_ready());
    const int64_t n_fft = 2 * next_fast_len_real(n_t);
    const int64_t nc = n_fft / 2 + 1;
    const int64_t per_series = n_fft * 8 * 2 + nc * 16 * 2 + n_t * 8 * 2;
    int64_t chunk = std::max<int64_t>(1, (int64_t(3) << 30) / per_series);
    chunk = std::min(chunk, n_series);
    FftCache fft;
    fft.n_fft = n_fft;
    // everything of the call — uploads, padding, transforms, products, the copy back — is ordered on ONE stream;
    // the two inputs have their own device buffers (`b` used to be uploaded into the buffer the
    // padding kernel of `a` might still be reading, on another stream)
    DeviceBuffer d_a, d_b, d_pad, d_fa, d_fb;
    std::vector<double> host(size_t(chunk) * n_fft);
    hipStream_t stream = nullptr;
    int rc = MDX_OK;
    auto run = [&]() -> int {
        MDX_TRY(stream_acquire(&stream));
        MDX_TRY(d_a.ensure(size_t(chunk) * n_t * 8));
        MDX_TRY(d_pad.ensure(size_t(chunk) * n_fft * 8));
        MDX_TRY(d_fa.ensure(size_t(chunk) * nc * 16));
        if (b) {
            MDX_TRY(d_b.ensure(size_t(chunk) * n_t * 8));
            MDX_TRY(d_fb.ensure(size_t(chunk) * nc * 16));
        }
        HostStager &ring = device_stager(dev);
        for (int64_t s0 = 0; s0 < n_series; s0 += chunk) {
            const int64_t c = std::min(chunk, n_series - s0);
            const unsigned gp = (unsigned)ceil_div(c * n_fft, 256);
            MDX_TRY(ring.upload(dev, stream, d_a.ptr, a + s0 * n_t, size_t(c) * n_t * 8));
            if (b)
                MDX_TRY(ring.upload(dev, stream, d_b.ptr, b + s0 * n_t, size_t(c) * n_t * 8));
            hipLaunchKernelGGL(corr_pad_kernel, dim3(gp), dim3(256), 0, stream, d_a.as<double>(), c, n_t,
                               n_fft, d_pad.as<double>());
            MDX_TRY(fft.exec(0, c, d_pad.ptr, d_fa.ptr, stream));
            if (b) {
                hipLaunchKernelGGL(corr_pad_kernel, dim3(gp), dim3(256), 0, stream, d_b.as<double>(), c,
                                   n_t, n_fft, d_pad.as<double>());
                MDX_TRY(fft.exec(0, c, d_pad.ptr, d_fb.ptr, stream));
            }
            hipLaunchKernelGGL(corr_product_kernel, dim3((unsigned)ceil_div(c * nc, 256)), dim3(256),
                               0, stream, d_fa.as<double2>(), b ? d_fb.as<double2>() : nullptr, c * nc,
                               d_fa.as<double2>());
            MDX_TRY(fft.exec(1, c, d_fa.ptr, d_pad.ptr, stream));
            MDX_HIP(hipGetLastError());
            // (returns when the bytes are in `host`: the ring's copy stream waits for `stream`, the call for the ring)
            MDX_TRY(ring.download(dev, stream, host.data(), d_pad.ptr, size_t(c) * n_fft * 8));
            const double inv_n = 1.0 / double(n_fft);
            for (int64_t s = 0; s < c; ++s) {
                const double *r = host.data() + s * n_fft;
                double *o = out + (s0 + s) * n_t;
                for (int64_t m = 0; m < n_t; ++m)
                    o[m] = r[m] * inv_n;
                if (neg_out) {
                    double *q = neg_out + (s0 + s) * n_t;
                    q[0] = r[0] * inv_n;
                    for (int64_t m = 1; m < n_t; ++m)
                        q[m] = r[n_fft - m] * inv_n;
                }
            }
        }
        return MDX_OK;
    };
    rc = run();
    (void)hipDeviceSynchronize();
    if (stream)
        stream_release(stream);
    fft.destroy();
    d_a.recycle();
    d_b.recycle();
    d_pad.recycle();
    d_fa.recycle();
    d_fb.recycle();
    return rc;
}

}  // extern "C"
