// mdx_sq.hip — static / partial structure factor on gfx950 (MI355X).
//
// Carries StructureFactor._single_frame (reference
// src/mdhelper/analysis/structure.py:1481-1527) and the Numba kernels under it
// (src/mdhelper/algorithm/accelerated.py:81-165 delta_fourier_transform_sum*,
// :167-247 inner*, :249-321 pythagorean_trigonometric_identity*):
//
//     rho_g(q) = sum_{j in group g} exp(i q . r_j)                       (fp64)
//     ssf[p](q) += |rho_j|^2            for a pair p = (j, j)
//               += 2 Re(rho_j rho_k^*)  for a pair p = (j, k), j != k
//               += |sum_g rho_g|^2      for mode=None, p = (None, None)
//
// The reference's form="exp" and form="trig" are the same numbers (cos/sin sums);
// one fused kernel serves both and the N_q x N table of q.r that the "trig" form
// materialises (134 MB at 512 x 32768) never exists.
//
// Kernel A (sq_rho_kernel): one thread owns QPT wavevectors in registers, the
// group's particles stream through LDS and are read by broadcast; phase q.r and
// sincos are fp64 (a float32 phase is off by ~1e-5 rad at |q.r| ~ 200, which
// breaks the 1e-6 parity bar).  Bound: fp64 VALU (≈40 fp64 instr per (q, r)).
// Kernel B (sq_pair_kernel): sums the particle splits in a fixed order, forms the
// pair products and accumulates over the frames of the slab in frame order, so
// results are run-to-run reproducible.
#include "mdx_common.hpp"
#include "mdx_internal.hpp"
#include "mdx_molecules.hpp"

using namespace mdx;

#include "mdx_sq_device.hpp"
#include "mdx_traj.hpp"

using namespace mdx_sq_dev;

namespace {

constexpr int PAIR_Q = 64, PAIR_FS = 16;   // wavevectors x interleaved frame subsets per block

__global__ __launch_bounds__(PAIR_Q * PAIR_FS) void sq_pair_kernel(
    const double2 *__restrict__ rho, int n_frames, int n_groups, int n_split, int n_q,
    const int *__restrict__ pairs, int n_pairs, double *__restrict__ acc)
{
    __shared__ double part[PAIR_FS][PAIR_Q];
    const int lane = threadIdx.x % PAIR_Q, fs = threadIdx.x / PAIR_Q;
    const int qi = blockIdx.x * PAIR_Q + lane;
    const int p = blockIdx.y;
    const int j = pairs[2 * p], k = pairs[2 * p + 1];
    double sum = 0.0;
    for (int f = fs; f < n_frames && qi < n_q; f += PAIR_FS) {
        const double2 *R = rho + int64_t(f) * n_groups * n_split * n_q;
        auto group_rho = [&](int g) {
            double2 r = make_double2(0.0, 0.0);
            for (int s = 0; s < n_split; ++s) {
                double2 v = R[(int64_t(g) * n_split + s) * n_q + qi];
                r.x += v.x;
                r.y += v.y;
            }
            return r;
        };
        if (j < 0) {
            double2 r = make_double2(0.0, 0.0);
            for (int g = 0; g < n_groups; ++g) {
                double2 v = group_rho(g);
                r.x += v.x;
                r.y += v.y;
            }
            sum += r.x * r.x + r.y * r.y;
        } else if (j == k) {
            double2 r = group_rho(j);
            sum += r.x * r.x + r.y * r.y;
        } else {
            double2 a = group_rho(j), b = group_rho(k);
            sum += 2.0 * (a.x * b.x + a.y * b.y);
        }
    }
    part[fs][lane] = sum;
    __syncthreads();
    if (fs == 0 && qi < n_q) {
        double total = 0.0;   // fixed order: the result does not depend on scheduling
        for (int s = 0; s < PAIR_FS; ++s)
            total += part[s][lane];
        acc[int64_t(p) * n_q + qi] += total;
    }
}

// float64 positions -> F[q] for the function-level drop-in (one pseudo-frame, one group)
__global__ __launch_bounds__(SQ_THREADS) void sq_fourier_sum_f64_kernel(
    const double *__restrict__ pos, int64_t n, const double *__restrict__ qv, int n_q, int n_split,
    double2 *__restrict__ part)
{
    __shared__ double sx[512], sy[512], sz[512];
    const int tid = threadIdx.x;
    const int qi = blockIdx.x * SQ_THREADS + tid;
    const int sp = blockIdx.y;
    const bool ok = qi < n_q;
    const double q0 = ok ? qv[3 * int64_t(qi)] : 0.0, q1 = ok ? qv[3 * int64_t(qi) + 1] : 0.0,
                 q2 = ok ? qv[3 * int64_t(qi) + 2] : 0.0;
    const int64_t per = (n + n_split - 1) / n_split;
    const int64_t lo = sp * per, hi = min(n, lo + per);
    double ac = 0.0, as = 0.0;
    for (int64_t base = lo; base < hi; base += 512) {
        const int cnt = (int)min<int64_t>(512, hi - base);
        __syncthreads();
        for (int e = tid; e < cnt * 3; e += SQ_THREADS) {
            double v = pos[base * 3 + e];
            int a = e / 3, k = e - 3 * a;
            (k == 0 ? sx : k == 1 ? sy : sz)[a] = v;
        }
        __syncthreads();
        for (int a = 0; a < cnt; ++a) {
            double ph = fma(q2, sz[a], fma(q1, sy[a], q0 * sx[a]));
            double s, c;
            sincos_f64(ph, s, c);
            ac += c;
            as += s;
        }
    }
    if (ok)
        part[int64_t(sp) * n_q + qi] = make_double2(ac, as);
}


// Row sums of cos / sin of a caller-supplied matrix x[n_rows][n_cols] (the `q . r` arrays the reference's
// trigonometric forms and its ISF take: accelerated.py:249-321, :323-627; structure.py:1238-1317).
// HBM-streaming: one block walks a column segment of one row with coalesced 8-byte loads, one float64 sincos per
// element; the partial (sum cos, sum sin) of every segment is written once — no atomics, fixed summation order.
constexpr int TRIG_THREADS = 256;
__global__ __launch_bounds__(TRIG_THREADS) void trig_rowsums_kernel(const double *__restrict__ x, int64_t n_cols,
                                                                    int n_split, double2 *__restrict__ part)
{
    __shared__ double2 red[TRIG_THREADS / 64];
    const int64_t row = blockIdx.y;
    const int sp = blockIdx.x;
    const int64_t per = ((n_cols + n_split - 1) / n_split + TRIG_THREADS - 1) / TRIG_THREADS * TRIG_THREADS;
    const int64_t lo = sp * per, hi = min(n_cols, lo + per);
    const double *__restrict__ xr = x + row * n_cols;
    double ac = 0.0, as = 0.0;
    for (int64_t c = lo + threadIdx.x; c < hi; c += TRIG_THREADS) {
        double sn, cs;
        const double v = xr[c];
        if (fabs(v) < 1.0e8)
            sincos_f64(v, sn, cs);
        else                      // caller-supplied phases of any size: the library's full-range reduction
            sincos(v, &sn, &cs);
        ac += cs;
        as += sn;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        ac += __shfl_down(ac, off, 64);
        as += __shfl_down(as, off, 64);
    }
    if ((threadIdx.x & 63) == 0)
        red[threadIdx.x >> 6] = make_double2(ac, as);
    __syncthreads();
    if (threadIdx.x == 0) {
        double2 t = red[0];
        for (int w = 1; w < TRIG_THREADS / 64; ++w) {
            t.x += red[w].x;
            t.y += red[w].y;
        }
        part[row * n_split + sp] = t;
    }
}

__global__ void trig_fold_kernel(const double2 *__restrict__ part, int64_t n_rows, int n_split,
                                 double *__restrict__ cos_out, double *__restrict__ sin_out)
{
    const int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (row >= n_rows)
        return;
    double c = 0.0, s = 0.0;
    for (int k = 0; k < n_split; ++k) {
        c += part[row * n_split + k].x;
        s += part[row * n_split + k].y;
    }
    if (cos_out) cos_out[row] = c;
    if (sin_out) sin_out[row] = s;
}

// out[i][j] = q_i . r_j (accelerated.py:167-247: a[0] b[0] + a[1] b[1] + a[2] b[2], in that order)
__global__ void inner_kernel(const double *__restrict__ q, int64_t n_q, const double *__restrict__ r, int64_t n,
                             double *__restrict__ out)
{
    const int64_t j = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= n)
        return;
    {
#pragma clang fp contract(off)   // the reference's three products and two sums, no FMA
        const double q0 = q[3 * i], q1 = q[3 * i + 1], q2 = q[3 * i + 2];
        out[i * n + j] = (q0 * r[3 * j] + q1 * r[3 * j + 1]) + q2 * r[3 * j + 2];
    }
}

}  // namespace

struct mdx_sq {
    int dev = 0;
    hipStream_t stream = nullptr;
    int64_t n_q = 0;
    int n_groups = 0, n_pairs = 0;
    int64_t n_total = 0;
    std::vector<int64_t> offsets;
    DeviceBuffer d_q, d_offsets, d_pairs, d_acc, d_rho, d_stage[2], d_index, d_mtrip;
    StagePipeline pipe;   // host-buffer / trajectory-file entry points
    MoleculeStage mol;    // optional centre-of-mass stage (groupings other than "atoms")
    StreamTimer timer;
    bool lattice = false;        // wavevectors are integer multiples of one base per axis
    SqLattice lat{};
    size_t lat_lds = 0;
    // column form of the lattice path (mdx_sq_device.hpp): items, block size, its own tile
    bool quads = false;          // register-blocked columns (sq_rho_quads_kernel)
    int n_qitems = 0, ipb = 1, n_sub = 1;
    SqLattice quad_lat{};
    size_t quad_lds = 0;
    int quad_regular = 0;        // row stride of the regular-item form of the quad kernel, 0: general items
    DeviceBuffer d_qitems;
    bool columns = false;
    int n_items = 0, col_threads = 0;
    SqLattice col_lat{};
    size_t col_lds = 0;
    DeviceBuffer d_items;
};

static int sq_accumulate_points(mdx_sq *h, const float *d_pos, int64_t n, int64_t n_frames);

// rows -> points: with a grouping set the incoming rows are particles sorted molecule by
// molecule and the Fourier sums run over the float32 centres of mass
static int sq_accumulate_device(mdx_sq *h, const float *d_pos, int64_t n, int64_t n_frames)
{
    if (!h->mol.active() || n_frames == 0)
        return sq_accumulate_points(h, d_pos, n, n_frames);
    MDX_REQUIRE(n == h->mol.n_atoms, "%lld particles given, the grouping was defined for %lld",
                (long long)n, (long long)h->mol.n_atoms);
    const float *centres = nullptr;
    MDX_TRY(h->mol.run(h->stream, d_pos, n_frames, nullptr, &centres));
    return sq_accumulate_points(h, centres, h->mol.n_groups, n_frames);
}

static int sq_accumulate_points(mdx_sq *h, const float *d_pos, int64_t n, int64_t n_frames)
{
    if (n_frames == 0)
        return MDX_OK;
    MDX_REQUIRE(n >= h->n_total, "positions hold %lld particles but the groups span %lld",
                (long long)n, (long long)h->n_total);
    const int qblocks = h->quads     ? (int)ceil_div(int64_t(h->n_qitems), int64_t(h->ipb))
                        : h->columns ? (int)ceil_div(h->n_items, h->col_threads)
                                     : (int)ceil_div(h->n_q, SQ_QPB);
    // split the particles when frames x q-blocks x groups alone would not fill 256 CUs
    int64_t max_group = 0;
    for (int g = 0; g < h->n_groups; ++g)
        max_group = std::max(max_group, h->offsets[g + 1] - h->offsets[g]);
    // ... counted in waves: 256 CUs x 4 SIMDs x ~8 resident waves
    const int block_waves = (h->quads ? SQ_QUAD_THREADS : h->columns ? h->col_threads : SQ_THREADS) / 64;
    constexpr int64_t want_waves = 4096;
    int n_split = 1;
    while (int64_t(qblocks) * block_waves * h->n_groups * n_split * std::min<int64_t>(n_frames, 4096) <
               want_waves &&
           n_split < 64 && max_group / (n_split * 2) >= 2 * SQ_TILE)
        n_split *= 2;
    const int64_t rho_per_frame = int64_t(h->n_groups) * n_split * h->n_q * 16;
    int64_t slab = std::max<int64_t>(1, (int64_t(512) << 20) / rho_per_frame);
    slab = std::min<int64_t>(std::min<int64_t>(slab, 32768), n_frames);
    MDX_TRY(h->d_rho.ensure(size_t(rho_per_frame) * slab));
    MDX_REQUIRE(int64_t(h->n_groups) * n_split <= 65535, "too many groups");
    hipEvent_t ev = h->timer.begin();
    for (int64_t f0 = 0; f0 < n_frames; f0 += slab) {
        const int64_t nf = std::min(slab, n_frames - f0);
        if (h->quads)
            hipLaunchKernelGGL(sq_rho_quads_pick(h->quad_regular), dim3(qblocks, h->n_groups * n_split, (unsigned)nf),
                               dim3(SQ_QUAD_THREADS), h->quad_lds, h->stream, d_pos + f0 * n * 3, n,
                               h->d_qitems.as<SqQuadItem>(), h->n_qitems, h->ipb, h->n_sub,
                               (int)h->n_q, h->quad_lat, h->d_offsets.as<int64_t>(), h->n_groups,
                               n_split, h->d_rho.as<double2>());
        else if (h->columns)
            hipLaunchKernelGGL(sq_rho_columns_kernel, dim3(qblocks, h->n_groups * n_split, (unsigned)nf),
                               dim3(h->col_threads), h->col_lds, h->stream, d_pos + f0 * n * 3, n,
                               h->d_items.as<SqColumnItem>(), h->n_items, (int)h->n_q, h->col_lat,
                               h->d_offsets.as<int64_t>(), h->n_groups, n_split,
                               h->d_rho.as<double2>());
        else if (h->lattice)
            hipLaunchKernelGGL(sq_rho_lattice_kernel, dim3(qblocks, h->n_groups * n_split, (unsigned)nf),
                               dim3(SQ_THREADS), h->lat_lds, h->stream, d_pos + f0 * n * 3, n,
                               h->d_mtrip.as<short4>(), (int)h->n_q, h->lat,
                               h->d_offsets.as<int64_t>(), h->n_groups, n_split,
                               h->d_rho.as<double2>());
        else
            hipLaunchKernelGGL(sq_rho_kernel, dim3(qblocks, h->n_groups * n_split, (unsigned)nf),
                               dim3(SQ_THREADS), 0, h->stream, d_pos + f0 * n * 3, n,
                               h->d_q.as<double>(), (int)h->n_q, h->d_offsets.as<int64_t>(),
                               h->n_groups, n_split, h->d_rho.as<double2>());
        hipLaunchKernelGGL(sq_pair_kernel, dim3((unsigned)ceil_div(h->n_q, PAIR_Q), h->n_pairs),
                           dim3(PAIR_Q * PAIR_FS), 0, h->stream, h->d_rho.as<double2>(), (int)nf, h->n_groups,
                           n_split, (int)h->n_q, h->d_pairs.as<int>(), h->n_pairs,
                           h->d_acc.as<double>());
    }
    h->timer.end(ev);
    MDX_HIP(hipGetLastError());
    return MDX_OK;
}

extern "C" {

int mdx_sq_create(mdx_sq_t *out, int dev, const double *wavevectors, int64_t n_q,
                  const int64_t *group_offsets, int n_groups, const int32_t *pairs, int n_pairs)
{
    MDX_REQUIRE(out && wavevectors && group_offsets && pairs, "NULL argument");
    MDX_REQUIRE(n_q >= 1 && n_q < (int64_t(1) << 30), "n_q out of range");
    MDX_REQUIRE(n_groups >= 1 && n_groups <= 1024, "n_groups out of range");
    MDX_REQUIRE(n_pairs >= 1 && n_pairs <= 65535, "n_pairs out of range");
    MDX_REQUIRE(group_offsets[0] >= 0, "group offsets must be non-negative");
    for (int g = 0; g < n_groups; ++g)
        MDX_REQUIRE(group_offsets[g + 1] >= group_offsets[g], "group offsets must not decrease");
    for (int p = 0; p < n_pairs; ++p) {
        int j = pairs[2 * p], k = pairs[2 * p + 1];
        MDX_REQUIRE((j == -1 && k == -1) || (j >= 0 && j < n_groups && k >= 0 && k < n_groups),
                    "pair %d = (%d, %d) is not a pair of groups", p, j, k);
    }
    MDX_TRY(set_device(dev));
    mdx_sq *h = new mdx_sq();
    h->dev = dev;
    h->n_q = n_q;
    h->n_groups = n_groups;
    h->n_pairs = n_pairs;
    h->offsets.assign(group_offsets, group_offsets + n_groups + 1);
    h->n_total = group_offsets[n_groups];
    int rc = MDX_OK;
    do {
        if ((rc = stream_acquire(&h->stream)) != MDX_OK)
            break;
        h->timer.stream = h->stream;
        if ((rc = h->d_q.ensure(size_t(24) * n_q)) != MDX_OK) break;
        if ((rc = h->d_offsets.ensure(size_t(8) * (n_groups + 1))) != MDX_OK) break;
        if ((rc = h->d_pairs.ensure(size_t(8) * n_pairs)) != MDX_OK) break;
        if ((rc = h->d_acc.ensure(size_t(8) * n_pairs * n_q)) != MDX_OK) break;
        if (hipMemcpy(h->d_q.ptr, wavevectors, size_t(24) * n_q, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_offsets.ptr, group_offsets, size_t(8) * (n_groups + 1),
                      hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_pairs.ptr, pairs, size_t(8) * n_pairs, hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(MDX_ERR_HIP, "upload failed");
            break;
        }
        std::vector<short> trip;
        h->lattice = sq_detect_lattice(wavevectors, n_q, h->lat, trip);
        if (h->lattice) {
            h->lat_lds = size_t(16) * h->lat.tile * (h->lat.R[0] + h->lat.R[1] + h->lat.R[2]);
            if ((rc = h->d_mtrip.ensure(size_t(8) * n_q)) != MDX_OK) break;
            if (hipMemcpy(h->d_mtrip.ptr, trip.data(), size_t(8) * n_q, hipMemcpyHostToDevice) !=
                    hipSuccess ||
                (h->lat_lds > 48 * 1024 &&
                 hipFuncSetAttribute(reinterpret_cast<const void *>(sq_rho_lattice_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)h->lat_lds) != hipSuccess)) {
                rc = fail(MDX_ERR_HIP, "lattice table setup failed");
                break;
            }
            std::vector<SqQuadItem> qitems;
            SqQuadShape shape;
            if (!getenv("MDX_SQ_NO_COLUMNS") && !getenv("MDX_SQ_NO_QUADS") &&
                sq_quad_plan(trip, n_q, h->lat, qitems, shape)) {
                h->n_qitems = shape.n_items;
                h->ipb = shape.ipb;
                h->n_sub = shape.n_sub;
                h->quad_lat = shape.lat;
                h->quad_lds = shape.lds;
                h->quad_regular = shape.regular_stride;
                if ((rc = h->d_qitems.ensure(sizeof(SqQuadItem) * qitems.size())) != MDX_OK) break;
                if (hipMemcpy(h->d_qitems.ptr, qitems.data(), sizeof(SqQuadItem) * qitems.size(),
                              hipMemcpyHostToDevice) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void *>(sq_rho_quads_pick(shape.regular_stride)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)h->quad_lds) != hipSuccess) {
                    rc = fail(MDX_ERR_HIP, "quad table setup failed");
                    break;
                }
                h->quads = true;
            }
            std::vector<SqColumnItem> items;
            if (!h->quads && !getenv("MDX_SQ_NO_COLUMNS") && sq_build_columns(trip, n_q, h->lat, items)) {
                h->n_items = (int)items.size();
                const int waves = (int)std::min<int64_t>(4, ceil_div(h->n_items, 64));
                h->col_threads = 64 * waves;
                // tables sized for ~24 waves per CU: 150 KB / (24 / waves) per block
                const int total_r = h->lat.R[0] + h->lat.R[1] + h->lat.R[2];
                const size_t budget = size_t(150) * 1024 * waves / 24;
                const int tile = (int)std::min<size_t>(64, budget / (size_t(16) * total_r));
                if (tile >= 4) {
                    h->col_lat = h->lat;
                    h->col_lat.tile = tile;
                    h->col_lds = size_t(16) * tile * total_r;
                    if ((rc = h->d_items.ensure(sizeof(SqColumnItem) * items.size())) != MDX_OK) break;
                    if (hipMemcpy(h->d_items.ptr, items.data(), sizeof(SqColumnItem) * items.size(),
                                  hipMemcpyHostToDevice) != hipSuccess) {
                        rc = fail(MDX_ERR_HIP, "column table upload failed");
                        break;
                    }
                    h->columns = true;
                }
            }
        }
    } while (0);
    if (rc != MDX_OK) {
        mdx_sq_destroy(h);
        return rc;
    }
    *out = h;
    return mdx_sq_reset(h);
}

int mdx_sq_destroy(mdx_sq_t h)
{
    if (!h)
        return MDX_OK;
    (void)hipSetDevice(h->dev);
    if (h->stream)
        (void)hipStreamSynchronize(h->stream);
    h->timer.destroy();
    h->pipe.destroy();      // waits for its copy stream
    // (every stream that touched them is idle: blocks and stream go back to the per-device pools, so that an
    // analysis object per call does not pay hipMalloc / hipFree / stream creation each time)
    for (DeviceBuffer *b : {&h->d_q, &h->d_offsets, &h->d_pairs, &h->d_acc, &h->d_rho, &h->d_stage[0],
                            &h->d_stage[1], &h->d_index, &h->d_mtrip, &h->d_items, &h->d_qitems})
        b->recycle();
    h->mol.recycle();
    if (h->stream)
        stream_release(h->stream);
    delete h;
    return MDX_OK;
}

int mdx_sq_reset(mdx_sq_t h)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipMemsetAsync(h->d_acc.ptr, 0, size_t(8) * h->n_pairs * h->n_q, h->stream));
    MDX_HIP(hipStreamSynchronize(h->stream));
    h->timer.reset();
    return MDX_OK;
}

int mdx_sq_set_grouping(mdx_sq_t h, int64_t n_molecules, const int64_t *offsets, const double *masses)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    MDX_REQUIRE(n_molecules <= 0 || n_molecules >= h->n_total,
                "%lld molecules given, the groups span %lld", (long long)n_molecules, (long long)h->n_total);
    return h->mol.set(n_molecules, offsets, masses);
}

int mdx_sq_accumulate_device(mdx_sq_t h, const float *d_pos, int64_t n, int64_t n_frames)
{
    MDX_REQUIRE(h && d_pos, "NULL argument");
    MDX_REQUIRE(n > 0 && n_frames >= 0, "bad size");
    MDX_TRY(set_device(h->dev));
    return sq_accumulate_device(h, d_pos, n, n_frames);
}

int mdx_sq_accumulate(mdx_sq_t h, const float *pos, int64_t n, int64_t n_frames)
{
    MDX_REQUIRE(h && pos, "NULL argument");
    MDX_REQUIRE(n > 0 && n_frames >= 0, "bad size");
    MDX_TRY(set_device(h->dev));
    // copies of slab k+1 overlap the kernels of slab k (StagePipeline)
    const int64_t slab = std::min<int64_t>(std::max<int64_t>(n_frames, 1),
                                           std::max<int64_t>(1, (int64_t(64) << 20) / (12 * n)));
    return h->pipe.run(
        h->stream, n_frames, slab,
        [&](int b, int64_t f0, int64_t nf) -> int {
            MDX_TRY(h->d_stage[b].ensure(size_t(12) * n * slab));
            return device_stager(h->dev).upload(h->dev, h->pipe.copy_stream, h->d_stage[b].ptr,
                                         pos + f0 * n * 3, size_t(12) * n * nf);
        },
        [&](int b, int64_t, int64_t nf) -> int {
            return sq_accumulate_device(h, h->d_stage[b].as<float>(), n, nf);
        });
}

// Frames straight from a trajectory file.  index: host int32[n_index] particle indices in the
// order of the concatenated groups (structure.py:1484-1486), or NULL for the file's first
// n_index particles.
int mdx_sq_accumulate_traj(mdx_sq_t h, mdx_traj_t traj, const int64_t *frames, int64_t n_frames,
                           const int32_t *index, int64_t n_index)
{
    MDX_REQUIRE(h && traj, "NULL handle");
    MDX_REQUIRE(n_frames >= 0 && (n_frames == 0 || frames), "bad frame list");
    MDX_TRY(set_device(h->dev));
    Trajectory *t = mdx_traj_internal(traj);
    const int64_t n = index ? n_index : (n_index > 0 ? n_index : t->n_atoms);
    MDX_REQUIRE(h->mol.active() || n >= h->n_total, "the selection holds %lld particles, the groups need %lld",
                (long long)n, (long long)h->n_total);
    MDX_REQUIRE(index || n <= t->n_atoms, "selection larger than the trajectory");
    if (n_frames == 0)
        return MDX_OK;
    MDX_TRY(h->pipe.ensure());
    const int *d_index = nullptr;
    if (index) {
        for (int64_t i = 0; i < n; ++i)
            if (index[i] < 0 || index[i] >= t->n_atoms)
                return fail(MDX_ERR_INVALID_VALUE, "particle index %d out of range [0, %lld)",
                            index[i], (long long)t->n_atoms);
        MDX_HIP(hipStreamSynchronize(h->pipe.copy_stream));
        MDX_TRY(h->d_index.ensure(size_t(4) * n));
        MDX_HIP(hipMemcpy(h->d_index.ptr, index, size_t(4) * n, hipMemcpyHostToDevice));
        d_index = h->d_index.as<int>();
    }
    const int64_t slab = std::min<int64_t>(
        n_frames, std::max<int64_t>(1, (int64_t(64) << 20) / (12 * t->n_atoms)));
    return h->pipe.run(
        h->stream, n_frames, slab,
        [&](int b, int64_t f0, int64_t nf) -> int {
            MDX_TRY(h->d_stage[b].ensure(size_t(12) * n * slab));
            TrajSelection sel{d_index, n, h->d_stage[b].as<float>()};
            return t->stage_async(h->dev, h->pipe.copy_stream, frames + f0, nf, &sel, 1);
        },
        [&](int b, int64_t, int64_t nf) -> int {
            return sq_accumulate_device(h, h->d_stage[b].as<float>(), n, nf);
        });
}

int mdx_sq_result(mdx_sq_t h, double *ssf)
{
    MDX_REQUIRE(h && ssf, "NULL argument");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    h->timer.collect();
    MDX_HIP(hipMemcpy(ssf, h->d_acc.ptr, size_t(8) * h->n_pairs * h->n_q, hipMemcpyDeviceToHost));
    return MDX_OK;
}

int mdx_sq_synchronize(mdx_sq_t h)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    return MDX_OK;
}

int mdx_sq_stats(mdx_sq_t h, int64_t *launches, double *kernel_ms)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    h->timer.collect();
    if (launches) *launches = h->timer.launches;
    if (kernel_ms) *kernel_ms = h->timer.total_ms;
    return MDX_OK;
}

int mdx_sq_enable_timing(mdx_sq_t h, int on)
{
    MDX_REQUIRE(h, "NULL handle");
    h->timer.enabled = on != 0;
    return MDX_OK;
}

int mdx_sq_internal_buffer(mdx_sq_t h, double **d_acc, int64_t *n, hipStream_t *stream)
{
    MDX_TRY(set_device(h->dev));
    *d_acc = h->d_acc.as<double>();
    *n = int64_t(h->n_pairs) * h->n_q;
    *stream = h->stream;
    return MDX_OK;
}

int mdx_fourier_sum(int dev, const double *wavevectors, int64_t n_q, const double *positions,
                    int64_t n, double *out_re_im)
{
    MDX_REQUIRE(wavevectors && positions && out_re_im, "NULL argument");
    MDX_REQUIRE(n_q >= 1 && n >= 0 && n_q < (int64_t(1) << 30), "bad size");
    MDX_TRY(set_device(dev));
    const int qblocks = (int)ceil_div(n_q, SQ_THREADS);
    int n_split = 1;
    while (qblocks * n_split < 512 && n_split < 256 && n / (n_split * 2) >= 512)
        n_split *= 2;
    DeviceBuffer dq, dp, dpart;
    int rc = MDX_OK;
    std::vector<double> part(size_t(2) * n_split * n_q);
    do {
        if ((rc = dq.ensure(size_t(24) * n_q)) != MDX_OK) break;
        if ((rc = dp.ensure(size_t(24) * std::max<int64_t>(n, 1))) != MDX_OK) break;
        if ((rc = dpart.ensure(size_t(16) * n_split * n_q)) != MDX_OK) break;
        if (hipMemcpy(dq.ptr, wavevectors, size_t(24) * n_q, hipMemcpyHostToDevice) != hipSuccess ||
            (n && hipMemcpy(dp.ptr, positions, size_t(24) * n, hipMemcpyHostToDevice) != hipSuccess)) {
            rc = fail(MDX_ERR_HIP, "upload failed");
            break;
        }
        hipLaunchKernelGGL(sq_fourier_sum_f64_kernel, dim3(qblocks, n_split), dim3(SQ_THREADS), 0, 0,
                           dp.as<double>(), n, dq.as<double>(), (int)n_q, n_split,
                           dpart.as<double2>());
        if (hipDeviceSynchronize() != hipSuccess ||
            hipMemcpy(part.data(), dpart.ptr, size_t(16) * n_split * n_q, hipMemcpyDeviceToHost) !=
                hipSuccess) {
            rc = fail(MDX_ERR_HIP, "fourier-sum kernel failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
    } while (0);
    dq.release();
    dp.release();
    dpart.release();
    if (rc != MDX_OK)
        return rc;
    for (int64_t q = 0; q < n_q; ++q) {
        double re = 0.0, im = 0.0;
        for (int s = 0; s < n_split; ++s) {
            re += part[2 * (size_t(s) * n_q + q)];
            im += part[2 * (size_t(s) * n_q + q) + 1];
        }
        out_re_im[2 * q] = re;
        out_re_im[2 * q + 1] = im;
    }
    return MDX_OK;
}

// ---- row sums of cos / sin of caller-supplied phase matrices (accelerated.py:249-321, :323-627)
static int trig_rowsums_device(int dev, const double *d_x, int64_t n_rows, int64_t n_cols, double *d_cos,
                               double *d_sin, hipStream_t stream, DeviceBuffer &d_part)
{
    // enough blocks to fill the chip, whole 256-column pieces, at most 65 535 rows per launch
    int n_split = 1;
    while (n_rows * n_split < 2048 && n_split < 1024 && n_cols / (2 * n_split) >= 4 * TRIG_THREADS)
        n_split *= 2;
    const int64_t rows_max = 65535;
    MDX_TRY(d_part.ensure(size_t(16) * std::min(n_rows, rows_max) * n_split));
    for (int64_t r0 = 0; r0 < n_rows; r0 += rows_max) {
        const int64_t nr = std::min(rows_max, n_rows - r0);
        hipLaunchKernelGGL(trig_rowsums_kernel, dim3((unsigned)n_split, (unsigned)nr), dim3(TRIG_THREADS), 0, stream,
                           d_x + r0 * n_cols, n_cols, n_split, d_part.as<double2>());
        hipLaunchKernelGGL(trig_fold_kernel, dim3((unsigned)ceil_div(nr, 256)), dim3(256), 0, stream,
                           d_part.as<double2>(), nr, n_split, d_cos ? d_cos + r0 : nullptr,
                           d_sin ? d_sin + r0 : nullptr);
    }
    MDX_HIP(hipGetLastError());
    return MDX_OK;
}

int mdx_trig_rowsums_device(int dev, const double *d_x, int64_t n_rows, int64_t n_cols, double *d_cos,
                            double *d_sin)
{
    MDX_REQUIRE(d_x && (d_cos || d_sin), "NULL argument");
    MDX_REQUIRE(n_rows >= 1 && n_cols >= 0, "bad size");
    MDX_TRY(set_device(dev));
    DeviceBuffer d_part;
    const int rc = trig_rowsums_device(dev, d_x, n_rows, n_cols, d_cos, d_sin, nullptr, d_part);
    const hipError_t e = hipDeviceSynchronize();
    d_part.recycle();
    MDX_TRY(rc);
    MDX_HIP(e);
    return MDX_OK;
}

int mdx_trig_rowsums(int dev, const double *x, int64_t n_rows, int64_t n_cols, double *cos_out, double *sin_out)
{
    MDX_REQUIRE(x && (cos_out || sin_out), "NULL argument");
    MDX_REQUIRE(n_rows >= 1 && n_cols >= 0, "bad size");
    MDX_TRY(set_device(dev));
    // row slabs of <= 1 GiB: the upload of slab k + 1 (pinned ring, or one DMA out of page-locked memory) is
    // queued behind the kernels of slab k on the same stream, the copy stream of the ring runs ahead of them
    const int64_t slab_rows = std::max<int64_t>(1, std::min<int64_t>(n_rows, (int64_t(1) << 30) / std::max<int64_t>(8, 8 * n_cols)));
    DeviceBuffer d_x[2], d_out, d_part;
    hipStream_t stream = nullptr;
    std::vector<double> host(size_t(2) * n_rows);
    auto run = [&]() -> int {
        MDX_TRY(stream_acquire(&stream));
        MDX_TRY(d_out.ensure(size_t(16) * n_rows));
        double *d_cos = d_out.as<double>(), *d_sin = d_cos + n_rows;
        int k = 0;
        for (int64_t r0 = 0; r0 < n_rows; r0 += slab_rows, ++k) {
            const int64_t nr = std::min(slab_rows, n_rows - r0);
            DeviceBuffer &buf = d_x[k & 1];
            MDX_TRY(buf.ensure(size_t(8) * std::max<int64_t>(1, nr * n_cols)));
            if (n_cols > 0)
                MDX_TRY(device_stager(dev).upload(dev, stream, buf.ptr, x + r0 * n_cols, size_t(8) * nr * n_cols));
            MDX_TRY(trig_rowsums_device(dev, buf.as<double>(), nr, n_cols, d_cos + r0, d_sin + r0, stream, d_part));
        }
        MDX_HIP(hipMemcpyAsync(host.data(), d_out.ptr, size_t(16) * n_rows, hipMemcpyDeviceToHost, stream));
        MDX_HIP(hipStreamSynchronize(stream));
        return MDX_OK;
    };
    const int rc = run();
    (void)hipDeviceSynchronize();
    if (stream)
        stream_release(stream);
    d_x[0].recycle();
    d_x[1].recycle();
    d_out.recycle();
    d_part.recycle();
    MDX_TRY(rc);
    if (cos_out)
        memcpy(cos_out, host.data(), size_t(8) * n_rows);
    if (sin_out)
        memcpy(sin_out, host.data() + n_rows, size_t(8) * n_rows);
    return MDX_OK;
}

int mdx_inner(int dev, const double *wavevectors, int64_t n_q, const double *positions, int64_t n, double *out)
{
    MDX_REQUIRE(wavevectors && positions && out, "NULL argument");
    MDX_REQUIRE(n_q >= 1 && n >= 1 && n_q <= 65535, "bad size");
    MDX_TRY(set_device(dev));
    DeviceBuffer dq, dp, dout;
    hipStream_t stream = nullptr;
    auto run = [&]() -> int {
        MDX_TRY(stream_acquire(&stream));
        MDX_TRY(dq.ensure(size_t(24) * n_q));
        MDX_TRY(dp.ensure(size_t(24) * n));
        MDX_TRY(dout.ensure(size_t(8) * n_q * n));
        HostStager &ring = device_stager(dev);
        MDX_TRY(ring.upload(dev, stream, dq.ptr, wavevectors, size_t(24) * n_q));
        MDX_TRY(ring.upload(dev, stream, dp.ptr, positions, size_t(24) * n));
        hipLaunchKernelGGL(inner_kernel, dim3((unsigned)ceil_div(n, 256), (unsigned)n_q), dim3(256), 0, stream,
                           dq.as<double>(), n_q, dp.as<double>(), n, dout.as<double>());
        MDX_HIP(hipGetLastError());
        return ring.download(dev, stream, out, dout.ptr, size_t(8) * n_q * n);
    };
    const int rc = run();
    (void)hipDeviceSynchronize();
    if (stream)
        stream_release(stream);
    dq.recycle();
    dp.recycle();
    dout.recycle();
    return rc;
}


}  // extern "C"
