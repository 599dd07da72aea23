// mdx_isf.hip — coherent / incoherent intermediate scattering functions on gfx950.
//
// Carries IntermediateScatteringFunction._single_frame (reference
// src/mdhelper/analysis/structure.py:1956-2083):
//
//   rho_g(q, f) = sum_{j in g} exp(i q . r_j(f))                          (sq_rho_kernel)
//   cisf[lag][p](q) += Re rho_j(f-lag) rho_j(f)^*                          p = (j, j)
//                   += Re rho_j(f-lag) rho_k(f)^* + Re rho_k(f-lag) rho_j(f)^*   p = (j, k)
//                   += Re rho(f-lag) rho(f)^*  with rho = sum_g rho_g      mode=None
//   iisf[lag][g](q) += sum_{j in g} cos(q . (r_j(f) - r_j(f-lag)))         (:1991-1996)
//
// for every analysed frame f and every lag <= min(n_lags - 1, f).  The reference
// keeps a ring of n_lags frames of positions and of rho on the host; here the rings
// live in HBM (2 n_lags slots, so a chunk of new frames never overwrites history a
// frame of the same chunk still needs) and both forms ("exp", "trig") map onto the
// same fp64 kernels.  The incoherent part is n_lags x the work of a structure
// factor per frame and dominates; it runs as one fused kernel over
// (wavevector block, group x particle split, lag) with the frames of a chunk
// looped inside, so sums are formed in a fixed order (run-to-run reproducible).
#include "mdx_common.hpp"
#include "mdx_internal.hpp"
#include "mdx_molecules.hpp"

using namespace mdx;

#include "mdx_sq_device.hpp"
#include "mdx_traj.hpp"

using namespace mdx_sq_dev;

namespace {

// cisf[lag][p][q] += sum over the chunk's frames f of the lagged products
__global__ __launch_bounds__(256) void isf_coherent_kernel(
    const double2 *__restrict__ ring, int ring_slots, int n_groups, int n_q,
    const int *__restrict__ pairs, int n_pairs, int n_lags, long long f_first, int n_new,
    double *__restrict__ cisf)
{
    const int qi = blockIdx.x * 256 + threadIdx.x;
    const int p = blockIdx.y, lag = blockIdx.z;
    if (qi >= n_q)
        return;
    const int j = pairs[2 * p], k = pairs[2 * p + 1];
    auto rho = [&](long long f, int g) {
        return ring[(int64_t(f % ring_slots) * n_groups + g) * n_q + qi];
    };
    auto total = [&](long long f) {
        double2 r = make_double2(0.0, 0.0);
        for (int g = 0; g < n_groups; ++g) {
            double2 v = rho(f, g);
            r.x += v.x;
            r.y += v.y;
        }
        return r;
    };
    double sum = 0.0;
    for (int i = 0; i < n_new; ++i) {
        const long long f = f_first + i;
        if (f < lag)
            continue;
        if (j < 0) {
            double2 a = total(f - lag), b = total(f);
            sum += a.x * b.x + a.y * b.y;
        } else if (j == k) {
            double2 a = rho(f - lag, j), b = rho(f, j);
            sum += a.x * b.x + a.y * b.y;
        } else {
            double2 aj = rho(f - lag, j), bk = rho(f, k), ak = rho(f - lag, k), bj = rho(f, j);
            sum += (aj.x * bk.x + aj.y * bk.y) + (ak.x * bj.x + ak.y * bj.y);
        }
    }
    cisf[(int64_t(lag) * n_pairs + p) * n_q + qi] += sum;
}

// part[split][lag][slot][q] = sum over the chunk's frames and the split's particles of
// cos(q . (r(f) - r(f - lag)));  `slot` indexes the groups that have an incoherent part
__global__ __launch_bounds__(SQ_THREADS) void isf_incoherent_kernel(
    const float *__restrict__ pos_ring, int ring_slots, int64_t n_atoms,
    const double *__restrict__ qv, int n_q, const int64_t *__restrict__ ranges /*[n_slots][2]*/,
    int n_slots, int n_split, int n_lags, long long f_first, int n_new, double *__restrict__ part)
{
    __shared__ float cx[SQ_TILE / 2], cy[SQ_TILE / 2], cz[SQ_TILE / 2];
    __shared__ float px[SQ_TILE / 2], py[SQ_TILE / 2], pz[SQ_TILE / 2];
    constexpr int TILE = SQ_TILE / 2;
    const int tid = threadIdx.x;
    const int qb = blockIdx.x;
    const int slot = blockIdx.y / n_split, sp = blockIdx.y % n_split;
    const int lag = blockIdx.z;

    double q0[SQ_QPT], q1[SQ_QPT], q2[SQ_QPT], ac[SQ_QPT];
#pragma unroll
    for (int u = 0; u < SQ_QPT; ++u) {
        int qi = qb * SQ_QPB + u * SQ_THREADS + tid;
        bool ok = qi < n_q;
        q0[u] = ok ? qv[3 * int64_t(qi) + 0] : 0.0;
        q1[u] = ok ? qv[3 * int64_t(qi) + 1] : 0.0;
        q2[u] = ok ? qv[3 * int64_t(qi) + 2] : 0.0;
        ac[u] = 0.0;
    }
    const int64_t g_lo = ranges[2 * slot], g_hi = ranges[2 * slot + 1];
    const int64_t per = (g_hi - g_lo + n_split - 1) / n_split;
    const int64_t lo = g_lo + sp * per, hi = min(g_hi, lo + per);

    for (int i = 0; i < n_new; ++i) {
        const long long f = f_first + i;
        if (f < lag)
            continue;
        const float *C = pos_ring + int64_t(f % ring_slots) * n_atoms * 3;
        const float *P = pos_ring + int64_t((f - lag) % ring_slots) * n_atoms * 3;
        for (int64_t base = lo; base < hi; base += TILE) {
            const int cnt = (int)min<int64_t>(TILE, hi - base);
            __syncthreads();
            for (int e = tid; e < cnt * 3; e += SQ_THREADS) {
                float vc = C[base * 3 + e], vp = P[base * 3 + e];
                int a = e / 3, k = e - 3 * a;
                (k == 0 ? cx : k == 1 ? cy : cz)[a] = vc;
                (k == 0 ? px : k == 1 ? py : pz)[a] = vp;
            }
            __syncthreads();
            for (int a = 0; a < cnt; ++a) {
                // float32 coordinates widened first: the difference is exact in fp64
                const double dx = (double)cx[a] - (double)px[a], dy = (double)cy[a] - (double)py[a],
                             dz = (double)cz[a] - (double)pz[a];
#pragma unroll
                for (int u = 0; u < SQ_QPT; ++u) {
                    double ph = fma(q2[u], dz, fma(q1[u], dy, q0[u] * dx));
                    double s, c;
                    sincos_f64(ph, s, c);
                    ac[u] += c;
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < SQ_QPT; ++u) {
        int qi = qb * SQ_QPB + u * SQ_THREADS + tid;
        if (qi < n_q)
            part[((int64_t(sp) * n_lags + lag) * n_slots + slot) * n_q + qi] = ac[u];
    }
}

// Lattice wavevectors (mdx_sq_device.hpp): the same sums through separable phase tables of
// the DISPLACEMENTS — one sincos per (particle, axis), two complex multiplies per (q, particle).
__global__ __launch_bounds__(SQ_THREADS, 6) void isf_incoherent_lattice_kernel(
    const float *__restrict__ pos_ring, int ring_slots, int64_t n_atoms,
    const short4 *__restrict__ mtrip, int n_q, SqLattice lat,
    const int64_t *__restrict__ ranges /*[n_slots][2]*/, int n_slots, int n_split, int n_lags,
    long long f_first, int n_new, double *__restrict__ part)
{
    extern __shared__ double2 lat_tab[];
    const int tid = threadIdx.x;
    const int qb = blockIdx.x;
    const int slot = blockIdx.y / n_split, sp = blockIdx.y % n_split;
    const int lag = blockIdx.z;
    const int A = lat.tile;
    double2 *tab[3] = {lat_tab, lat_tab + size_t(A) * lat.R[0],
                       lat_tab + size_t(A) * (lat.R[0] + lat.R[1])};
    int i0[SQ_QPT], i1[SQ_QPT], i2[SQ_QPT];
    double ac[SQ_QPT];
#pragma unroll
    for (int u = 0; u < SQ_QPT; ++u) {
        int qi = qb * SQ_QPB + u * SQ_THREADS + tid;
        short4 m = mtrip[min(qi, n_q - 1)];
        i0[u] = m.x - lat.mmin[0];
        i1[u] = m.y - lat.mmin[1];
        i2[u] = m.z - lat.mmin[2];
        ac[u] = 0.0;
    }
    const int64_t g_lo = ranges[2 * slot], g_hi = ranges[2 * slot + 1];
    const int64_t per = (g_hi - g_lo + n_split - 1) / n_split;
    const int64_t lo = g_lo + sp * per, hi = min(g_hi, lo + per);

    for (int i = 0; i < n_new; ++i) {
        const long long f = f_first + i;
        if (f < lag)
            continue;
        const float *C = pos_ring + int64_t(f % ring_slots) * n_atoms * 3;
        const float *P = pos_ring + int64_t((f - lag) % ring_slots) * n_atoms * 3;
        for (int64_t base = lo; base < hi; base += A) {
            const int cnt = (int)min<int64_t>(A, hi - base);
            __syncthreads();
            for (int t = tid; t < cnt * 3; t += SQ_THREADS) {
                const int a = t / 3, k = t - 3 * a;
                // float32 coordinates widened first: the difference is exact in fp64
                const double d = (double)C[(base + a) * 3 + k] - (double)P[(base + a) * 3 + k];
                sq_lattice_fill_row(tab[k] + size_t(a) * lat.R[k], lat.base[k] * d, lat.mmin[k],
                                    lat.R[k]);
            }
            __syncthreads();
            for (int a = 0; a < cnt; ++a) {
                const double2 *r0 = tab[0] + size_t(a) * lat.R[0], *r1 = tab[1] + size_t(a) * lat.R[1],
                              *r2 = tab[2] + size_t(a) * lat.R[2];
#pragma unroll
                for (int u = 0; u < SQ_QPT; ++u) {
                    const double2 ex = r0[i0[u]], ey = r1[i1[u]], ez = r2[i2[u]];
                    const double tr = fma(ex.x, ey.x, -ex.y * ey.y), ti = fma(ex.x, ey.y, ex.y * ey.x);
                    ac[u] += fma(tr, ez.x, -ti * ez.y);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < SQ_QPT; ++u) {
        int qi = qb * SQ_QPB + u * SQ_THREADS + tid;
        if (qi < n_q)
            part[((int64_t(sp) * n_lags + lag) * n_slots + slot) * n_q + qi] = ac[u];
    }
}

// iisf[lag][slot][q] += sum over splits (fixed order)
__global__ void isf_reduce_kernel(const double *__restrict__ part, int n_split, int64_t n,
                                  double *__restrict__ iisf)
{
    int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    double s = 0.0;
    for (int sp = 0; sp < n_split; ++sp)
        s += part[int64_t(sp) * n + i];
    iisf[i] += s;
}

}  // namespace

// ring[frame][g][q] = sum over the particle parts of parts[frame][g][part][q], fixed order
__global__ __launch_bounds__(256) void isf_rho_merge_kernel(const double2 *__restrict__ parts, int n_parts,
                                                            int n_q, int64_t n_out,
                                                            double2 *__restrict__ ring)
{
    const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;   // (frame, g, q)
    if (i >= n_out)
        return;
    const int64_t fg = i / n_q;
    const int q = int(i - fg * n_q);
    double2 r = make_double2(0.0, 0.0);
    for (int s = 0; s < n_parts; ++s) {
        const double2 v = parts[(fg * n_parts + s) * n_q + q];
        r.x += v.x;
        r.y += v.y;
    }
    ring[i] = r;
}

struct mdx_isf {
    int dev = 0;
    hipStream_t stream = nullptr;
    int64_t n_q = 0, n_total = 0;
    int n_groups = 0, n_pairs = 0, n_lags = 0, n_slots = 0, ring_slots = 0;
    bool incoherent = false;
    long long frames_seen = 0;
    std::vector<int64_t> offsets;
    std::vector<int64_t> ranges;      // particle range of every incoherent slot
    DeviceBuffer d_q, d_offsets, d_pairs, d_ranges, d_rho_ring, d_pos_ring, d_cisf, d_iisf, d_part,
        d_pos_stage, d_index, d_mtrip, d_row_stage, d_rho_parts;
    MoleculeStage mol;      // optional centre-of-mass stage (groupings other than "atoms")
    bool lattice = false;   // grid wavevectors: separable phase tables (mdx_sq_device.hpp)
    SqLattice lat{};
    size_t lat_lds = 0;
    // incoherent part through the register-blocked column form (isf_incoherent_quads_kernel)
    bool quads = false;
    SqQuadShape quad{};
    DeviceBuffer d_qitems;
    StreamTimer timer;
    StagePipeline pipe;             // host-buffer / trajectory-file entry points: slab k+1 is staged while slab k runs
    DeviceBuffer d_stage[2];
};

// Frames must arrive in analysis order; consecutive calls continue the same series.
// source(d_dst, done, nf) queues on h->stream whatever brings frames [done, done+nf) of the
// caller's series into d_dst (float32[nf][n][3]).
template <typename Source>
static int isf_accumulate(mdx_isf *h, int64_t n, int64_t n_frames, Source source)
{
    MDX_TRY(set_device(h->dev));
    // with a grouping set the incoming rows are particles sorted molecule by molecule; the rings
    // and kernels below see the float32 centres of mass
    const int64_t n_rows = n;
    if (h->mol.active()) {
        MDX_REQUIRE(n == h->mol.n_atoms, "%lld particles given, the grouping was defined for %lld",
                    (long long)n, (long long)h->mol.n_atoms);
        n = h->mol.n_groups;
    }
    MDX_REQUIRE(n >= h->n_total && n_frames >= 0, "bad size");
    if (h->incoherent)
        MDX_TRY(h->d_pos_ring.ensure(size_t(12) * n * h->ring_slots));
    const int qblocks = (int)ceil_div(h->n_q, SQ_QPB);
    int64_t done = 0;
    while (done < n_frames) {
        // a chunk: at most n_lags new frames, contiguous in the ring
        const long long f0 = h->frames_seen;
        const int slot0 = int(f0 % h->ring_slots);
        const int64_t nf = std::min<int64_t>(std::min<int64_t>(h->n_lags, h->ring_slots - slot0),
                                             n_frames - done);
        float *d_new = nullptr;
        if (h->incoherent) {
            d_new = h->d_pos_ring.as<float>() + int64_t(slot0) * n * 3;
        } else {
            MDX_TRY(h->d_pos_stage.ensure(size_t(12) * n * nf));
            d_new = h->d_pos_stage.as<float>();
        }
        if (h->mol.active()) {
            MDX_TRY(h->d_row_stage.ensure(size_t(12) * n_rows * nf));
            MDX_TRY(source(h->d_row_stage.as<float>(), done, nf));
            MDX_TRY(h->mol.run(h->stream, h->d_row_stage.as<float>(), nf, d_new, nullptr));
        } else {
            MDX_TRY(source(d_new, done, nf));
        }
        hipEvent_t ev = h->timer.begin();
        {
            // rho_g(q) of the new frames -> ring.  A chunk holds at most n_lags frames, so the
            // particles of a group are split (and the parts summed in a fixed order) until the
            // grid fills the chip.
            int64_t max_group = 0;
            for (int g = 0; g < h->n_groups; ++g)
                max_group = std::max(max_group, h->offsets[g + 1] - h->offsets[g]);
            const int rblocks = h->quads ? h->quad.blocks() : qblocks;
            int rs = 1;
            while (int64_t(rblocks) * 4 * h->n_groups * rs * nf < 4096 && rs < 64 &&
                   max_group / (rs * 2) >= 1024)
                rs *= 2;
            double2 *ring = h->d_rho_ring.as<double2>() + int64_t(slot0) * h->n_groups * h->n_q;
            double2 *dst = ring;
            if (rs > 1) {
                MDX_TRY(h->d_rho_parts.ensure(size_t(16) * nf * h->n_groups * rs * h->n_q));
                dst = h->d_rho_parts.as<double2>();
            }
            if (h->quads)
                hipLaunchKernelGGL(sq_rho_quads_pick(h->quad.regular_stride), dim3(rblocks, h->n_groups * rs, (unsigned)nf),
                                   dim3(SQ_QUAD_THREADS), h->quad.lds, h->stream, d_new, n,
                                   h->d_qitems.as<SqQuadItem>(), h->quad.n_items, h->quad.ipb,
                                   h->quad.n_sub, (int)h->n_q, h->quad.lat, h->d_offsets.as<int64_t>(),
                                   h->n_groups, rs, dst);
            else if (h->lattice)
                hipLaunchKernelGGL(sq_rho_lattice_kernel, dim3(qblocks, h->n_groups * rs, (unsigned)nf),
                                   dim3(SQ_THREADS), h->lat_lds, h->stream, d_new, n,
                                   h->d_mtrip.as<short4>(), (int)h->n_q, h->lat,
                                   h->d_offsets.as<int64_t>(), h->n_groups, rs, dst);
            else
                hipLaunchKernelGGL(sq_rho_kernel, dim3(qblocks, h->n_groups * rs, (unsigned)nf),
                                   dim3(SQ_THREADS), 0, h->stream, d_new, n, h->d_q.as<double>(),
                                   (int)h->n_q, h->d_offsets.as<int64_t>(), h->n_groups, rs, dst);
            if (rs > 1) {
                const int64_t n_out = nf * h->n_groups * h->n_q;
                hipLaunchKernelGGL(isf_rho_merge_kernel, dim3((unsigned)ceil_div(n_out, 256)), dim3(256), 0,
                                   h->stream, dst, rs, (int)h->n_q, n_out, ring);
            }
        }
        hipLaunchKernelGGL(isf_coherent_kernel,
                           dim3((unsigned)ceil_div(h->n_q, 256), h->n_pairs, h->n_lags), dim3(256), 0,
                           h->stream, h->d_rho_ring.as<double2>(), h->ring_slots, h->n_groups,
                           (int)h->n_q, h->d_pairs.as<int>(), h->n_pairs, h->n_lags, f0, (int)nf,
                           h->d_cisf.as<double>());
        if (h->incoherent) {
            int64_t max_range = 0;
            for (int s = 0; s < h->n_slots; ++s)
                max_range = std::max(max_range, h->ranges[2 * s + 1] - h->ranges[2 * s]);
            int n_split = 1;
            while (int64_t(qblocks) * h->n_slots * n_split * h->n_lags < 1024 && n_split < 64 &&
                   max_range / (n_split * 2) >= SQ_TILE)
                n_split *= 2;
            const int64_t n_out = int64_t(h->n_lags) * h->n_slots * h->n_q;
            MDX_TRY(h->d_part.ensure(size_t(8) * n_out * n_split));
            if (h->quads) {
                const int iblocks = h->quad.blocks();
                n_split = 1;   // 4 waves per block here
                while (int64_t(iblocks) * 4 * h->n_slots * n_split * h->n_lags < 4096 && n_split < 64 &&
                       max_range / (n_split * 2) >= 4 * h->quad.lat.tile)
                    n_split *= 2;
                MDX_TRY(h->d_part.ensure(size_t(8) * n_out * n_split));
                hipLaunchKernelGGL(isf_incoherent_quads_pick(h->quad.regular_stride),
                                   dim3(iblocks, h->n_slots * n_split, h->n_lags), dim3(SQ_QUAD_THREADS),
                                   h->quad.lds, h->stream, h->d_pos_ring.as<float>(), h->ring_slots, n,
                                   h->d_qitems.as<SqQuadItem>(), h->quad.n_items, h->quad.ipb,
                                   h->quad.n_sub, (int)h->n_q, h->quad.lat, h->d_ranges.as<int64_t>(),
                                   h->n_slots, n_split, h->n_lags, f0, (int)nf, h->d_part.as<double>());
            } else if (h->lattice)
                hipLaunchKernelGGL(isf_incoherent_lattice_kernel,
                                   dim3(qblocks, h->n_slots * n_split, h->n_lags), dim3(SQ_THREADS),
                                   h->lat_lds, h->stream, h->d_pos_ring.as<float>(), h->ring_slots, n,
                                   h->d_mtrip.as<short4>(), (int)h->n_q, h->lat,
                                   h->d_ranges.as<int64_t>(), h->n_slots, n_split, h->n_lags, f0,
                                   (int)nf, h->d_part.as<double>());
            else
                hipLaunchKernelGGL(isf_incoherent_kernel,
                                   dim3(qblocks, h->n_slots * n_split, h->n_lags), dim3(SQ_THREADS), 0,
                                   h->stream, h->d_pos_ring.as<float>(), h->ring_slots, n,
                                   h->d_q.as<double>(), (int)h->n_q, h->d_ranges.as<int64_t>(),
                                   h->n_slots, n_split, h->n_lags, f0, (int)nf,
                                   h->d_part.as<double>());
            hipLaunchKernelGGL(isf_reduce_kernel, dim3((unsigned)ceil_div(n_out, 256)), dim3(256), 0,
                               h->stream, h->d_part.as<double>(), n_split, n_out,
                               h->d_iisf.as<double>());
        }
        h->timer.end(ev);
        MDX_HIP(hipGetLastError());
        // (no wait here: no source reads caller memory on this stream any more — host and file frames are staged
        // by the pipeline, whose events guard the staging sets, and device frames follow the *_device contract —
        // so the chunks queue back to back and the next slab's copy runs beside them)
        h->frames_seen += nf;
        done += nf;
    }
    return MDX_OK;
}

extern "C" {

int mdx_isf_set_grouping(mdx_isf_t h, int64_t n_molecules, const int64_t *offsets, const double *masses)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    MDX_REQUIRE(h->frames_seen == 0, "the grouping cannot change in the middle of a series");
    MDX_REQUIRE(n_molecules <= 0 || n_molecules >= h->n_total,
                "%lld molecules given, the groups span %lld", (long long)n_molecules, (long long)h->n_total);
    return h->mol.set(n_molecules, offsets, masses);
}

// frames per staged slab: ~64 MB of coordinates, a whole number of n_lags chunks
static int64_t isf_slab_frames(const mdx_isf *h, int64_t n_rows)
{
    const int64_t by_bytes = (int64_t(64) << 20) / (12 * std::max<int64_t>(n_rows, 1));
    return std::max<int64_t>(1, by_bytes / h->n_lags) * h->n_lags;
}

int mdx_isf_accumulate(mdx_isf_t h, const float *pos, int64_t n, int64_t n_frames)
{
    MDX_REQUIRE(h && pos, "NULL argument");
    MDX_REQUIRE(n > 0 && n_frames >= 0, "bad size");
    MDX_TRY(set_device(h->dev));
    // Slabs of frames travel host -> pinned ring -> HBM on the pipeline's copy stream while the kernels of the
    // slab before run (a pageable hipMemcpyAsync on the compute stream, the first form of this entry point, sat
    // between the kernels: 0.87 of the resident rate at 32 768 particles x 64 lags); the frames then enter the
    // position / rho rings by a device copy, as for mdx_isf_accumulate_device.
    // (whole multiples of n_lags: the kernels take the frames in chunks of n_lags, and a ragged last chunk per
    // slab is a launch of its own over a fraction of the work)
    const int64_t slab = std::min<int64_t>(std::max<int64_t>(n_frames, 1), isf_slab_frames(h, n));
    return h->pipe.run(
        h->stream, n_frames, slab,
        [&](int b, int64_t f0, int64_t nf) -> int {
            MDX_TRY(h->d_stage[b].ensure(size_t(12) * n * slab));
            return device_stager(h->dev).upload(h->dev, h->pipe.copy_stream, h->d_stage[b].ptr,
                                                pos + f0 * n * 3, size_t(12) * n * nf);
        },
        [&](int b, int64_t, int64_t nf) -> int {
            const float *d_pos = h->d_stage[b].as<float>();
            return isf_accumulate(h, n, nf, [&](float *d_dst, int64_t done, int64_t m) -> int {
                MDX_HIP(hipMemcpyAsync(d_dst, d_pos + done * n * 3, size_t(12) * n * m, hipMemcpyDeviceToDevice,
                                       h->stream));
                return MDX_OK;
            });
        });
}

int mdx_isf_accumulate_device(mdx_isf_t h, const float *d_pos, int64_t n, int64_t n_frames)
{
    MDX_REQUIRE(h && d_pos, "NULL argument");
    return isf_accumulate(h, n, n_frames, [&](float *d_dst, int64_t done, int64_t nf) -> int {
        MDX_HIP(hipMemcpyAsync(d_dst, d_pos + done * n * 3, size_t(12) * n * nf,
                               hipMemcpyDeviceToDevice, h->stream));
        return MDX_OK;
    });
}

// Frames straight from a trajectory file, in the order listed; index as for
// mdx_sq_accumulate_traj.
int mdx_isf_accumulate_traj(mdx_isf_t h, mdx_traj_t traj, const int64_t *frames, int64_t n_frames,
                            const int32_t *index, int64_t n_index)
{
    MDX_REQUIRE(h && traj, "NULL handle");
    MDX_REQUIRE(n_frames >= 0 && (n_frames == 0 || frames), "bad frame list");
    MDX_TRY(set_device(h->dev));
    Trajectory *t = mdx_traj_internal(traj);
    const int64_t n = index ? n_index : (n_index > 0 ? n_index : t->n_atoms);
    MDX_REQUIRE(index || n <= t->n_atoms, "selection larger than the trajectory");
    const int *d_index = nullptr;
    if (index) {
        for (int64_t i = 0; i < n; ++i)
            if (index[i] < 0 || index[i] >= t->n_atoms)
                return fail(MDX_ERR_INVALID_VALUE, "particle index %d out of range [0, %lld)",
                            index[i], (long long)t->n_atoms);
        MDX_HIP(hipStreamSynchronize(h->stream));
        MDX_TRY(h->d_index.ensure(size_t(4) * n));
        MDX_HIP(hipMemcpy(h->d_index.ptr, index, size_t(4) * n, hipMemcpyHostToDevice));
        d_index = h->d_index.as<int>();
    }
    if (n_frames == 0)
        return MDX_OK;
    // file -> pinned ring -> HBM on the copy stream, a slab ahead of the kernels
    const int64_t slab = std::min<int64_t>(n_frames, isf_slab_frames(h, t->n_atoms));
    return h->pipe.run(
        h->stream, n_frames, slab,
        [&](int b, int64_t f0, int64_t nf) -> int {
            MDX_TRY(h->d_stage[b].ensure(size_t(12) * n * slab));
            TrajSelection sel{d_index, n, h->d_stage[b].as<float>()};
            return t->stage_async(h->dev, h->pipe.copy_stream, frames + f0, nf, &sel, 1);
        },
        [&](int b, int64_t, int64_t nf) -> int {
            const float *d_pos = h->d_stage[b].as<float>();
            return isf_accumulate(h, n, nf, [&](float *d_dst, int64_t done, int64_t m) -> int {
                MDX_HIP(hipMemcpyAsync(d_dst, d_pos + done * n * 3, size_t(12) * n * m, hipMemcpyDeviceToDevice,
                                       h->stream));
                return MDX_OK;
            });
        });
}

int mdx_isf_create(mdx_isf_t *out, int dev, const double *wavevectors, int64_t n_q,
                   const int64_t *group_offsets, int n_groups, const int32_t *pairs, int n_pairs,
                   int n_lags, int incoherent)
{
    MDX_REQUIRE(out && wavevectors && group_offsets && pairs, "NULL argument");
    MDX_REQUIRE(n_q >= 1 && n_q < (int64_t(1) << 30), "n_q out of range");
    MDX_REQUIRE(n_groups >= 1 && n_groups <= 1024, "n_groups out of range");
    MDX_REQUIRE(n_pairs >= 1 && n_pairs <= 65535, "n_pairs out of range");
    MDX_REQUIRE(n_lags >= 1 && n_lags <= 32768, "n_lags out of range");
    for (int g = 0; g < n_groups; ++g)
        MDX_REQUIRE(group_offsets[g + 1] >= group_offsets[g] && group_offsets[0] >= 0,
                    "group offsets must be non-negative and non-decreasing");
    for (int p = 0; p < n_pairs; ++p) {
        int j = pairs[2 * p], k = pairs[2 * p + 1];
        MDX_REQUIRE((j == -1 && k == -1) || (j >= 0 && j < n_groups && k >= 0 && k < n_groups),
                    "pair %d = (%d, %d) is not a pair of groups", p, j, k);
    }
    MDX_TRY(set_device(dev));
    mdx_isf *h = new mdx_isf();
    h->dev = dev;
    h->n_q = n_q;
    h->n_groups = n_groups;
    h->n_pairs = n_pairs;
    h->n_lags = n_lags;
    h->ring_slots = 2 * n_lags;
    h->incoherent = incoherent != 0;
    h->offsets.assign(group_offsets, group_offsets + n_groups + 1);
    h->n_total = group_offsets[n_groups];
    // incoherent slots: all particles for mode=None, else one per group (structure.py:1933-1938);
    // a group without a (g, g) pair keeps zeros, as in the reference
    const bool total_mode = pairs[0] == -1;
    h->n_slots = total_mode ? 1 : n_groups;
    h->ranges.assign(size_t(2) * h->n_slots, 0);
    if (total_mode) {
        h->ranges[0] = group_offsets[0];
        h->ranges[1] = group_offsets[n_groups];
    } else {
        for (int p = 0; p < n_pairs; ++p)
            if (pairs[2 * p] == pairs[2 * p + 1]) {
                int g = pairs[2 * p];
                h->ranges[2 * g] = group_offsets[g];
                h->ranges[2 * g + 1] = group_offsets[g + 1];
            }
    }
    int rc = MDX_OK;
    do {
        if ((rc = stream_acquire(&h->stream)) != MDX_OK)
            break;
        h->timer.stream = h->stream;
        if ((rc = h->d_q.ensure(size_t(24) * n_q)) != MDX_OK) break;
        if ((rc = h->d_offsets.ensure(size_t(8) * (n_groups + 1))) != MDX_OK) break;
        if ((rc = h->d_pairs.ensure(size_t(8) * n_pairs)) != MDX_OK) break;
        if ((rc = h->d_ranges.ensure(size_t(16) * h->n_slots)) != MDX_OK) break;
        if ((rc = h->d_rho_ring.ensure(size_t(16) * h->ring_slots * n_groups * n_q)) != MDX_OK) break;
        if ((rc = h->d_cisf.ensure(size_t(8) * n_lags * n_pairs * n_q)) != MDX_OK) break;
        if (h->incoherent &&
            (rc = h->d_iisf.ensure(size_t(8) * n_lags * h->n_slots * n_q)) != MDX_OK) break;
        if (hipMemcpy(h->d_q.ptr, wavevectors, size_t(24) * n_q, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_offsets.ptr, group_offsets, size_t(8) * (n_groups + 1),
                      hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_pairs.ptr, pairs, size_t(8) * n_pairs, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(h->d_ranges.ptr, h->ranges.data(), size_t(16) * h->n_slots,
                      hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(MDX_ERR_HIP, "upload failed");
            break;
        }
        std::vector<short> trip;
        h->lattice = sq_detect_lattice(wavevectors, n_q, h->lat, trip);
        if (h->lattice) {
            h->lat_lds = size_t(16) * h->lat.tile * (h->lat.R[0] + h->lat.R[1] + h->lat.R[2]);
            if ((rc = h->d_mtrip.ensure(size_t(8) * n_q)) != MDX_OK) break;
            if (hipMemcpy(h->d_mtrip.ptr, trip.data(), size_t(8) * n_q, hipMemcpyHostToDevice) !=
                hipSuccess) {
                rc = fail(MDX_ERR_HIP, "lattice table setup failed");
                break;
            }
            std::vector<SqQuadItem> qitems;
            if (!getenv("MDX_ISF_NO_QUADS") && sq_quad_plan(trip, n_q, h->lat, qitems, h->quad)) {
                if ((rc = h->d_qitems.ensure(sizeof(SqQuadItem) * qitems.size())) != MDX_OK) break;
                if (hipMemcpy(h->d_qitems.ptr, qitems.data(), sizeof(SqQuadItem) * qitems.size(),
                              hipMemcpyHostToDevice) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void *>(isf_incoherent_quads_pick(h->quad.regular_stride)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)h->quad.lds) != hipSuccess ||
                    hipFuncSetAttribute(reinterpret_cast<const void *>(sq_rho_quads_pick(h->quad.regular_stride)),
                                        hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)h->quad.lds) != hipSuccess) {
                    rc = fail(MDX_ERR_HIP, "quad table setup failed");
                    break;
                }
                h->quads = true;
            }
        }
    } while (0);
    if (rc != MDX_OK) {
        mdx_isf_destroy(h);
        return rc;
    }
    *out = h;
    return mdx_isf_reset(h);
}

int mdx_isf_destroy(mdx_isf_t h)
{
    if (!h)
        return MDX_OK;
    (void)hipSetDevice(h->dev);
    if (h->stream)
        (void)hipStreamSynchronize(h->stream);
    h->timer.destroy();
    h->pipe.destroy();      // waits for its copy stream
    for (DeviceBuffer *b : {&h->d_q, &h->d_offsets, &h->d_pairs, &h->d_ranges, &h->d_rho_ring,
                            &h->d_pos_ring, &h->d_cisf, &h->d_iisf, &h->d_part, &h->d_pos_stage,
                            &h->d_index, &h->d_mtrip, &h->d_row_stage, &h->d_qitems, &h->d_rho_parts,
                            &h->d_stage[0], &h->d_stage[1]})
        b->recycle();
    h->mol.recycle();
    if (h->stream)
        stream_release(h->stream);
    delete h;
    return MDX_OK;
}

int mdx_isf_reset(mdx_isf_t h)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipMemsetAsync(h->d_cisf.ptr, 0, size_t(8) * h->n_lags * h->n_pairs * h->n_q, h->stream));
    if (h->incoherent)
        MDX_HIP(hipMemsetAsync(h->d_iisf.ptr, 0, size_t(8) * h->n_lags * h->n_slots * h->n_q,
                               h->stream));
    MDX_HIP(hipStreamSynchronize(h->stream));
    h->frames_seen = 0;
    h->timer.reset();
    return MDX_OK;
}

int mdx_isf_result(mdx_isf_t h, double *cisf, double *iisf)
{
    MDX_REQUIRE(h && cisf, "NULL argument");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    h->timer.collect();
    MDX_HIP(hipMemcpy(cisf, h->d_cisf.ptr, size_t(8) * h->n_lags * h->n_pairs * h->n_q,
                      hipMemcpyDeviceToHost));
    if (iisf) {
        if (!h->incoherent)
            return fail(MDX_ERR_STATE, "the engine was created without the incoherent part");
        MDX_HIP(hipMemcpy(iisf, h->d_iisf.ptr, size_t(8) * h->n_lags * h->n_slots * h->n_q,
                          hipMemcpyDeviceToHost));
    }
    return MDX_OK;
}

int mdx_isf_synchronize(mdx_isf_t h)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    return MDX_OK;
}

int mdx_isf_stats(mdx_isf_t h, int64_t *launches, double *kernel_ms, int64_t *frames)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    h->timer.collect();
    if (launches) *launches = h->timer.launches;
    if (kernel_ms) *kernel_ms = h->timer.total_ms;
    if (frames) *frames = h->frames_seen;
    return MDX_OK;
}

int mdx_isf_enable_timing(mdx_isf_t h, int on)
{
    MDX_REQUIRE(h, "NULL handle");
    h->timer.enabled = on != 0;
    return MDX_OK;
}

}  // extern "C"
