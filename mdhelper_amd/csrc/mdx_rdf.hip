// mdx_rdf.hip — radial pair-distance histogram on gfx950 (MI355X).
//
// Carries reference src/mdhelper/analysis/structure.py:32-104 (radial_histogram:
// capped pair search + exclusion + numpy.histogram) for a batch of frames, i.e.
// the `counts +=` body of RadialDistributionFunction._single_frame (:750-791).
//
// Result contract: for every ordered pair (i, j)
//     fd  = (float)(x_j - x_i)                       float32 subtract
//     s   = (double)(float)(1.0 / L) * (double)fd     exact in double
//     w   = (double)L * (s - round(s))                one rounding
//     rsq = (w_x*w_x + w_y*w_y) + w_z*w_z             no FMA contraction
//     d   = sqrt(rsq);   kept when  r0 - eps < d <= r1  and  i/e0 != j/e1
//     bin = numpy.histogram(d, n_bins, (r0, r1))
// Bin *decisions* are bit-exact with that contract.  Nothing here computes a
// double sqrt: sqrt is monotone, so every comparison `d >= edge` is replaced by
// `rsq >= T(edge)` with T(edge) = the smallest double whose correctly rounded
// sqrt is >= edge (computed on the host, see make_thresholds()).
//
// Three pair kernels share one skeleton (LDS-staged j tile read by broadcast,
// i atoms in registers, per-wave LDS histograms, one uint64 flush per block):
//   EXACT   the contract arithmetic on every pair (fp64 VALU bound);
//   FILTER  float32 distance with a proven error bound; only pairs whose
//           float32 distance falls within that bound of a bin edge (or of the
//           range ends) are re-evaluated with the contract arithmetic;
//   CELL    (mdx_rdf_cell.hip) cell-sorted tiles with culled tile pairs on
//           top of FILTER.
//
// This translation unit is compiled with -ffp-contract=off.

#include "mdx_common.hpp"
#include "mdx_internal.hpp"
#include "mdx_molecules.hpp"
#include "mdx_rdf_device.hpp"
#include "mdx_rdf_cell.hpp"
#include "mdx_traj.hpp"

#include <cmath>

namespace mdx {

// ---------------------------------------------------------------- host helpers

// smallest double t >= 0 with sqrt(t) >= e   (e >= 0; IEEE sqrt is correctly rounded)
static double thresh_ge(double e)
{
    if (!(e > 0.0))
        return 0.0;
    double t = e * e;
    while (std::sqrt(t) >= e && t > 0.0)
        t = std::nextafter(t, 0.0);
    while (std::sqrt(t) < e)
        t = std::nextafter(t, INFINITY);
    return t;
}

// smallest double t with sqrt(t) > e
static double thresh_gt(double e)
{
    if (e < 0.0)
        return 0.0;
    return thresh_ge(std::nextafter(e, INFINITY));
}

}  // namespace mdx

using namespace mdx;

// ------------------------------------------------------------------- kernels

// xyz float32[F][n][3] -> float4[F][n_pad] with w = exclusion tag; pads are NaN.
// Also folds max |coordinate| into *maxabs_bits (non-negative floats order as ints).
__global__ __launch_bounds__(256) void rdf_pack_kernel(const float *__restrict__ pos,
                                                       float4 *__restrict__ out, int n, int n_pad,
                                                       int64_t excl, unsigned *maxabs_bits)
{
    const int frame = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    float m = 0.0f;
    if (a < n_pad) {
        float4 v;
        if (a < n) {
            const float *p = pos + (int64_t(frame) * n + a) * 3;
            v.x = p[0];
            v.y = p[1];
            v.z = p[2];
            int tag = excl > 0 ? int(int64_t(a) / excl) : a;
            v.w = __int_as_float(tag);
            m = fmaxf(fabsf(v.x), fmaxf(fabsf(v.y), fabsf(v.z)));
        } else {
            v.x = v.y = v.z = __int_as_float(0x7fc00000);
            v.w = __int_as_float(-1);
        }
        out[int64_t(frame) * n_pad + a] = v;
    }
    // NaN/inf coordinates propagate into the bound and force the exact path
    unsigned bits = __float_as_uint(m == m ? m : __int_as_float(0x7f800000));
    for (int off = 32; off > 0; off >>= 1)
        bits = max(bits, (unsigned)__shfl_xor((int)bits, off));
    if ((threadIdx.x & 63) == 0 && bits)
        atomicMax(maxabs_bits, bits);
}

// status bit 0: non-orthorhombic or non-positive box
__global__ void rdf_check_boxes_kernel(const float *__restrict__ boxes, int64_t n_frames,
                                       unsigned *status)
{
    int64_t f = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (f >= n_frames)
        return;
    const float *b = boxes + f * 6;
    bool ok = b[0] > 0.f && b[1] > 0.f && b[2] > 0.f && b[3] == 90.f && b[4] == 90.f &&
              b[5] == 90.f;
    if (!ok)
        atomicOr(status, 1u);
}

// GH: the histogram (and the threshold table) stay in global memory — only for bin
// counts whose tables do not fit the LDS budget.
template <int MODE, int IPT, bool PBC, bool EXCL, bool GH>
__global__ __launch_bounds__(256) void rdf_tile_kernel(RdfArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    constexpr int T = 256 * IPT;
    float4 *sj = reinterpret_cast<float4 *>(smem_raw);                       // [T]
    double *sT = reinterpret_cast<double *>(smem_raw + sizeof(float4) * T);  // [n_bins+1]
    unsigned *sh = reinterpret_cast<unsigned *>(sT + (a.n_bins + 1));        // [n_hist][n_bins]
    __shared__ unsigned s_exact;

    const int tid = threadIdx.x;
    const int frame = blockIdx.z + a.frame0;
    const int I = blockIdx.y;
    int J0 = blockIdx.x * a.chunk;
    int J1 = min(J0 + a.chunk, a.nt2);
    if (a.self)
        J0 = max(J0, I);
    if (J0 >= J1)
        return;

    if (!GH) {
        for (int b = tid; b <= a.n_bins; b += 256)
            sT[b] = a.thresh[b];
        for (int b = tid; b < a.n_hist * a.n_bins; b += 256)
            sh[b] = 0u;
    }
    if (tid == 0)
        s_exact = 0u;

    PairCtx<PBC> ctx;
    ctx.init(a, frame);
    unsigned long long *out =
        a.counts + int64_t((blockIdx.x + 3 * blockIdx.y + 7 * blockIdx.z) % a.n_rep) * a.n_bins;
    const double *thr = GH ? a.thresh : sT;
    HistLds hl{sh + (GH ? 0 : ((tid >> 6) % a.n_hist) * a.n_bins)};
    HistGlobal hg{out};

    float4 pi[IPT];
    const float4 *P1 = a.p1 + int64_t(frame) * a.n1p + int64_t(I) * T;
#pragma unroll
    for (int u = 0; u < IPT; ++u)
        pi[u] = P1[tid + 256 * u];

    unsigned n_exact = 0;
    for (int J = J0; J < J1; ++J) {
        const float4 *P2 = a.p2 + int64_t(frame) * a.n2p + int64_t(J) * T;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < IPT; ++u)
            sj[tid + 256 * u] = P2[tid + 256 * u];
        __syncthreads();
        const unsigned w = (a.self && J != I) ? 2u : 1u;
#pragma unroll 4
        for (int jj = 0; jj < T; ++jj) {
            const float4 pj = sj[jj];
#pragma unroll
            for (int u = 0; u < IPT; ++u) {
                if (MODE == MDX_RDF_ALGO_EXACT_F64) {
                    if (GH) pair_exact<PBC, EXCL>(ctx, a, thr, hg, pi[u], pj, w);
                    else pair_exact<PBC, EXCL>(ctx, a, thr, hl, pi[u], pj, w);
                } else {
                    if (GH) pair_filter<PBC, EXCL>(ctx, a, thr, hg, pi[u], pj, w, n_exact);
                    else pair_filter<PBC, EXCL>(ctx, a, thr, hl, pi[u], pj, w, n_exact);
                }
            }
        }
    }
    __syncthreads();
    if (MODE != MDX_RDF_ALGO_EXACT_F64) {
        for (int off = 32; off > 0; off >>= 1)
            n_exact += __shfl_xor((int)n_exact, off);
        if ((tid & 63) == 0 && n_exact)
            atomicAdd(&s_exact, n_exact);
        __syncthreads();
        if (tid == 0 && s_exact)
            atomicAdd(a.exact_counter + rdf_stat_offset(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)),
                      (unsigned long long)s_exact);
    }
    if (!GH) {
        for (int b = tid; b < a.n_bins; b += 256) {
            unsigned long long s = 0;
            for (int h = 0; h < a.n_hist; ++h)
                s += sh[h * a.n_bins + b];
            if (s)
                atomicAdd(out + b, s);
        }
    }
}

// ------------------------------------------------------------------ triclinic cells
//
// Contract (restating MDAnalysis' brute-force triclinic path):
// both sets are moved into the central cell (c, then b, then a axis; double arithmetic on the
// float32 coordinates, result stored as float32), dx = (double)(float)(x_j - x_i), and the
// squared distance is the strictly smallest of the 27 images dx + ix a + iy b + iz c scanned in
// double with ix outermost.  tri: float[frames][9], row-major lower-triangular cell matrix.
__global__ __launch_bounds__(256) void rdf_tri_pack_kernel(const float *__restrict__ pos,
                                                           const float *__restrict__ tri,
                                                           float4 *__restrict__ out, int n,
                                                           int n_pad, int64_t excl)
{
    const int frame = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= n_pad)
        return;
    float4 v;
    if (a < n) {
        const float *p = pos + (int64_t(frame) * n + a) * 3;
        const float *B = tri + int64_t(frame) * 9;
        double r[3] = {(double)p[0], (double)p[1], (double)p[2]};
#pragma unroll
        for (int k = 2; k >= 0; --k) {
            const double s = floor(r[k] / (double)B[4 * k]);
#pragma unroll
            for (int c = 0; c <= k; ++c)
                r[c] -= s * (double)B[3 * k + c];
        }
        v.x = (float)r[0];
        v.y = (float)r[1];
        v.z = (float)r[2];
        v.w = __int_as_float(excl > 0 ? int(int64_t(a) / excl) : a);
    } else {
        v.x = v.y = v.z = __int_as_float(0x7fc00000);
        v.w = __int_as_float(-1);
    }
    out[int64_t(frame) * n_pad + a] = v;
}

struct TriArgs {
    const float4 *p1, *p2;
    const float *tri;
    const double *thresh;
    unsigned long long *counts;
    double t_lo, t_hi;
    float r0f, inv_wf;
    int n1p, n2p, nt1, nt2;
    int n_bins, n_hist, n_rep;
    int chunk, self, frame0;
};

template <bool EXCL, bool GH>
__global__ __launch_bounds__(256) void rdf_tri_tile_kernel(TriArgs a)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float4 *sj = reinterpret_cast<float4 *>(smem_raw);                         // [256]
    double *sT = reinterpret_cast<double *>(smem_raw + sizeof(float4) * 256);  // [n_bins+1]
    unsigned *sh = reinterpret_cast<unsigned *>(sT + (a.n_bins + 1));          // [n_hist][n_bins]

    const int tid = threadIdx.x;
    const int frame = blockIdx.z + a.frame0;
    const int I = blockIdx.y;
    int J0 = blockIdx.x * a.chunk;
    int J1 = min(J0 + a.chunk, a.nt2);
    if (a.self)
        J0 = max(J0, I);
    if (J0 >= J1)
        return;
    if (!GH) {
        for (int b = tid; b <= a.n_bins; b += 256)
            sT[b] = a.thresh[b];
        for (int b = tid; b < a.n_hist * a.n_bins; b += 256)
            sh[b] = 0u;
    }
    const float *Bf = a.tri + int64_t(frame) * 9;
    const double b00 = Bf[0], b10 = Bf[3], b11 = Bf[4], b20 = Bf[6], b21 = Bf[7], b22 = Bf[8];
    unsigned long long *out =
        a.counts + int64_t((blockIdx.x + 3 * blockIdx.y + 7 * blockIdx.z) % a.n_rep) * a.n_bins;
    const double *thr = GH ? a.thresh : sT;
    HistLds hl{sh + (GH ? 0 : ((tid >> 6) % a.n_hist) * a.n_bins)};
    HistGlobal hg{out};
    const float4 pi = a.p1[int64_t(frame) * a.n1p + int64_t(I) * 256 + tid];

    for (int J = J0; J < J1; ++J) {
        __syncthreads();
        sj[tid] = a.p2[int64_t(frame) * a.n2p + int64_t(J) * 256 + tid];
        __syncthreads();
        const unsigned w = (a.self && J != I) ? 2u : 1u;
        for (int jj = 0; jj < 256; ++jj) {
            const float4 pj = sj[jj];
            const double dx = (double)(pj.x - pi.x), dy = (double)(pj.y - pi.y),
                         dz = (double)(pj.z - pi.z);
            double best = 1.0e300;
#pragma unroll
            for (int ix = -1; ix < 2; ++ix) {
                const double rx = dx + b00 * (double)ix;
#pragma unroll
                for (int iy = -1; iy < 2; ++iy) {
                    const double ry0 = rx + b10 * (double)iy;
                    const double ry1 = dy + b11 * (double)iy;
#pragma unroll
                    for (int iz = -1; iz < 2; ++iz) {
                        const double rz0 = ry0 + b20 * (double)iz;
                        const double rz1 = ry1 + b21 * (double)iz;
                        const double rz2 = dz + b22 * (double)iz;
                        const double dsq = (rz0 * rz0 + rz1 * rz1) + rz2 * rz2;
                        best = dsq < best ? dsq : best;
                    }
                }
            }
            // NaN (padding) fails both comparisons
            bool in = (best >= a.t_lo) && (best < a.t_hi);
            if (EXCL)
                in = in && (__float_as_int(pi.w) != __float_as_int(pj.w));
            if (in) {
                const int k = rdf_bin_exact(best, thr, a.n_bins, a.r0f, a.inv_wf);
                if (GH) hg.add(k, w);
                else hl.add(k, w);
            }
        }
    }
    __syncthreads();
    if (!GH) {
        for (int b = tid; b < a.n_bins; b += 256) {
            unsigned long long s = 0;
            for (int h = 0; h < a.n_hist; ++h)
                s += sh[h * a.n_bins + b];
            if (s)
                atomicAdd(out + b, s);
        }
    }
}

__global__ void rdf_reduce_kernel(const unsigned long long *__restrict__ rep, int n_rep, int n_bins,
                                  unsigned long long *__restrict__ total)
{
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_bins)
        return;
    unsigned long long s = 0;
    for (int r = 0; r < n_rep; ++r)
        s += rep[int64_t(r) * n_bins + b];
    total[b] = s;
}

// --------------------------------------------------------------------- handle

struct mdx_rdf {
    int dev = 0;
    hipStream_t stream = nullptr;
    int n_bins = 0;
    std::vector<double> edges;
    std::vector<double> thresh;
    double t_lo = 0, t_hi = 0;
    int64_t excl1 = 0, excl2 = 0;
    int algo = MDX_RDF_ALGO_AUTO;
    int n_rep = 32;
    DeviceBuffer d_thresh, d_counts, d_total, d_pack1, d_pack2, d_misc, d_tri;
    DeviceBuffer d_stats;   // RDF_STAT_SHARDS x RDF_STAT_STRIDE 64-bit words (mdx_rdf_device.hpp)
    DeviceBuffer d_work;    // work counters of the persistent pair kernel (one line per XCD)
    const void *occ_kernel = nullptr;   // occupancy of the persistent pair kernel: last kernel / LDS size asked for
    size_t occ_lds = 0;
    int64_t occ_blocks_per_xcd = 1;
    // what mdx_rdf_debug_sorted needs to find the most recent slab: its set of sorted copies, whether the
    // sorted originals were materialised, its frame count
    size_t last_offset = 0;   // float4 elements from the start of d_pw1 / d_po1 to the slab
    int64_t last_frames = 0, last_n_pad = 0;
    bool last_lazy = false;
    // optional centre-of-mass stage per set: incoming rows are particles grouped into molecules
    MoleculeStage grouping[2];
    // drop_axis (2-D mode, structure.py:761-770): coordinate zeroed, box length -> max(lx, ly, lz)
    int drop_axis = -1;
    DeviceBuffer d_drop[2], d_drop_box;
    // host-buffer entry point: double-buffered staging, copy stream, hand-over events
    DeviceBuffer d_stage1[2], d_stage2[2], d_boxes[2], d_index[2];
    DeviceBuffer d_rawslab[2];   // trajectory files: a slab of raw frames between the copy and the unpack kernel
    StagePipeline pipe;
    DeviceBuffer d_pw1, d_po1, d_bb1, d_pw2, d_po2, d_bb2;   // cell path: sorted copies + tile boxes
    DeviceBuffer d_bb16_1, d_bb16_2;                         // boxes of the CELL_CHUNK-particle chunks
    // cell path: the sort of slab k + 1 runs on its own stream beside the pair kernel of slab k (two sets of the
    // sorted copies); events hand the sets back and forth
    StreamTimer timer;
    int64_t pairs_evaluated = 0;    // ordered pair space covered: frames * n1 * n2
    int64_t pairs_bruteforce = 0;   // distance evaluations executed by the brute-force tiles
    bool reduced_global = false;   // counts replica 0 holds an all-reduced total
};

static int launch_tiles(mdx_rdf *h, RdfArgs &a, int mode, int ipt, bool pbc, bool excl,
                        int64_t n_frames)
{
    const int T = 256 * ipt;
    // LDS: j tile + thresholds + per-wave histograms
    size_t base = sizeof(float4) * T + sizeof(double) * (h->n_bins + 1);
    const size_t lds_budget = 64 * 1024;   // keep >= 2 blocks per CU resident
    int n_hist = 4;
    while (n_hist > 1 && base + size_t(n_hist) * h->n_bins * 4 > lds_budget)
        n_hist >>= 1;
    size_t lds = base + size_t(n_hist) * h->n_bins * 4;
    const bool gh = lds > lds_budget;   // tables too large for LDS: global histogram
    if (gh)
        lds = sizeof(float4) * T;
    a.n_hist = n_hist;
    a.chunk = a.self ? 4 : 8;
    void (*kern)(RdfArgs) = nullptr;
#define MDX_PICK(M, I, P, E) \
    kern = gh ? rdf_tile_kernel<M, I, P, E, true> : rdf_tile_kernel<M, I, P, E, false>
#define MDX_PICK_PE(M, I)                         \
    do {                                          \
        if (pbc && excl) MDX_PICK(M, I, true, true);   \
        else if (pbc) MDX_PICK(M, I, true, false);     \
        else if (excl) MDX_PICK(M, I, false, true);    \
        else MDX_PICK(M, I, false, false);             \
    } while (0)
    if (mode == MDX_RDF_ALGO_EXACT_F64) {
        if (ipt == 1) MDX_PICK_PE(MDX_RDF_ALGO_EXACT_F64, 1);
        else MDX_PICK_PE(MDX_RDF_ALGO_EXACT_F64, 2);
    } else {
        if (ipt == 1) MDX_PICK_PE(MDX_RDF_ALGO_FILTER_F32, 1);
        else MDX_PICK_PE(MDX_RDF_ALGO_FILTER_F32, 2);
    }
#undef MDX_PICK_PE
#undef MDX_PICK
    if (lds > 48 * 1024)
        MDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t ev = h->timer.begin();
    for (int64_t f0 = 0; f0 < n_frames; f0 += 32768) {
        int64_t nf = std::min<int64_t>(32768, n_frames - f0);
        a.frame0 = (int)f0;
        dim3 grid((unsigned)ceil_div(a.nt2, a.chunk), (unsigned)a.nt1, (unsigned)nf);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, h->stream, a);
    }
    h->timer.end(ev);
    MDX_HIP(hipGetLastError());
    return MDX_OK;
}

// Cell-sorted path (mdx_rdf_cell.hpp): sort + tile boxes per frame, then the culled pair kernel.
// d_tri != nullptr: triclinic frames, d_tri = their cell matrices [n_frames][9] (d_boxes unused)
static int accumulate_cell(mdx_rdf *h, const float *d_pos1, int64_t n1, const float *d_pos2,
                           int64_t n2, const float *d_boxes, int64_t n_frames, bool self, bool excl,
                           const float *d_tri = nullptr)
{
    const bool tri = d_tri != nullptr;
    const int64_t n1p = ceil_div(n1, 128) * 128, n2p = ceil_div(n2, 128) * 128;
    // per frame: wrapped + original float4 copies, one box per 64 and per 16 particles
    // (two-particle chunks take their boxes from the staged rows: no chunk-box array)
    constexpr int64_t CHUNK_BOX_BYTES = CELL_CHUNK >= 4 ? 32 / CELL_CHUNK : 0;
    const int64_t per_frame = (32 + 1 + CHUNK_BOX_BYTES) * (n1p + (self ? 0 : n2p));
    // (MDX_RDF_SLAB_BYTES: test hook, so that small inputs run through several slabs and both sets)
    const char *slab_env = getenv("MDX_RDF_SLAB_BYTES");
    // (2.5 GiB = 2 048 frames at C2: the persistent kernel ends on a tail of about one item per block and the sort
    // between two launches is serial, so long launches pay: 1 024 frames 45.3 k, 2 048 45.6–45.8 k, 4 096 45.7 k frames/s)
    const int64_t slab_bytes = slab_env ? std::max<int64_t>(1, atoll(slab_env)) : (int64_t(5) << 29);
    int64_t slab = std::max<int64_t>(1, slab_bytes / per_frame);
    slab = std::min<int64_t>(slab, 32768);
    // whole rounds of the sort kernel: one 1 024-thread block per frame, one block per CU (65 VGPRs) —
    // 993 frames were 3.9 rounds of 256 CUs and took 4
    if (slab >= 256)
        slab -= slab % 256;
    slab = std::min<int64_t>(slab, n_frames);
    // One set of the sorted copies; sort and pair kernel follow each other on the handle's stream.  (The sort of
    // slab k + 1 on a stream of its own beside the pair kernel of slab k — the form of round 2 — lost once the pair
    // blocks became persistent: they hold every wave slot until their last items, the sort trickles in behind
    // retiring blocks and the next pair kernel waits for it anyway: 44.1 k against 45.4 k frames/s at C2(i),
    // NOTES.md round 3.  The second set, its stream and its events are gone.)
    // exclusion 0 or 1: a particle's tag is its row, and the exact path reads the frames as they came in —
    // no sorted copy of the original coordinates (half of the sort kernel's scattered stores)
    const bool lazy_orig = !tri && h->excl1 <= 1 && h->excl2 <= 1;
    const size_t e_p1 = size_t(n1p) * slab, e_b1 = size_t(n1p / 64) * 2 * slab;
    const size_t e_c1 = CELL_CHUNK >= 4 ? size_t(n1p / CELL_CHUNK) * 2 * slab : 16;
    const size_t e_p2 = size_t(n2p) * slab, e_b2 = size_t(n2p / 64) * 2 * slab;
    const size_t e_c2 = CELL_CHUNK >= 4 ? size_t(n2p / CELL_CHUNK) * 2 * slab : 16;
    MDX_TRY(h->d_pw1.ensure(16 * e_p1));
    if (!lazy_orig)
        MDX_TRY(h->d_po1.ensure(16 * e_p1));
    MDX_TRY(h->d_bb1.ensure(16 * e_b1));
    MDX_TRY(h->d_bb16_1.ensure(16 * e_c1));
    if (!self) {
        MDX_TRY(h->d_bb16_2.ensure(16 * e_c2));
        MDX_TRY(h->d_pw2.ensure(16 * e_p2));
        if (!lazy_orig)
            MDX_TRY(h->d_po2.ensure(16 * e_p2));
        MDX_TRY(h->d_bb2.ensure(16 * e_b2));
    }
    unsigned *d_misc = h->d_misc.as<unsigned>();
    if (!tri)
        hipLaunchKernelGGL(rdf_check_boxes_kernel, dim3((unsigned)ceil_div(n_frames, 256)), dim3(256),
                           0, h->stream, d_boxes, n_frames, d_misc + 1);

    // LDS of the pair kernel: 4 wave slabs + thresholds + the histogram in n_hist interleaved replicas
    // (HistLdsRep): 8 while seven blocks of a CU still fit (the kernel's static LDS is 8.4 KB), fewer for
    // long bin tables
    size_t base = sizeof(float4) * 256 + sizeof(double) * (h->n_bins + 1);
    const size_t lds_budget = 64 * 1024;
    const size_t lds_six_blocks = 13 * 1024;   // seven blocks of a CU: (160 KB / 7) - 8.4 KB static
    int n_hist = 8;
    while (n_hist > 4 && base + size_t(n_hist) * cell_hist_stride(h->n_bins) * 4 > lds_six_blocks)
        n_hist >>= 1;
    while (n_hist > 1 && base + size_t(n_hist) * cell_hist_stride(h->n_bins) * 4 > lds_budget)
        n_hist >>= 1;
    size_t lds = base + size_t(n_hist) * cell_hist_stride(h->n_bins) * 4;
    const bool gh = lds > lds_budget;
    if (gh)
        lds = sizeof(float4) * 256;
    const bool lower = h->edges.front() > 0.0;
    void (*kern)(CellArgs) = nullptr;
#define MDX_CELL_PICK(E, L)                                                                     \
    kern = tri ? (gh ? rdf_cell_pair_kernel<E, L, 1, true> : rdf_cell_pair_kernel<E, L, 0, true>) \
               : (gh ? rdf_cell_pair_kernel<E, L, 1> : rdf_cell_pair_kernel<E, L, 0>)
    if (excl && lower) MDX_CELL_PICK(true, true);
    else if (excl) MDX_CELL_PICK(true, false);
    else if (lower) MDX_CELL_PICK(false, true);
    else MDX_CELL_PICK(false, false);
#undef MDX_CELL_PICK
    // resident blocks per XCD (32 CUs each) of this kernel at this LDS size (asked once per kernel and size)
    if (h->occ_kernel != reinterpret_cast<const void *>(kern) || h->occ_lds != lds) {
        int per_cu = 0, cus = 0;
        MDX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), 256, lds));
        MDX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->dev));
        h->occ_kernel = reinterpret_cast<const void *>(kern);
        h->occ_lds = lds;
        h->occ_blocks_per_xcd = std::max<int64_t>(1, int64_t(std::max(per_cu, 1)) * std::max(cus, 8) / 8);
    }
    const int64_t blocks_per_xcd = h->occ_blocks_per_xcd;

    hipStream_t s_sort = h->stream;
    for (int64_t f0 = 0; f0 < n_frames; f0 += slab) {
        const int64_t nf = std::min(slab, n_frames - f0);
        float4 *pw1 = h->d_pw1.as<float4>();
        float4 *po1 = lazy_orig ? nullptr : h->d_po1.as<float4>();
        float4 *bb1 = h->d_bb1.as<float4>(), *bc1 = h->d_bb16_1.as<float4>();
        float4 *pw2 = self ? pw1 : h->d_pw2.as<float4>();
        float4 *po2 = self ? po1 : (lazy_orig ? nullptr : h->d_po2.as<float4>());
        float4 *bb2 = self ? bb1 : h->d_bb2.as<float4>();
        float4 *bc2 = self ? bc1 : h->d_bb16_2.as<float4>();
        // the batch's largest |coordinate| (the filter's error bound grows with it) is folded by the sort
        unsigned *d_maxabs = d_misc;
        h->last_offset = 0;
        h->last_frames = nf;
        h->last_n_pad = n1p;
        h->last_lazy = lazy_orig;
        {
            // (Measured and dropped, round 3: the sort as a GATHER — slots noted per particle in LDS, rows written
            // in slot order as whole lines — 0.66 ms per 1 000 frames against 0.74–0.82, +0.5 % on the step, but its
            // 12-byte reads scattered over frames that are no longer in L2 fetch 4.0 MB per frame where the scatter's
            // partial-sector stores cost 1.1: 6.5 MB per frame in all against 4.4.  Neither two blocks per CU (the
            // kernel sits at 65 VGPRs: one over), float32 cell keys, a DPP box reduction nor a two-barrier scan moved
            // the kernel's time: all 256 blocks of a round read, then write, in step — 25 µs counting from HBM, 81 µs
            // writing rows — and the memory system sets the pace.)
            auto sort = tri ? rdf_cell_sort_kernel<true> : rdf_cell_sort_kernel<false>;
            const float *cells = tri ? d_tri + f0 * 9 : d_boxes + f0 * 6;
            hipLaunchKernelGGL(sort, dim3((unsigned)nf), dim3(SORT_THREADS), 0, s_sort,
                               d_pos1 + f0 * n1 * 3, cells, (int)n1, (int)n1p, excl ? h->excl1 : 0,
                               pw1, po1, bb1, bc1, d_maxabs);
            if (!self)
                hipLaunchKernelGGL(sort, dim3((unsigned)nf), dim3(SORT_THREADS), 0, s_sort,
                                   d_pos2 + f0 * n2 * 3, cells, (int)n2, (int)n2p,
                                   excl ? h->excl2 : 0, pw2, po2, bb2, bc2, d_maxabs);
        }
        CellArgs a{};
        a.pw1 = pw1;
        a.po1 = po1;
        a.bb1 = bb1;
        a.pw2 = pw2;
        a.po2 = po2;
        a.bb2 = bb2;
        a.bb16_2 = bc2;
        a.in1 = lazy_orig ? d_pos1 + f0 * n1 * 3 : nullptr;
        a.in2 = lazy_orig ? d_pos2 + f0 * n2 * 3 : nullptr;
        a.n1_in = (int)n1;
        a.n2_in = (int)n2;
        a.tags_everywhere = (self && h->excl1 == 1 && h->excl2 == 1) ? 0 : 1;
        a.boxes = tri ? nullptr : d_boxes + f0 * 6;
        a.tri = tri ? d_tri + f0 * 9 : nullptr;
        a.thresh = h->d_thresh.as<double>();
        a.counts = h->d_counts.as<unsigned long long>();
        a.maxabs_bits = d_maxabs;
        a.exact_counter = h->d_stats.as<unsigned long long>();
        a.tilepair_counter = a.exact_counter + 1;
        a.clock_counter = h->timer.enabled ? a.exact_counter + 3 : nullptr;
        a.t_lo = h->t_lo;
        a.t_hi = h->t_hi;
        a.r0 = h->edges.front();
        a.r1 = h->edges.back();
        a.n1p = (int)n1p;
        a.n2p = (int)n2p;
        a.n_bins = h->n_bins;
        a.n_hist = n_hist;
        a.n_rep = h->n_rep;
        a.self = self ? 1 : 0;
        // persistent blocks: what the chip holds at once (blocks per CU from the kernel's own occupancy), a
        // multiple of 8 so that every XCD gets the same number; fewer when there is less work than that
        a.work = h->d_work.as<unsigned>();
        {
            const char *fu = getenv("MDX_RDF_LDS_FLUSH_UNITS");   // test hook: flush the LDS bins (much) more often
            a.flush_units = fu ? (unsigned)std::min<long long>(std::max<long long>(atoll(fu), 1), 1 << 18) : (1u << 18);
        }
        const int64_t tiles = n1p / 128;
        hipEvent_t ev = h->timer.begin();
        // (frames per launch: the item index is 32 bits wide)
        const int64_t launch_frames = std::max<int64_t>(8, std::min<int64_t>(32768, ((int64_t(1) << 30) / tiles) * 8));
        for (int64_t g0 = 0; g0 < nf; g0 += launch_frames) {
            a.frame0 = (int)g0;
            a.n_frames = (int)std::min<int64_t>(launch_frames, nf - g0);
            a.spread = a.n_frames < 8 ? 1 : 0;
            const int64_t items_per_xcd = a.spread ? ceil_div(int64_t(a.n_frames) * tiles, 8)
                                                   : ceil_div(a.n_frames, 8) * tiles;
            const unsigned grid = 8u * (unsigned)std::min<int64_t>(blocks_per_xcd, items_per_xcd);
            MDX_HIP(hipMemsetAsync(h->d_work.ptr, 0, CELL_WORK_BYTES, h->stream));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, h->stream, a);
        }
        h->timer.end(ev);
        MDX_HIP(hipGetLastError());
    }
    h->pairs_evaluated += n_frames * n1 * n2;
    return MDX_OK;
}

static int accumulate_ortho(mdx_rdf *h, const float *d_pos1, int64_t n1, const float *d_pos2,
                            int64_t n2, const float *d_boxes, int64_t n_frames)
{
    if (n_frames == 0 || n1 == 0 || n2 == 0)
        return MDX_OK;
    if (h->reduced_global)
        return fail(MDX_ERR_STATE, "handle holds all-reduced counts; call mdx_rdf_reset() first");
    const bool same = (d_pos2 == nullptr || (d_pos2 == d_pos1 && n2 == n1));
    if (d_pos2 == nullptr) {
        d_pos2 = d_pos1;
        n2 = n1;
    }
    const bool excl = h->excl1 > 0;
    const bool self = same && (!excl || h->excl1 == h->excl2);
    MDX_REQUIRE(n1 < (int64_t(1) << 30) && n2 < (int64_t(1) << 30), "too many particles");

    // AUTO: the cell-sorted kernel whenever there is a periodic box — at every size.  Below 1 024 particles the
    // tiles cannot be culled (the cutoff spans the cell), but its hot step (13 VALU instructions + the
    // hand-placed tail per 64 evaluations, symmetric tile pairs evaluated once) is what the brute-force tile kernel
    // spends 48 on: measured 1.1 x (16 particles) .. 3.4 x (1 000 particles, range to L/2: 0.88 -> 3.0 M frames/s)
    // the frames per second, counts identical (scripts/diag/rdf_small_sweep.py, round 4).  Without a box: the
    // brute-force tiles with the float32 filter.
    int algo = h->algo;
    if (algo == MDX_RDF_ALGO_AUTO)
        algo = d_boxes ? MDX_RDF_ALGO_CELL : MDX_RDF_ALGO_FILTER_F32;
    if (algo == MDX_RDF_ALGO_CELL && !d_boxes)
        algo = MDX_RDF_ALGO_FILTER_F32;   // the cell grid is defined by the periodic box
    if (algo == MDX_RDF_ALGO_CELL)
        return accumulate_cell(h, d_pos1, n1, d_pos2, n2, d_boxes, n_frames, self, excl);
    const int ipt = (std::max(n1, n2) >= 2048) ? 2 : 1;
    const int T = 256 * ipt;
    const int64_t n1p = ceil_div(n1, T) * T, n2p = ceil_div(n2, T) * T;

    // frames are processed in slabs so the packed copies stay bounded
    const int64_t slab_bytes = int64_t(1) << 30;
    int64_t slab = std::max<int64_t>(1, slab_bytes / (int64_t(16) * (n1p + (self ? 0 : n2p))));
    slab = std::min<int64_t>(slab, 32768);
    MDX_TRY(h->d_pack1.ensure(size_t(16) * n1p * std::min(slab, n_frames)));
    if (!self)
        MDX_TRY(h->d_pack2.ensure(size_t(16) * n2p * std::min(slab, n_frames)));

    unsigned *d_misc = h->d_misc.as<unsigned>();   // [0] maxabs bits, [1] status, [2..3] exact counter
    if (d_boxes) {
        hipLaunchKernelGGL(rdf_check_boxes_kernel, dim3((unsigned)ceil_div(n_frames, 256)),
                           dim3(256), 0, h->stream, d_boxes, n_frames, d_misc + 1);
    }
    for (int64_t f0 = 0; f0 < n_frames; f0 += slab) {
        const int64_t nf = std::min(slab, n_frames - f0);
        hipLaunchKernelGGL(rdf_pack_kernel, dim3((unsigned)(n1p / 256), (unsigned)nf), dim3(256), 0,
                           h->stream, d_pos1 + f0 * n1 * 3, h->d_pack1.as<float4>(), (int)n1,
                           (int)n1p, excl ? h->excl1 : 0, d_misc);
        if (!self)
            hipLaunchKernelGGL(rdf_pack_kernel, dim3((unsigned)(n2p / 256), (unsigned)nf), dim3(256),
                               0, h->stream, d_pos2 + f0 * n2 * 3, h->d_pack2.as<float4>(), (int)n2,
                               (int)n2p, excl ? h->excl2 : 0, d_misc);
        RdfArgs a{};
        a.p1 = h->d_pack1.as<float4>();
        a.p2 = self ? a.p1 : h->d_pack2.as<float4>();
        a.n1p = (int)n1p;
        a.n2p = (int)n2p;
        a.nt1 = (int)(n1p / T);
        a.nt2 = (int)(n2p / T);
        a.self = self ? 1 : 0;
        a.boxes = d_boxes ? d_boxes + f0 * 6 : nullptr;
        a.n_bins = h->n_bins;
        a.thresh = h->d_thresh.as<double>();
        a.t_lo = h->t_lo;
        a.t_hi = h->t_hi;
        a.r0 = h->edges.front();
        a.r1 = h->edges.back();
        a.counts = h->d_counts.as<unsigned long long>();
        a.n_rep = h->n_rep;
        a.maxabs_bits = d_misc;
        a.exact_counter = h->d_stats.as<unsigned long long>();
        MDX_TRY(launch_tiles(h, a, algo, ipt, d_boxes != nullptr, excl, nf));
        const int64_t tile_pairs = self ? int64_t(a.nt1) * (a.nt1 + 1) / 2 : int64_t(a.nt1) * a.nt2;
        h->pairs_bruteforce += nf * tile_pairs * T * T;
    }
    h->pairs_evaluated += n_frames * n1 * n2;
    return MDX_OK;
}

// cell matrix of (lx, ly, lz, alpha, beta, gamma): float64 arithmetic through libm, float32
// entries, exact zeros for right angles
static void tri_vectors(const float *box6, float *B)
{
    const double lx = box6[0], ly = box6[1], lz = box6[2];
    const double deg = 3.14159265358979323846 / 180.0;
    const double ca = box6[3] == 90.0f ? 0.0 : std::cos((double)box6[3] * deg);
    const double cb = box6[4] == 90.0f ? 0.0 : std::cos((double)box6[4] * deg);
    const double cg = box6[5] == 90.0f ? 0.0 : std::cos((double)box6[5] * deg);
    const double sg = box6[5] == 90.0f ? 1.0 : std::sin((double)box6[5] * deg);
    for (int i = 0; i < 9; ++i)
        B[i] = 0.0f;
    B[0] = (float)lx;
    B[3] = (float)(ly * cg);
    B[4] = (float)(ly * sg);
    const double cx = lz * cb;
    const double cy = lz * (ca - cb * cg) / sg;
    B[6] = (float)cx;
    B[7] = (float)cy;
    B[8] = (float)std::sqrt(lz * lz - cx * cx - cy * cy);
}

static inline bool box_is_ortho(const float *b) { return b[3] == 90.f && b[4] == 90.f && b[5] == 90.f; }

// Triclinic frames: brute-force tiles, 27-image search in double on every pair.
static int accumulate_triclinic(mdx_rdf *h, const float *d_pos1, int64_t n1, const float *d_pos2,
                                int64_t n2, const float *h_boxes, int64_t n_frames)
{
    if (n_frames == 0 || n1 == 0 || n2 == 0)
        return MDX_OK;
    if (h->reduced_global)
        return fail(MDX_ERR_STATE, "handle holds all-reduced counts; call mdx_rdf_reset() first");
    const bool same = (d_pos2 == nullptr || (d_pos2 == d_pos1 && n2 == n1));
    if (d_pos2 == nullptr) {
        d_pos2 = d_pos1;
        n2 = n1;
    }
    const bool excl = h->excl1 > 0;
    const bool self = same && (!excl || h->excl1 == h->excl2);
    MDX_REQUIRE(n1 < (int64_t(1) << 30) && n2 < (int64_t(1) << 30), "too many particles");
    std::vector<float> tri(size_t(9) * n_frames);
    for (int64_t f = 0; f < n_frames; ++f) {
        const float *b = h_boxes + 6 * f;
        if (!(b[0] > 0.f && b[1] > 0.f && b[2] > 0.f && b[3] > 0.f && b[3] < 180.f && b[4] > 0.f &&
              b[4] < 180.f && b[5] > 0.f && b[5] < 180.f))
            return fail(MDX_ERR_INVALID_VALUE, "frame %lld: invalid cell (%g %g %g %g %g %g)",
                        (long long)f, b[0], b[1], b[2], b[3], b[4], b[5]);
        tri_vectors(b, tri.data() + 9 * f);
        if (!(tri[9 * f + 8] > 0.f) || !(tri[9 * f + 4] > 0.f))
            return fail(MDX_ERR_INVALID_VALUE, "frame %lld: the cell angles do not span a volume",
                        (long long)f);
    }
    // the previous call's kernels may still read the cell matrices
    MDX_HIP(hipStreamSynchronize(h->stream));
    MDX_TRY(h->d_tri.ensure(size_t(36) * n_frames));
    MDX_HIP(hipMemcpyAsync(h->d_tri.ptr, tri.data(), size_t(36) * n_frames, hipMemcpyHostToDevice,
                           h->stream));
    MDX_HIP(hipStreamSynchronize(h->stream));   // `tri` leaves scope

    // Culled cell-sorted kernel (27 tile images, float32 filter, 27-image contract for the
    // undecided pairs) when the range ends below half the smallest cell height in every frame:
    // then at most one image of a pair can come within the cut.  Otherwise (or for small
    // systems, or algo = exact / filter) the brute-force 27-image kernel below.
    bool culled = h->algo == MDX_RDF_ALGO_AUTO || h->algo == MDX_RDF_ALGO_CELL;
    culled = culled && std::max(n1, n2) >= 1024 && !getenv("MDX_RDF_TRI_BRUTE");
    for (int64_t f = 0; culled && f < n_frames; ++f) {
        const float *B = tri.data() + 9 * f;
        const double b00 = B[0], b10 = B[3], b11 = B[4], b20 = B[6], b21 = B[7], b22 = B[8];
        const double vol = b00 * b11 * b22;
        const double bcx = b11 * b22, bcy = -b10 * b22, bcz = b10 * b21 - b11 * b20;
        const double hx = vol / std::sqrt(bcx * bcx + bcy * bcy + bcz * bcz);
        const double hy = b11 * b22 / std::sqrt(b22 * b22 + b21 * b21);
        const double r_in = 0.5 * std::min(hx, std::min(hy, b22));
        culled = h->edges.back() * 1.02 + 1e-3 < r_in;
    }
    if (culled)
        return accumulate_cell(h, d_pos1, n1, d_pos2, n2, nullptr, n_frames, self,
                               excl, h->d_tri.as<float>());

    const int64_t n1p = ceil_div(n1, 256) * 256, n2p = ceil_div(n2, 256) * 256;
    int64_t slab = std::max<int64_t>(1, (int64_t(1) << 30) / (int64_t(16) * (n1p + (self ? 0 : n2p))));
    slab = std::min<int64_t>(std::min<int64_t>(slab, 32768), n_frames);
    MDX_TRY(h->d_pack1.ensure(size_t(16) * n1p * slab));
    if (!self)
        MDX_TRY(h->d_pack2.ensure(size_t(16) * n2p * slab));

    size_t base = sizeof(float4) * 256 + sizeof(double) * (h->n_bins + 1);
    int n_hist = 4;
    while (n_hist > 1 && base + size_t(n_hist) * h->n_bins * 4 > size_t(64) * 1024)
        n_hist >>= 1;
    size_t lds = base + size_t(n_hist) * h->n_bins * 4;
    const bool gh = lds > size_t(64) * 1024;
    if (gh)
        lds = sizeof(float4) * 256;
    void (*kern)(TriArgs) = excl ? (gh ? rdf_tri_tile_kernel<true, true> : rdf_tri_tile_kernel<true, false>)
                                 : (gh ? rdf_tri_tile_kernel<false, true> : rdf_tri_tile_kernel<false, false>);
    if (lds > 48 * 1024)
        MDX_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const double width = (h->edges.back() - h->edges.front()) / h->n_bins;
    for (int64_t f0 = 0; f0 < n_frames; f0 += slab) {
        const int64_t nf = std::min(slab, n_frames - f0);
        const float *d_tri = h->d_tri.as<float>() + f0 * 9;
        hipLaunchKernelGGL(rdf_tri_pack_kernel, dim3((unsigned)(n1p / 256), (unsigned)nf), dim3(256),
                           0, h->stream, d_pos1 + f0 * n1 * 3, d_tri, h->d_pack1.as<float4>(),
                           (int)n1, (int)n1p, excl ? h->excl1 : 0);
        if (!self)
            hipLaunchKernelGGL(rdf_tri_pack_kernel, dim3((unsigned)(n2p / 256), (unsigned)nf),
                               dim3(256), 0, h->stream, d_pos2 + f0 * n2 * 3, d_tri,
                               h->d_pack2.as<float4>(), (int)n2, (int)n2p, excl ? h->excl2 : 0);
        TriArgs a{};
        a.p1 = h->d_pack1.as<float4>();
        a.p2 = self ? a.p1 : h->d_pack2.as<float4>();
        a.tri = d_tri;
        a.thresh = h->d_thresh.as<double>();
        a.counts = h->d_counts.as<unsigned long long>();
        a.t_lo = h->t_lo;
        a.t_hi = h->t_hi;
        a.r0f = (float)h->edges.front();
        a.inv_wf = (float)(1.0 / width);
        a.n1p = (int)n1p;
        a.n2p = (int)n2p;
        a.nt1 = (int)(n1p / 256);
        a.nt2 = (int)(n2p / 256);
        a.n_bins = h->n_bins;
        a.n_hist = n_hist;
        a.n_rep = h->n_rep;
        a.chunk = self ? 4 : 8;
        a.self = self ? 1 : 0;
        a.frame0 = 0;
        hipEvent_t ev = h->timer.begin();
        dim3 grid((unsigned)ceil_div(a.nt2, a.chunk), (unsigned)a.nt1, (unsigned)nf);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, h->stream, a);
        h->timer.end(ev);
        MDX_HIP(hipGetLastError());
        const int64_t tile_pairs = self ? int64_t(a.nt1) * (a.nt1 + 1) / 2 : int64_t(a.nt1) * a.nt2;
        h->pairs_bruteforce += nf * tile_pairs * 256 * 256;
    }
    h->pairs_evaluated += n_frames * n1 * n2;
    return MDX_OK;
}

// Frames are dispatched by cell type: orthorhombic runs take the filter / cell-sorted kernels,
// triclinic runs the 27-image kernel.  h_boxes: the same boxes on the host, or nullptr (then
// they are copied back from d_boxes — 24 bytes per frame and one stream synchronisation).
static int accumulate_device_points(mdx_rdf *h, const float *d_pos1, int64_t n1,
                                    const float *d_pos2, int64_t n2, const float *d_boxes,
                                    const float *h_boxes, int64_t n_frames);

// Entry of every accumulate variant: sets with a grouping are reduced to centres of mass first.
// drop_axis: dst = src with coordinate `axis` set to zero (may run in place)
__global__ __launch_bounds__(256) void rdf_drop_axis_kernel(const float *src, float *dst,
                                                            int64_t n_rows, int axis)
{
    const int64_t i = blockIdx.x * int64_t(256) + threadIdx.x;
    if (i >= n_rows * 3)
        return;
    dst[i] = (int)(i % 3) == axis ? 0.f : src[i];
}

// ... and the frame's cell: lengths[axis] = max(lengths) (structure.py:766)
__global__ __launch_bounds__(256) void rdf_drop_box_kernel(const float *__restrict__ src,
                                                           float *__restrict__ dst, int64_t n_frames,
                                                           int axis)
{
    const int64_t f = blockIdx.x * int64_t(256) + threadIdx.x;
    if (f >= n_frames)
        return;
    float b[6];
    for (int k = 0; k < 6; ++k)
        b[k] = src[6 * f + k];
    b[axis] = fmaxf(b[0], fmaxf(b[1], b[2]));
    for (int k = 0; k < 6; ++k)
        dst[6 * f + k] = b[k];
}

static int accumulate_device(mdx_rdf *h, const float *d_pos1, int64_t n1, const float *d_pos2,
                             int64_t n2, const float *d_boxes, const float *h_boxes,
                             int64_t n_frames)
{
    if (!h->grouping[0].n_groups && !h->grouping[1].n_groups && h->drop_axis < 0)
        return accumulate_device_points(h, d_pos1, n1, d_pos2, n2, d_boxes, h_boxes, n_frames);
    if (n_frames == 0)
        return MDX_OK;
    const bool same = (d_pos2 == nullptr || (d_pos2 == d_pos1 && n2 == n1));
    const float *src[2] = {d_pos1, same ? d_pos1 : d_pos2};
    int64_t n[2] = {n1, same ? n1 : n2};
    const float *pts[2] = {src[0], src[1]};
    bool owned[2] = {false, false};
    MDX_REQUIRE(!same || !h->grouping[1].n_groups || h->grouping[0].n_groups,
                "a self histogram takes the grouping of set 1");
    for (int g = 0; g < (same ? 1 : 2); ++g) {
        MoleculeStage &G = h->grouping[g];
        if (!G.active())
            continue;
        MDX_REQUIRE(n[g] == G.n_atoms, "set %d holds %lld particles, its grouping was defined for %lld",
                    g + 1, (long long)n[g], (long long)G.n_atoms);
        MDX_TRY(G.run(h->stream, src[g], n_frames, nullptr, &pts[g]));
        n[g] = G.n_groups;
        owned[g] = true;
    }
    std::vector<float> boxes_host;
    if (h->drop_axis >= 0) {
        // after the centres of mass, as the reference orders it (structure.py:753-766)
        for (int g = 0; g < (same ? 1 : 2); ++g) {
            float *dst = const_cast<float *>(pts[g]);
            if (!owned[g]) {
                MDX_TRY(h->d_drop[g].ensure(size_t(12) * n[g] * n_frames));
                dst = h->d_drop[g].as<float>();
            }
            const int64_t rows = n[g] * n_frames;
            hipLaunchKernelGGL(rdf_drop_axis_kernel, dim3((unsigned)ceil_div(rows * 3, 256)), dim3(256), 0,
                               h->stream, pts[g], dst, rows, h->drop_axis);
            pts[g] = dst;
        }
        if (d_boxes) {
            MDX_TRY(h->d_drop_box.ensure(size_t(24) * n_frames));
            hipLaunchKernelGGL(rdf_drop_box_kernel, dim3((unsigned)ceil_div(n_frames, 256)), dim3(256), 0,
                               h->stream, d_boxes, h->d_drop_box.as<float>(), n_frames, h->drop_axis);
            d_boxes = h->d_drop_box.as<float>();
            if (h_boxes) {
                boxes_host.assign(h_boxes, h_boxes + 6 * n_frames);
                for (int64_t f = 0; f < n_frames; ++f) {
                    float *b = boxes_host.data() + 6 * f;
                    b[h->drop_axis] = std::max(b[0], std::max(b[1], b[2]));
                }
                h_boxes = boxes_host.data();
            }
        }
    }
    MDX_HIP(hipGetLastError());
    if (same)
        return accumulate_device_points(h, pts[0], n[0], nullptr, n[0], d_boxes, h_boxes, n_frames);
    return accumulate_device_points(h, pts[0], n[0], pts[1], n[1], d_boxes, h_boxes, n_frames);
}

static int accumulate_device_points(mdx_rdf *h, const float *d_pos1, int64_t n1,
                                    const float *d_pos2, int64_t n2, const float *d_boxes,
                                    const float *h_boxes, int64_t n_frames)
{
    if (!d_boxes || n_frames == 0)
        return accumulate_ortho(h, d_pos1, n1, d_pos2, n2, d_boxes, n_frames);
    std::vector<float> copy;
    if (!h_boxes) {
        copy.resize(size_t(6) * n_frames);
        MDX_HIP(hipMemcpyAsync(copy.data(), d_boxes, size_t(24) * n_frames, hipMemcpyDeviceToHost,
                               h->stream));
        MDX_HIP(hipStreamSynchronize(h->stream));
        h_boxes = copy.data();
    }
    for (int64_t f0 = 0; f0 < n_frames;) {
        const bool ortho = box_is_ortho(h_boxes + 6 * f0);
        int64_t f1 = f0 + 1;
        while (f1 < n_frames && box_is_ortho(h_boxes + 6 * f1) == ortho)
            ++f1;
        const float *p1 = d_pos1 + f0 * n1 * 3;
        const float *p2 = d_pos2 ? d_pos2 + f0 * n2 * 3 : nullptr;
        if (ortho)
            MDX_TRY(accumulate_ortho(h, p1, n1, p2, n2, d_boxes + f0 * 6, f1 - f0));
        else
            MDX_TRY(accumulate_triclinic(h, p1, n1, p2, n2, h_boxes + 6 * f0, f1 - f0));
        f0 = f1;
    }
    return MDX_OK;
}

static int check_host_boxes(const float *boxes, int64_t n_frames)
{
    for (int64_t f = 0; boxes && f < n_frames; ++f) {
        const float *b = boxes + 6 * f;
        MDX_REQUIRE(b[0] > 0.f && b[1] > 0.f && b[2] > 0.f,
                    "frame %lld: box lengths must be positive", (long long)f);
    }
    return MDX_OK;
}

// Host-buffer and trajectory-file entry points: slabs of frames go through the double-buffered
// staging sets (StagePipeline).  fill(b, f0, nf) queues on the copy stream whatever brings
// frames [f0, f0+nf) into d_stage1[b] (and d_stage2[b] unless `same`).
// prepare(b, f0, nf): queued on the COMPUTE stream ahead of the slab's kernels (the unpack of raw frames).
template <typename Fill, typename Prepare>
static int accumulate_pipelined(mdx_rdf *h, int64_t n1, int64_t n2, bool same, const float *boxes,
                                int64_t n_frames, int64_t source_bytes_per_frame, Fill fill, Prepare prepare)
{
    // slabs of ~256 MiB (MDX_RDF_PIPE_MB), the first two a quarter and a half of that: each slab is one sort + one
    // pair launch on the compute stream; the persistent pair kernel ends on a tail of about one item per block
    // (2 % of a launch of 336 frames at C2, 0.7 % of one of 1 000) and the sort cannot run beside it, so longer
    // launches are worth more than finer overlap; a multiple of 8 frames, so that every XCD gets the same number
    const char *mb_env = getenv("MDX_RDF_PIPE_MB");
    const int64_t pipe_mb = mb_env ? std::min<long long>(std::max<long long>(atoll(mb_env), 1), 4096) : 256;
    int64_t slab = std::max<int64_t>(1, (pipe_mb << 20) / source_bytes_per_frame);
    if (slab >= 8)
        slab -= slab % 8;
    slab = std::min<int64_t>(n_frames, slab);
    return h->pipe.run(
        h->stream, n_frames, slab,
        [&](int b, int64_t f0, int64_t nf) -> int {
            MDX_TRY(h->d_stage1[b].ensure(size_t(12) * n1 * slab));
            if (!same)
                MDX_TRY(h->d_stage2[b].ensure(size_t(12) * n2 * slab));
            MDX_TRY(fill(b, f0, nf, slab));
            if (boxes) {
                MDX_TRY(h->d_boxes[b].ensure(size_t(24) * slab));
                MDX_HIP(hipMemcpyAsync(h->d_boxes[b].ptr, boxes + f0 * 6, size_t(24) * nf,
                                       hipMemcpyHostToDevice, h->pipe.copy_stream));
            }
            return MDX_OK;
        },
        [&](int b, int64_t f0, int64_t nf) -> int {
            MDX_TRY(prepare(b, f0, nf));
            return accumulate_device(h, h->d_stage1[b].as<float>(), n1,
                                     same ? nullptr : h->d_stage2[b].as<float>(), n2,
                                     boxes ? h->d_boxes[b].as<float>() : nullptr,
                                     boxes ? boxes + f0 * 6 : nullptr, nf);
        },
        8);
}

extern "C" {

int mdx_rdf_create(mdx_rdf_t *out, int dev, int n_bins, const double *edges, int64_t excl1,
                   int64_t excl2, int algo)
{
    MDX_REQUIRE(out && edges, "NULL argument");
    MDX_REQUIRE(n_bins >= 1 && n_bins <= (1 << 20), "n_bins=%d out of range", n_bins);
    MDX_REQUIRE((excl1 > 0) == (excl2 > 0) && excl1 >= 0 && excl2 >= 0,
                "exclusion must be two positive integers (or 0, 0 for none)");
    MDX_REQUIRE(algo >= MDX_RDF_ALGO_AUTO && algo <= MDX_RDF_ALGO_CELL, "unknown algo %d", algo);
    for (int b = 0; b < n_bins; ++b)
        MDX_REQUIRE(edges[b] < edges[b + 1], "edges must increase strictly");
    MDX_REQUIRE(edges[0] >= 0.0 && std::isfinite(edges[n_bins]), "range must be finite and >= 0");
    MDX_TRY(set_device(dev));
    mdx_rdf *h = new mdx_rdf();
    h->dev = dev;
    h->n_bins = n_bins;
    h->edges.assign(edges, edges + n_bins + 1);
    h->excl1 = excl1;
    h->excl2 = excl2;
    h->algo = algo;
    // thresholds in the squared-distance domain
    h->thresh.resize(n_bins + 1);
    for (int b = 0; b <= n_bins; ++b)
        h->thresh[b] = thresh_ge(edges[b]);
    // capped_distance: d > r0 - eps (structure.py:94); numpy.histogram: r0 <= d <= r1
    const double min_cut = edges[0] - 2.220446049250313e-16;
    h->t_lo = std::max(h->thresh[0], thresh_gt(min_cut));
    h->t_hi = thresh_gt(edges[n_bins]);
    int rc = MDX_OK;
    do {
        if ((rc = stream_acquire(&h->stream)) != MDX_OK)
            break;
        h->timer.stream = h->stream;
        if ((rc = h->d_thresh.ensure(sizeof(double) * (n_bins + 1))) != MDX_OK) break;
        if ((rc = h->d_counts.ensure(sizeof(uint64_t) * size_t(h->n_rep) * n_bins)) != MDX_OK) break;
        if ((rc = h->d_total.ensure(sizeof(uint64_t) * n_bins)) != MDX_OK) break;
        if ((rc = h->d_misc.ensure(64)) != MDX_OK) break;
        if ((rc = h->d_stats.ensure(RDF_STAT_BYTES)) != MDX_OK) break;
        if ((rc = h->d_work.ensure(CELL_WORK_BYTES)) != MDX_OK) break;
        if (hipMemcpy(h->d_thresh.ptr, h->thresh.data(), sizeof(double) * (n_bins + 1),
                      hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(MDX_ERR_HIP, "threshold upload failed");
            break;
        }
    } while (0);
    if (rc != MDX_OK) {
        mdx_rdf_destroy(h);
        return rc;
    }
    *out = h;
    return mdx_rdf_reset(h);
}

int mdx_rdf_destroy(mdx_rdf_t h)
{
    if (!h)
        return MDX_OK;
    (void)hipSetDevice(h->dev);
    // every stream that may still touch the handle's memory drains first: the blocks go back to the device's
    // cache (DeviceBuffer::recycle), not through hipFree, which would wait for the device itself
    if (h->stream)
        (void)hipStreamSynchronize(h->stream);
    if (h->pipe.copy_stream)
        (void)hipStreamSynchronize(h->pipe.copy_stream);
    h->timer.destroy();
    h->pipe.destroy();
    for (DeviceBuffer *b : {&h->d_thresh, &h->d_counts, &h->d_total, &h->d_pack1, &h->d_pack2,
                            &h->d_stage1[0], &h->d_stage2[0], &h->d_boxes[0], &h->d_stage1[1],
                            &h->d_stage2[1], &h->d_boxes[1], &h->d_rawslab[0], &h->d_rawslab[1], &h->d_index[0], &h->d_index[1], &h->d_tri, &h->d_misc, &h->d_stats, &h->d_work, &h->d_pw1,
                            &h->d_po1, &h->d_bb1, &h->d_pw2, &h->d_po2, &h->d_bb2, &h->d_bb16_1,
                            &h->d_bb16_2, &h->d_drop[0], &h->d_drop[1], &h->d_drop_box})
        b->recycle();
    h->grouping[0].recycle();
    h->grouping[1].recycle();
    if (h->stream)
        stream_release(h->stream);
    delete h;
    return MDX_OK;
}

int mdx_rdf_reset(mdx_rdf_t h)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    MDX_HIP(hipMemsetAsync(h->d_counts.ptr, 0, sizeof(uint64_t) * size_t(h->n_rep) * h->n_bins,
                           h->stream));
    MDX_HIP(hipMemsetAsync(h->d_misc.ptr, 0, 64, h->stream));
    MDX_HIP(hipMemsetAsync(h->d_stats.ptr, 0, RDF_STAT_BYTES, h->stream));
    MDX_HIP(hipStreamSynchronize(h->stream));
    h->timer.reset();
    h->pairs_evaluated = 0;
    h->pairs_bruteforce = 0;
    h->reduced_global = false;
    return MDX_OK;
}

int mdx_rdf_set_drop_axis(mdx_rdf_t h, int axis)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_REQUIRE(axis >= -1 && axis <= 2, "axis must be 0, 1, 2 or -1 (none)");
    h->drop_axis = axis;
    return MDX_OK;
}

int mdx_rdf_set_grouping(mdx_rdf_t h, int which, int64_t n_groups, const int64_t *offsets,
                         const double *masses)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_REQUIRE(which == 1 || which == 2, "which must be 1 or 2");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    return h->grouping[which - 1].set(n_groups, offsets, masses);
}

int mdx_rdf_accumulate_device(mdx_rdf_t h, const float *d_pos1, int64_t n1, const float *d_pos2,
                              int64_t n2, const float *d_boxes, int64_t n_frames)
{
    MDX_REQUIRE(h && d_pos1, "NULL argument");
    MDX_REQUIRE(n1 >= 0 && n2 >= 0 && n_frames >= 0, "negative size");
    MDX_TRY(set_device(h->dev));
    return accumulate_device(h, d_pos1, n1, d_pos2, n2, d_boxes, nullptr, n_frames);
}

int mdx_rdf_accumulate(mdx_rdf_t h, const float *pos1, int64_t n1, const float *pos2, int64_t n2,
                       const float *boxes, int64_t n_frames)
{
    MDX_REQUIRE(h && pos1, "NULL argument");
    MDX_REQUIRE(n1 >= 0 && n2 >= 0 && n_frames >= 0, "negative size");
    MDX_TRY(set_device(h->dev));
    if (n_frames == 0 || n1 == 0)
        return MDX_OK;
    const bool same = (pos2 == nullptr || (pos2 == pos1 && n2 == n1));
    if (pos2 == nullptr)
        n2 = n1;
    if (n2 == 0)
        return MDX_OK;
    MDX_TRY(check_host_boxes(boxes, n_frames));
    return accumulate_pipelined(
        h, n1, n2, same, boxes, n_frames, 12 * std::max(n1, n2),
        [&](int b, int64_t f0, int64_t nf, int64_t) -> int {
            // caller memory -> pinned ring (host threads) -> HBM; pinned / registered caller
            // memory is read by the DMA engine where it lies
            MDX_TRY(device_stager(h->dev).upload(h->dev, h->pipe.copy_stream, h->d_stage1[b].ptr,
                                          pos1 + f0 * n1 * 3, size_t(12) * n1 * nf));
            if (!same)
                MDX_TRY(device_stager(h->dev).upload(h->dev, h->pipe.copy_stream, h->d_stage2[b].ptr,
                                              pos2 + f0 * n2 * 3, size_t(12) * n2 * nf));
            return MDX_OK;
        },
        [](int, int64_t, int64_t) -> int { return MDX_OK; });
}

// Frames straight from a trajectory file (mdx_traj.hip): raw records -> pinned -> HBM ->
// unpack/gather into the staging slabs, overlapped with the pair kernels of the previous slab.
int mdx_rdf_accumulate_traj(mdx_rdf_t h, mdx_traj_t traj, const int64_t *frames, int64_t n_frames,
                            const float *boxes, const int32_t *index1, int64_t n1,
                            const int32_t *index2, int64_t n2)
{
    MDX_REQUIRE(h && traj, "NULL handle");
    MDX_REQUIRE(n1 >= 0 && n2 >= 0 && n_frames >= 0, "negative size");
    MDX_REQUIRE(n_frames == 0 || frames, "NULL frame list");
    MDX_TRY(set_device(h->dev));
    Trajectory *t = mdx_traj_internal(traj);
    const bool same = (index2 == nullptr && n2 == 0);
    if (index1 == nullptr)
        n1 = t->n_atoms;
    if (same)
        n2 = n1;
    else if (index2 == nullptr)
        n2 = t->n_atoms;
    if (n_frames == 0 || n1 == 0 || n2 == 0)
        return MDX_OK;
    MDX_TRY(check_host_boxes(boxes, n_frames));
    // particle selections live on the device for the unpack kernel
    const int32_t *host_idx[2] = {index1, same ? nullptr : index2};
    const int64_t n_idx[2] = {n1, n2};
    const int *d_idx[2] = {nullptr, nullptr};
    for (int g = 0; g < 2; ++g) {
        if (!host_idx[g])
            continue;
        for (int64_t i = 0; i < n_idx[g]; ++i)
            if (host_idx[g][i] < 0 || host_idx[g][i] >= t->n_atoms)
                return fail(MDX_ERR_INVALID_VALUE, "particle index %d out of range [0, %lld)",
                            host_idx[g][i], (long long)t->n_atoms);
        MDX_TRY(h->pipe.ensure());
        // an earlier call's unpack kernels may still be reading the previous selection
        MDX_HIP(hipStreamSynchronize(h->pipe.copy_stream));
        MDX_TRY(h->d_index[g].ensure(size_t(4) * n_idx[g]));
        MDX_HIP(hipMemcpy(h->d_index[g].ptr, host_idx[g], size_t(4) * n_idx[g],
                          hipMemcpyHostToDevice));
        d_idx[g] = h->d_index[g].as<int>();
    }
    // raw frames -> pinned ring -> HBM by DMA on the ring's stream, beside the kernels of the previous slab;
    // the unpack kernel (byte swap / plane transpose / gather) is queued on the compute stream ahead of the
    // slab's sort: the persistent pair kernel leaves no wave slot for a kernel on another stream
    return accumulate_pipelined(
        h, n1, n2, same, boxes, n_frames, 12 * t->n_atoms,
        [&](int b, int64_t f0, int64_t nf, int64_t slab) -> int {
            MDX_TRY(h->d_rawslab[b].ensure(size_t(12) * t->n_atoms * slab));
            return t->stage_raw_async(h->dev, h->pipe.copy_stream, frames + f0, nf, h->d_rawslab[b].ptr);
        },
        [&](int b, int64_t, int64_t nf) -> int {
            TrajSelection sel[2] = {{d_idx[0], n1, h->d_stage1[b].as<float>()},
                                    {d_idx[1], n2, same ? nullptr : h->d_stage2[b].as<float>()}};
            return t->unpack_async(h->stream, h->d_rawslab[b].ptr, nf, sel, same ? 1 : 2);
        });
}

// Sum of the statistics shards after the stream has drained: [0] exact pairs, [1] units, [2] general units,
// [3] engine-clock ticks, [4] 100 MHz ticks.
static int read_stats(mdx_rdf *h, unsigned long long out[5])
{
    MDX_HIP(hipStreamSynchronize(h->stream));
    std::vector<unsigned long long> raw(size_t(RDF_STAT_SHARDS) * RDF_STAT_STRIDE);
    MDX_HIP(hipMemcpy(raw.data(), h->d_stats.ptr, RDF_STAT_BYTES, hipMemcpyDeviceToHost));
    for (int k = 0; k < 5; ++k)
        out[k] = 0;
    for (unsigned sh = 0; sh < RDF_STAT_SHARDS; ++sh)
        for (int k = 0; k < 5; ++k)
            out[k] += raw[size_t(sh) * RDF_STAT_STRIDE + k];
    return MDX_OK;
}

int mdx_rdf_synchronize(mdx_rdf_t h)
{
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    h->timer.collect();
    unsigned misc[4] = {0, 0, 0, 0};
    MDX_HIP(hipMemcpy(misc, h->d_misc.ptr, sizeof(misc), hipMemcpyDeviceToHost));
    if (misc[1] & 1u)
        return fail(MDX_ERR_UNSUPPORTED,
                    "a frame has a non-orthorhombic or non-positive box; counts are invalid");
    return MDX_OK;
}

static int reduce_to_total(mdx_rdf *h)
{
    hipLaunchKernelGGL(rdf_reduce_kernel, dim3((unsigned)ceil_div(h->n_bins, 256)), dim3(256), 0,
                       h->stream, h->d_counts.as<unsigned long long>(), h->n_rep, h->n_bins,
                       h->d_total.as<unsigned long long>());
    MDX_HIP(hipGetLastError());
    return MDX_OK;
}

int mdx_rdf_counts(mdx_rdf_t h, int64_t *counts)
{
    MDX_REQUIRE(h && counts, "NULL argument");
    MDX_TRY(set_device(h->dev));
    MDX_TRY(reduce_to_total(h));
    MDX_TRY(mdx_rdf_synchronize(h));
    MDX_HIP(hipMemcpy(counts, h->d_total.ptr, sizeof(int64_t) * h->n_bins, hipMemcpyDeviceToHost));
    return MDX_OK;
}

int mdx_rdf_stats(mdx_rdf_t h, int64_t *launches, double *kernel_ms, int64_t *pairs_evaluated,
                  int64_t *pairs_exact, int64_t *pairs_computed)
{
    if (pairs_computed && h) {
        // distance evaluations actually executed: brute-force tiles + culled cell tiles
        unsigned long long st[5];
        if (set_device(h->dev) == MDX_OK && read_stats(h, st) == MDX_OK)
            *pairs_computed = h->pairs_bruteforce + (int64_t)st[1] * 64 * CELL_CHUNK;
    }
    MDX_REQUIRE(h, "NULL handle");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    h->timer.collect();
    if (launches) *launches = h->timer.launches;
    if (kernel_ms) *kernel_ms = h->timer.total_ms;
    if (pairs_evaluated) *pairs_evaluated = h->pairs_evaluated;
    if (pairs_exact) {
        unsigned long long st[5];
        MDX_TRY(read_stats(h, st));
        *pairs_exact = (int64_t)st[0];
    }
    return MDX_OK;
}

int mdx_rdf_kernel_clock(mdx_rdf_t h, double *hz)
{
    MDX_REQUIRE(h && hz, "NULL argument");
    MDX_TRY(set_device(h->dev));
    unsigned long long st[5];
    MDX_TRY(read_stats(h, st));
    // s_memrealtime counts at 100 MHz
    *hz = st[4] ? double(st[3]) / double(st[4]) * 1.0e8 : 0.0;
    return MDX_OK;
}

int mdx_rdf_debug_counters(mdx_rdf_t h, int64_t out[4])
{
    MDX_REQUIRE(h && out, "NULL argument");
    MDX_TRY(set_device(h->dev));
    unsigned long long raw[5];
    MDX_TRY(read_stats(h, raw));
    out[0] = (int64_t)raw[0];   // pairs re-evaluated exactly
    out[1] = (int64_t)raw[1];   // (64 i) x (16 j) units evaluated by the cell kernel
    out[2] = (int64_t)raw[2];   // ... of which on the per-pair image-search path
    out[3] = h->pairs_bruteforce;
    return MDX_OK;
}

int mdx_rdf_debug_sorted(mdx_rdf_t h, int64_t frame, int64_t n_pad, float *pw, float *po)
{
    MDX_REQUIRE(h && pw && po, "NULL argument");
    MDX_TRY(set_device(h->dev));
    MDX_HIP(hipStreamSynchronize(h->stream));
    MDX_REQUIRE(h->last_frames > 0, "no slab has gone through the cell-sorted path yet");
    MDX_REQUIRE(frame >= 0 && frame < h->last_frames, "frame %lld outside the last slab (%lld frames)",
                (long long)frame, (long long)h->last_frames);
    MDX_REQUIRE(n_pad == h->last_n_pad, "n_pad is %lld for the last slab", (long long)h->last_n_pad);
    // the slab's own set of sorted copies (two sets alternate when the sort runs beside the pair kernel)
    MDX_HIP(hipMemcpy(pw, h->d_pw1.as<float4>() + h->last_offset + frame * n_pad, size_t(16) * n_pad,
                      hipMemcpyDeviceToHost));
    if (h->last_lazy)
        // exclusion 0 or 1: no sorted copy of the original coordinates exists (the exact path reads the
        // incoming frame by the tag in pw[..].w); say so instead of returning something else in its place
        return fail(MDX_ERR_STATE, "the last slab has no sorted originals (exclusion 0 or 1): pw is filled, "
                    "po is not; MDX_RDF_SORTED_ORIGINALS=1 materialises them");
    MDX_HIP(hipMemcpy(po, h->d_po1.as<float4>() + h->last_offset + frame * n_pad, size_t(16) * n_pad,
                      hipMemcpyDeviceToHost));
    return MDX_OK;
}

int mdx_rdf_enable_timing(mdx_rdf_t h, int on)
{
    MDX_REQUIRE(h, "NULL handle");
    h->timer.enabled = on != 0;
    return MDX_OK;
}

// used by mdx_comm.hip
int mdx_rdf_internal_total(mdx_rdf_t h, unsigned long long **d_total, hipStream_t *stream)
{
    MDX_TRY(set_device(h->dev));
    MDX_TRY(reduce_to_total(h));
    *d_total = h->d_total.as<unsigned long long>();
    *stream = h->stream;
    return MDX_OK;
}

int mdx_rdf_internal_nbins(mdx_rdf_t h) { return h->n_bins; }

int mdx_rdf_internal_adopt_total(mdx_rdf_t h)
{
    // replica 0 <- all-reduced total, other replicas <- 0
    MDX_HIP(hipMemsetAsync(h->d_counts.ptr, 0, sizeof(uint64_t) * size_t(h->n_rep) * h->n_bins,
                           h->stream));
    MDX_HIP(hipMemcpyAsync(h->d_counts.ptr, h->d_total.ptr, sizeof(uint64_t) * h->n_bins,
                           hipMemcpyDeviceToDevice, h->stream));
    h->reduced_global = true;
    return MDX_OK;
}

int mdx_radial_histogram(int dev, const float *pos1, int64_t n1, const float *pos2, int64_t n2,
                         int n_bins, const double *edges, const float dims[6], int64_t excl1,
                         int64_t excl2, int64_t *counts)
{
    MDX_REQUIRE(counts, "counts is NULL");
    mdx_rdf_t h = nullptr;
    MDX_TRY(mdx_rdf_create(&h, dev, n_bins, edges, excl1, excl2, MDX_RDF_ALGO_AUTO));
    int rc = mdx_rdf_accumulate(h, pos1, n1, pos2, n2, dims, 1);
    if (rc == MDX_OK)
        rc = mdx_rdf_counts(h, counts);
    mdx_rdf_destroy(h);
    return rc;
}

}  // extern "C"
