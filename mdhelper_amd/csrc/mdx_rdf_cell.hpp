// mdx_rdf_cell.hpp — cell-sorted, tile-culled pair histogram (RDF algo "cell").
//
// Same result contract as mdx_rdf.hip (bin decisions bit-exact with the contract
// arithmetic); what changes is which pairs are looked at and how cheaply.
//
//   rdf_cell_sort_kernel   one 1024-thread block per frame: wraps the particles into
//        the box, counting-sorts them by cell (cells ~8 particles, ordered in 2x2x2
//        bricks so that 64 consecutive particles form a compact blob), writes
//        the sorted wrapped coordinates Pw (float4, w = exclusion tag), the sorted
//        ORIGINAL coordinates Po (the contract arithmetic needs those) and one
//        bounding box per 64-particle tile.  Everything in LDS, no global atomics.
//   rdf_cell_pair_kernel   one 256-thread block per (frame, 128-particle i tile);
//        each WAVE walks its share of the 64-particle j tiles on its own (no block
//        barrier in the loop): 64 candidate j tiles are tested per instruction
//        against the i tile's bounding box (minimum-image gap > r_max + error bound
//        -> culled), survivors are staged into a wave-private LDS slab — already
//        shifted by the tile pair's periodic image, so the inner loop has NO
//        per-pair image search — and read back by broadcast, two i particles per
//        lane.  Inner loop per pair: 3 sub, 1 mul, 2 fma, 1 compare (float32).
//        Candidates take the same float32-filter / exact-fallback path as the
//        FILTER kernel; the rare exact evaluation re-reads the original coordinates.
//
// Why the culled pairs can be skipped: the bounding boxes are those of the very
// float32 wrapped coordinates the filter distance is computed from, the filter
// distance is within `margin_d` of the contract distance, and a
// tile pair is dropped only if its box gap exceeds r_max + margin_d + slack.
#pragma once

#include "mdx_rdf_device.hpp"

struct CellArgs {
    // per set (1 = i side, 2 = j side; identical pointers for a self histogram)
    const float4 *pw1, *po1, *bb1;
    const float4 *pw2, *po2, *bb2;
    const float4 *bb16_2;        // boxes of the CELL_CHUNK-particle chunks of the j side
    // po1 / po2 == nullptr: the sorted originals are not materialised; a particle's tag IS its index in the
    // set (exclusion 0 or 1) and the exact path reads in1 / in2 = the frames as they came in, [frame][n][3]
    const float *in1, *in2;
    int n1_in, n2_in;
    const float *boxes;          // [frames][6] (orthorhombic frames)
    const float *tri;            // [frames][9] cell matrices (triclinic kernel variant)
    const double *thresh;        // [n_bins+1]
    unsigned long long *counts;  // [n_rep][n_bins]
    const unsigned *maxabs_bits;
    // words of the statistics shards (rdf_stat_offset; mdx_rdf_device.hpp)
    unsigned long long *exact_counter;      // word 0
    unsigned long long *tilepair_counter;   // words 1, 2: (64 i) x (CELL_CHUNK j) units evaluated, of which general
    unsigned long long *clock_counter;      // words 3, 4 (timing runs): engine-clock ticks, 100 MHz ticks
    double t_lo, t_hi, r0, r1;
    int n1p, n2p;                // padded particle counts (multiples of 128)
    int n_bins, n_hist, n_rep;
    int self, frame0;
    int n_frames;                // frames of this launch
    int tags_everywhere;         // 0: exclusion tags can only collide inside the diagonal tiles
    // persistent blocks: eight work counters (one per XCD, 32 words apart, zeroed before the launch) hand out
    // the (frame, i tile) items of the XCD's frames
    unsigned *work;
    // 32-bit LDS bins are flushed to the 64-bit replicas before this many j tiles of 2^14 possible adds each
    // could have gone into one bin since the last flush (2^18; MDX_RDF_LDS_FLUSH_UNITS: test hook)
    unsigned flush_units;
    // launches of fewer than eight frames (the function-level drop-in hands over ONE): the items are dealt to the
    // XCDs one by one instead of frame by frame — with a frame per XCD seven eighths of the chip would idle, and
    // the 0.5 MB of a frame's sorted copy read by eight L2s instead of one is nothing
    int spread;
};

constexpr unsigned CELL_WORK_STRIDE = 32;   // words between the work counters of two XCDs (one 128-byte line each)
constexpr size_t CELL_WORK_BYTES = 8 * CELL_WORK_STRIDE * 4;

constexpr int CELL_MAX = 16384;   // cells per frame (64 KiB of LDS counters)
constexpr int SORT_THREADS = 1024;
constexpr int CELL_QCAP = 1024;   // surviving j tiles queued per round of the pair kernel
constexpr int CELL_CHUNK = 1;     // particles per j chunk of the second-level cull (1, 2, 4, 8, 16)
constexpr int CELL_NCHUNK = 64 / CELL_CHUNK;
constexpr int CELL_TODO = 128;    // per-wave list of pairs waiting for the exact arithmetic
// Per-wave LDS histogram: n_bins bins and one slot that absorbs a (proven impossible)
// index n_bins instead of letting it alias the next histogram's bin 0.
__host__ __device__ constexpr int cell_hist_stride(int n_bins) { return n_bins + 1; }

// LDS histogram of the pair kernel: n_bins + 1 bins x R interleaved replicas, word (bin * R + lane % R).
// One wave instruction is served in two groups of 32 lanes on 32 banks; with the replicas interleaved two
// lanes of a group meet on a bank only when lane % R agrees AND the bins agree modulo 32 / R — at R = 8 the
// ~8-17 adding lanes of a step hardly ever do, where per-wave histograms (bank = bin % 32) made them queue
// 2-3 deep.  (LDS executes one wave instruction at a time: histograms private to a wave bought nothing.)
struct HistLdsRep {
    unsigned *h;   // replica of this lane: base + lane % R
    int sh;        // log2 R
    __device__ inline void add(int k, unsigned w) const { atomicAdd(h + (k << sh), w); }
};
__device__ inline int cell_hist_shift(const HistLdsRep &h) { return h.sh; }
template <typename Hist> __device__ inline int cell_hist_shift(const Hist &) { return 0; }

struct CellGrid {
    int nc[3];
    float Lf[3];
    double Ld[3], invLd[3];
    __device__ inline void init(const float *box, int n)
    {
        double vol = 1.0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            Lf[k] = box[k];
            Ld[k] = (double)box[k];
            invLd[k] = 1.0 / Ld[k];
            vol *= Ld[k];
        }
        // Columns of cross-section a x a along z, a^3 = the volume of 64 particles, cut into layers of ~4
        // particles (a / 16 thick; with ~8 per layer 1 % slower, with ~2 the sort costs what the boxes gain):
        // sorted column by column and layer by layer, 64 consecutive particles are a slab of a column
        // about `a` thick — a box of a x a x a wherever the 64 begin.  (With cubic cells in 2x2x2 bricks a group of 64
        // that straddled two bricks was two bricks long: mean extents 8.4 x 9.1 x 15.0 A at C2, now 8.5 x 9.2 x 8.8;
        // the volume within the cutoff of such a box, which is what the pair kernel evaluates, is 16 % smaller.)
        const double a = cbrt(64.0 * vol / (double)max(n, 1));
        nc[0] = min(max((int)rint(Ld[0] / a), 1), 64);
        nc[1] = min(max((int)rint(Ld[1] / a), 1), 64);
        nc[2] = max((int)rint(16.0 * Ld[2] / a), 1);
        while (nc[0] * nc[1] * nc[2] > CELL_MAX)   // thicker layers until the table fits
            nc[2] = (nc[2] + 1) / 2;
    }
    __device__ inline int n_cells() const { return nc[0] * nc[1] * nc[2]; }
    // wrapped coordinate in [0, L] (one float32 rounding of the exact wrap) and its cell
    __device__ inline float wrap(float x, int k, int &c) const
    {
        double xd = (double)x;
        double wd = xd - Ld[k] * floor(xd * invLd[k]);
        float w = (float)wd;
        c = min(max((int)(wd * invLd[k] * nc[k]), 0), nc[k] - 1);
        return w;
    }
    __device__ inline int key(int cx, int cy, int cz) const
    {
        // columns in a serpentine (boustrophedon) path through x, y; the layers of a column run up or down in
        // turn, so that consecutive cells are always neighbours — no jump from the end of one column to the
        // start of the next, which would give a tile a box-long extent
        if (cx & 1)
            cy = nc[1] - 1 - cy;
        const int col = cx * nc[1] + cy;
        if (col & 1)
            cz = nc[2] - 1 - cz;
        return col * nc[2] + cz;
    }
};

// Triclinic frames (TRI): `boxes` holds the 9-float cell matrices B (rows a, b, c).  A particle
// is moved into the central cell exactly as the contract prescribes (c, b, a in turn, double
// arithmetic, float32 result: DESIGN.md §4.5) — that wrapped position is BOTH sorted copies —
// and it is binned by its fractional coordinates, on a grid sized from the cell heights.
struct TriCell {
    double B[9];
    double h[3];   // perpendicular heights
    __device__ inline void init(const float *b)
    {
#pragma unroll
        for (int i = 0; i < 9; ++i)
            B[i] = (double)b[i];
        const double vol = B[0] * B[4] * B[8];
        const double bcx = B[4] * B[8], bcy = -B[3] * B[8], bcz = B[3] * B[7] - B[4] * B[6];
        h[0] = vol / sqrt(bcx * bcx + bcy * bcy + bcz * bcz);
        h[1] = B[4] * B[8] / sqrt(B[8] * B[8] + B[7] * B[7]);
        h[2] = B[8];
    }
    // contract wrap (the arithmetic of rdf_tri_pack_kernel)
    __device__ inline void wrap(float x, float y, float z, float &wx, float &wy, float &wz) const
    {
        double r[3] = {(double)x, (double)y, (double)z};
#pragma unroll
        for (int k = 2; k >= 0; --k) {
            const double s = floor(r[k] / B[4 * k]);
#pragma unroll
            for (int c = 0; c <= k; ++c)
                r[c] -= s * B[3 * k + c];
        }
        wx = (float)r[0];
        wy = (float)r[1];
        wz = (float)r[2];
    }
    // cell of a wrapped position on an nc[0] x nc[1] x nc[2] grid in fractional coordinates
    __device__ inline void cell(float wx, float wy, float wz, const int *nc, int &cx, int &cy,
                                int &cz) const
    {
        const double sz = (double)wz / B[8];
        const double sy = ((double)wy - sz * B[7]) / B[4];
        const double sx = ((double)wx - sy * B[3] - sz * B[6]) / B[0];
        cx = min(max((int)floor(sx * nc[0]), 0), nc[0] - 1);
        cy = min(max((int)floor(sy * nc[1]), 0), nc[1] - 1);
        cz = min(max((int)floor(sz * nc[2]), 0), nc[2] - 1);
    }
};

template <bool TRI>
__global__ __launch_bounds__(SORT_THREADS) void rdf_cell_sort_kernel(
    const float *__restrict__ pos, const float *__restrict__ boxes, int n, int n_pad, int64_t excl,
    float4 *__restrict__ pw, float4 *__restrict__ po, float4 *__restrict__ bb,
    float4 *__restrict__ bb16, unsigned *maxabs_bits)
{
    __shared__ unsigned cnt[CELL_MAX];
    __shared__ unsigned part[SORT_THREADS];
    const int tid = threadIdx.x;
    const int frame = blockIdx.x;
    const float *P = pos + int64_t(frame) * n * 3;
    float4 *PW = pw + int64_t(frame) * n_pad;
    float4 *PO = po + int64_t(frame) * n_pad;
    CellGrid g;
    TriCell tc;
    if (TRI) {
        tc.init(boxes + int64_t(frame) * 9);
        const float hb[3] = {(float)tc.h[0], (float)tc.h[1], (float)tc.h[2]};
        g.init(hb, n);   // grid dimensions from the heights
    } else {
        g.init(boxes + int64_t(frame) * 6, n);
    }
    const int ncell = g.n_cells();

    for (int c = tid; c < ncell; c += SORT_THREADS)
        cnt[c] = 0u;
    __syncthreads();

    float m = 0.0f;
    // (unrolled: the loads of four particles are in flight before the first is counted; a block is one
    // frame and nothing else hides their latency)
#pragma unroll 4
    for (int a = tid; a < n; a += SORT_THREADS) {
        float x = P[3 * a], y = P[3 * a + 1], z = P[3 * a + 2];
        int cx, cy, cz;
        if (TRI) {
            tc.wrap(x, y, z, x, y, z);   // the bound below is on the WRAPPED coordinates
            tc.cell(x, y, z, g.nc, cx, cy, cz);
        } else {
            g.wrap(x, 0, cx);
            g.wrap(y, 1, cy);
            g.wrap(z, 2, cz);
        }
        atomicAdd(&cnt[g.key(cx, cy, cz)], 1u);
        float am = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
        m = fmaxf(m, am == am ? am : __int_as_float(0x7f800000));
    }
    {
        // one atomicMax per block, not per wave: device atomics on one address complete one per ~24 ns, and
        // sixteen per frame were a third of this kernel
        unsigned bits = __float_as_uint(m);
        for (int off = 32; off > 0; off >>= 1)
            bits = max(bits, (unsigned)__shfl_xor((int)bits, off));
        if ((tid & 63) == 0)
            part[tid >> 6] = bits;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned bits = 0u;
        for (int w = 0; w < SORT_THREADS / 64; ++w)
            bits = max(bits, part[w]);
        if (bits)
            atomicMax(maxabs_bits, bits);
    }
    __syncthreads();   // part[] is reused by the scan

    // exclusive scan of cnt[0..ncell): each thread owns a contiguous run
    const int per = (ncell + SORT_THREADS - 1) / SORT_THREADS;
    const int c0 = tid * per, c1 = min(c0 + per, ncell);
    unsigned local = 0;
    for (int c = c0; c < c1; ++c)
        local += cnt[c];
    part[tid] = local;
    __syncthreads();
    for (int off = 1; off < SORT_THREADS; off <<= 1) {
        unsigned v = (tid >= off) ? part[tid - off] : 0u;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    unsigned run = part[tid] - local;
    for (int c = c0; c < c1; ++c) {
        unsigned v = cnt[c];
        cnt[c] = run;
        run += v;
    }
    __syncthreads();

    // scatter: the cursor of a cell hands out its slots
#pragma unroll 4
    for (int a = tid; a < n; a += SORT_THREADS) {
        float x = P[3 * a], y = P[3 * a + 1], z = P[3 * a + 2];
        int cx, cy, cz;
        float wx, wy, wz;
        if (TRI) {
            tc.wrap(x, y, z, wx, wy, wz);
            tc.cell(wx, wy, wz, g.nc, cx, cy, cz);
            x = wx, y = wy, z = wz;   // the contract evaluates the wrapped positions
        } else {
            wx = g.wrap(x, 0, cx), wy = g.wrap(y, 1, cy), wz = g.wrap(z, 2, cz);
        }
        unsigned slot = atomicAdd(&cnt[g.key(cx, cy, cz)], 1u);
        float tag = __int_as_float(excl > 0 ? int(int64_t(a) / excl) : a);
        PW[slot] = make_float4(wx, wy, wz, tag);
        if (po)
            PO[slot] = make_float4(x, y, z, tag);
    }
    const float qnan = __int_as_float(0x7fc00000);
    for (int a = n + tid; a < n_pad; a += SORT_THREADS) {
        PW[a] = make_float4(qnan, qnan, qnan, __int_as_float(-1));
        if (po)
            PO[a] = make_float4(qnan, qnan, qnan, __int_as_float(-1));
    }
    __threadfence_block();
    __syncthreads();

    // bounding boxes (of the float32 wrapped coordinates): one per 16-particle chunk,
    // one per 64-particle tile
    const int lane = tid & 63, wave = tid >> 6;
    const int n_tiles = n_pad / 64;
    float4 *BB = bb + int64_t(frame) * n_tiles * 2;
    float4 *BB16 = bb16 + int64_t(frame) * n_tiles * 2 * CELL_NCHUNK;
#pragma unroll 2
    for (int t = wave; t < n_tiles; t += SORT_THREADS / 64) {
        float4 v = PW[t * 64 + lane];
        const float inf = __int_as_float(0x7f800000);
        bool ok = v.x == v.x;
        float lo[3] = {ok ? v.x : inf, ok ? v.y : inf, ok ? v.z : inf};
        float hi[3] = {ok ? v.x : -inf, ok ? v.y : -inf, ok ? v.z : -inf};
        for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
                hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
            }
            if (CELL_CHUNK >= 4 && off == CELL_CHUNK / 2 && (lane & (CELL_CHUNK - 1)) == 0) {
                const int c = t * CELL_NCHUNK + lane / CELL_CHUNK;
                BB16[c * 2] = make_float4(lo[0], lo[1], lo[2], 0.f);
                BB16[c * 2 + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
            }
        }
        if (lane == 0) {
            BB[2 * t] = make_float4(lo[0], lo[1], lo[2], 0.f);
            BB[2 * t + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
        }
    }
}

// Exact re-evaluation of one pair with the contract arithmetic on the ORIGINAL coordinates.
// Triclinic contract: the strictly smallest of the 27 images, in double.
__device__ inline double rdf_rsq_contract_tri(const float *Bf, const float4 &pi, const float4 &pj)
{
    const double b00 = Bf[0], b10 = Bf[3], b11 = Bf[4], b20 = Bf[6], b21 = Bf[7], b22 = Bf[8];
    const double dx = (double)(pj.x - pi.x), dy = (double)(pj.y - pi.y), dz = (double)(pj.z - pi.z);
    double best = 1.0e300;
    // rolled loops: this is cold code and must not raise the register count of the hot loop
#pragma unroll 1
    for (int ix = -1; ix < 2; ++ix) {
        const double rx = dx + b00 * (double)ix;
#pragma unroll 1
        for (int iy = -1; iy < 2; ++iy) {
            const double ry0 = rx + b10 * (double)iy;
            const double ry1 = dy + b11 * (double)iy;
#pragma unroll 1
            for (int iz = -1; iz < 2; ++iz) {
                const double rz0 = ry0 + b20 * (double)iz;
                const double rz1 = ry1 + b21 * (double)iz;
                const double rz2 = dz + b22 * (double)iz;
                const double dsq = (rz0 * rz0 + rz1 * rz1) + rz2 * rz2;
                best = dsq < best ? dsq : best;
            }
        }
    }
    return best;
}

template <typename Hist>
__device__ inline void cell_pair_exact(const PairCtx<true> &c, const CellArgs &a,
                                             const double *sT, const Hist &hist, const float4 &po_i,
                                             const float4 &po_j, unsigned w,
                                             const float *tri = nullptr)
{
    double rsq = tri ? rdf_rsq_contract_tri(tri, po_i, po_j)
                     : rdf_rsq_contract<true>(c, po_i, po_j);
    if ((rsq >= a.t_lo) && (rsq < a.t_hi))
        hist.add(rdf_bin_exact(rsq, sT, a.n_bins, c.r0f, c.inv_wf), w);
}

// Original (unwrapped) coordinates of sorted particle `idx` of a frame, for the exact path: the sorted copy where
// it exists, else the incoming row named by the particle's tag (CellArgs::in1).  Half the scattered stores of the
// sort kernel — two thirds of its time — were this copy, read by ~10^-3 of the evaluations.
struct CellOrig {
    const float4 *po;   // sorted originals of the frame, or nullptr
    const float4 *pw;   // sorted wrapped copies of the frame (.w = tag = row index when po is nullptr)
    const float *in;    // the frame as it came in, [n][3]
    __device__ inline float4 at(unsigned idx) const
    {
        if (po)
            return po[idx];
        const float tag = pw[idx].w;
        const float *r = in + 3 * (int64_t)__float_as_int(tag);
        return make_float4(r[0], r[1], r[2], tag);
    }
};

// wave-uniform float constants of the hot loop, forced into SGPRs
struct CellHot {
    float cand_hi, cand_lo, inv_w, sure_w;   // sure_w = 1 - 2 eta
    float pos0;   // -r0/width - eta; kept in a VECTOR register (an fma takes one scalar operand)
    float L[3], invL[3];   // box lengths and fl32(1/L): widened to fp64 only by the exact passes
    const float *tri;      // triclinic variant: the frame's cell matrix (LDS), else nullptr
};

__device__ inline float cell_uniform(float v)
{
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// A wave-uniform value the optimiser cannot see through: what is derived from the copy is computed where it is
// used instead of being hoisted out of the loops around it (and kept in scalar registers across them).
// The launch's arguments through a pointer the optimiser cannot see through: a by-value kernel argument read
// inside the item loop is a loop-invariant scalar load, which is hoisted to the top of the kernel and then
// occupies scalar registers across every loop below it; read through this pointer it is loaded (s_load from the
// kernarg segment) where the item's prologue needs it.
__device__ inline const __constant__ CellArgs *cell_args()
{
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (const __constant__ CellArgs *)p;
}

__device__ inline unsigned cell_opaque(unsigned v)
{
    v = __builtin_amdgcn_readfirstlane(v);   // (a no-op for a value already in a scalar register)
    asm volatile("" : "+s"(v));
    return v;
}

// wave-uniform bookkeeping of the undecided pairs (scalar registers)
struct CellWave {
    uint2 *todo;         // this wave's list in LDS: (i index, j index | (weight - 1) << 31)
    unsigned n_todo;     // entries waiting
    unsigned overflow;   // an append did not fit since the mark was taken
    unsigned n_exact;    // pairs sent to the exact arithmetic so far
};

// The float32 filter of one pair: bin coordinate, candidate and sure flags.
template <bool LOWER, int TAGS>
__device__ inline void cell_filter(const CellHot &c, float fx, float fy, float fz, int tag_i,
                                   int tag_j, float &pos, bool &cand, bool &sure)
{
    float r2 = __fmaf_rn(fz, fz, __fmaf_rn(fy, fy, fx * fx));
    // pos = (sqrt(r2) - r0) / width - eta; raw v_sqrt_f32 (a denormal r2 ends on the exact path).
    // "Farther than eta from both neighbouring bin edges" is fract(pos) < 1 - 2 eta in this
    // shifted coordinate: with pos = k + t, the unshifted fraction is t + eta, which lies in
    // (eta, 1 - eta) exactly when t < 1 - 2 eta (t + eta >= 1 means it wrapped to < eta).  One
    // v_fract and one compare; for a sure pair floor(pos) is the bin, and since a candidate has
    // pos > -2 eta (window) the truncating v_cvt_i32_f32 is that floor.
    pos = __fmaf_rn(__builtin_amdgcn_sqrtf(r2), c.inv_w, c.pos0);
    const float t = __builtin_amdgcn_fractf(pos);
    cand = LOWER ? (r2 < c.cand_hi && r2 >= c.cand_lo) : (r2 < c.cand_hi);
    sure = t < c.sure_w;
    if (TAGS)
        cand = cand && (tag_i != tag_j);
}

__device__ inline void cell_exact_ctx(const CellHot &c, const CellArgs &a, PairCtx<true> &cx)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {   // L and fl32(1/L), widened only here
        cx.Ld[k] = (double)c.L[k];
        cx.invd[k] = (double)c.invL[k];
    }
    cx.r0f = (float)a.r0;
    cx.inv_wf = c.inv_w;
}

// One float32 distance evaluation + binning.  The bin arithmetic is unconditional (after
// culling nearly every wave step holds a candidate, so a branch around it would always be
// taken); only the histogram add is predicated.  An undecided pair is not evaluated here, a
// lane or two at a time with fp64 temporaries in the middle of the hot loop: it is appended to
// the wave's list in LDS (slot = list length + rank among the undecided lanes; the length lives
// in a scalar register) and cell_flush evaluates the list 64 pairs at a time.  Measured at C2:
// evaluating in place cost 15-20 % of the kernel (3 % of the wave steps took a ~70-instruction
// detour with one or two lanes active) and 66 KB of code (the detour inlined ~50 times).
// MODE: 0 per-wave LDS histograms, 1 global histogram (bin tables too large for LDS).
// TAGS: 0 no exclusion, 1 compare exclusion tags.
template <bool LOWER, int TAGS, int MODE, typename Hist>
__device__ inline void cell_step(const CellHot &c, const CellArgs &a, const Hist &hist, float fx,
                                 float fy, float fz, int tag_i, int tag_j, unsigned i_base,
                                 unsigned j_idx, unsigned w, CellWave &wv)
{
    float r2 = __fmaf_rn(fz, fz, __fmaf_rn(fy, fy, fx * fx));
    // pos = (sqrt(r2) - r0) / width - eta; raw v_sqrt_f32 (a denormal r2 ends on the exact path).
    // "Farther than eta from both neighbouring bin edges" is fract(pos) < 1 - 2 eta (see cell_filter).
    const float pos = __fmaf_rn(__builtin_amdgcn_sqrtf(r2), c.inv_w, c.pos0);
    const float t = __builtin_amdgcn_fractf(pos);
    unsigned long long m_todo;
    if (MODE == 1) {   // global histogram (bin tables too large for LDS): compiler-generated masks
        bool cand = LOWER ? (r2 < c.cand_hi && r2 >= c.cand_lo) : (r2 < c.cand_hi);
        if (TAGS)
            cand = cand && (tag_i != tag_j);
        const bool sure = t < c.sure_w;
        m_todo = __builtin_amdgcn_ballot_w64(cand && !sure);
        if (cand && sure && pos < (float)a.n_bins)
            hist.add((int)pos, w);
    } else {
        // The scalar unit issues one instruction per ~4 cycles per SIMD (scripts/issue_bench.hip), and
        // the compiler's predication of the add (s_andn2, s_cmp, s_and, s_and_saveexec, s_or exec +
        // two branches per step) made that unit, not the VALU, the bound of this loop.  Hand-placed:
        // the candidate test narrows EXEC itself (v_cmpx), the undecided test is one compare inside
        // it, the sure lanes are what is left, and EXEC is restored once — two scalar instructions.
        // EXEC is all ones here: every call site sits in wave-uniform control flow of full waves.
        const unsigned hbase = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned *)hist.h;
        const int hshift = cell_hist_shift(hist) + 2;   // byte offset of a bin = bin << (2 + log2 R)
        unsigned tmp;
        if (LOWER && TAGS)
            asm volatile("v_cmpx_gt_f32_e32 %[hi], %[r2]\n\tv_cmpx_le_f32_e32 %[lo], %[r2]\n\tv_cmpx_ne_u32_e32 %[ti], %[tj]\n\t"
                         "v_cmp_le_f32_e64 %[mt], %[sw], %[t]\n\tv_cvt_i32_f32_e32 %[tmp], %[pos]\n\t"
                         "s_andn2_b64 exec, exec, %[mt]\n\tv_lshl_add_u32 %[tmp], %[tmp], %[sh], %[hb]\n\t"
                         "ds_add_u32 %[tmp], %[w]\n\ts_mov_b64 exec, -1"
                         : [mt] "=&s"(m_todo), [tmp] "=&v"(tmp)
                         : [hi] "s"(c.cand_hi), [lo] "s"(c.cand_lo), [r2] "v"(r2), [sw] "s"(c.sure_w), [t] "v"(t),
                           [pos] "v"(pos), [hb] "v"(hbase), [w] "v"(w), [sh] "s"(hshift), [ti] "v"(tag_i), [tj] "v"(tag_j)
                         : "memory");
        else if (LOWER)
            asm volatile("v_cmpx_gt_f32_e32 %[hi], %[r2]\n\tv_cmpx_le_f32_e32 %[lo], %[r2]\n\t"
                         "v_cmp_le_f32_e64 %[mt], %[sw], %[t]\n\tv_cvt_i32_f32_e32 %[tmp], %[pos]\n\t"
                         "s_andn2_b64 exec, exec, %[mt]\n\tv_lshl_add_u32 %[tmp], %[tmp], %[sh], %[hb]\n\t"
                         "ds_add_u32 %[tmp], %[w]\n\ts_mov_b64 exec, -1"
                         : [mt] "=&s"(m_todo), [tmp] "=&v"(tmp)
                         : [hi] "s"(c.cand_hi), [lo] "s"(c.cand_lo), [r2] "v"(r2), [sw] "s"(c.sure_w), [t] "v"(t),
                           [pos] "v"(pos), [hb] "v"(hbase), [w] "v"(w), [sh] "s"(hshift)
                         : "memory");
        else if (TAGS)
            asm volatile("v_cmpx_gt_f32_e32 %[hi], %[r2]\n\tv_cmpx_ne_u32_e32 %[ti], %[tj]\n\t"
                         "v_cmp_le_f32_e64 %[mt], %[sw], %[t]\n\tv_cvt_i32_f32_e32 %[tmp], %[pos]\n\t"
                         "s_andn2_b64 exec, exec, %[mt]\n\tv_lshl_add_u32 %[tmp], %[tmp], %[sh], %[hb]\n\t"
                         "ds_add_u32 %[tmp], %[w]\n\ts_mov_b64 exec, -1"
                         : [mt] "=&s"(m_todo), [tmp] "=&v"(tmp)
                         : [hi] "s"(c.cand_hi), [r2] "v"(r2), [sw] "s"(c.sure_w), [t] "v"(t), [pos] "v"(pos),
                           [hb] "v"(hbase), [w] "v"(w), [sh] "s"(hshift), [ti] "v"(tag_i), [tj] "v"(tag_j)
                         : "memory");
        else
            asm volatile("v_cmpx_gt_f32_e32 %[hi], %[r2]\n\t"
                         "v_cmp_le_f32_e64 %[mt], %[sw], %[t]\n\tv_cvt_i32_f32_e32 %[tmp], %[pos]\n\t"
                         "s_andn2_b64 exec, exec, %[mt]\n\tv_lshl_add_u32 %[tmp], %[tmp], %[sh], %[hb]\n\t"
                         "ds_add_u32 %[tmp], %[w]\n\ts_mov_b64 exec, -1"
                         : [mt] "=&s"(m_todo), [tmp] "=&v"(tmp)
                         : [hi] "s"(c.cand_hi), [r2] "v"(r2), [sw] "s"(c.sure_w), [t] "v"(t), [pos] "v"(pos),
                           [hb] "v"(hbase), [w] "v"(w), [sh] "s"(hshift)
                         : "memory");
    }
    // An undecided pair is not evaluated here, a lane or two at a time with fp64 temporaries in the
    // middle of the hot loop: it is appended to the wave's list in LDS (slot = list length + rank
    // among the undecided lanes; the length lives in a scalar register) and cell_flush evaluates the
    // list 64 pairs at a time.
    if (__builtin_expect(m_todo != 0ull, 0)) {
        const unsigned cnt = (unsigned)__popcll(m_todo);
        if (wv.n_todo + cnt <= (unsigned)CELL_TODO) {
            if ((m_todo >> (threadIdx.x & 63u)) & 1ull) {
                const unsigned rank = __builtin_amdgcn_mbcnt_hi(
                    (unsigned)(m_todo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_todo, 0u));
                // i index = (wave-uniform base of the half tile) + lane, formed only here
                wv.todo[wv.n_todo + rank] =
                    make_uint2(i_base + (threadIdx.x & 63u), j_idx);   // j_idx carries the weight flag (bit 31)
            }
            wv.n_todo += cnt;
        } else {
            // the list is full (adversarial input: everything on a bin edge): the caller rolls
            // the list back to its mark and redoes the unit with cell_slow_unit
            wv.overflow = 1u;
        }
    }
}

// Exact arithmetic for the listed pairs, 64 at a time, one per lane (the list is wave-private).
template <typename Hist>
__device__ inline void cell_flush(const CellHot &c, const CellArgs &a, const double *sT,
                                  const Hist &hist, CellWave &wv, const CellOrig &po1f,
                                  const CellOrig &po2f)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PairCtx<true> cx;
    cell_exact_ctx(c, a, cx);
    const unsigned lane = threadIdx.x & 63u;
    for (unsigned e0 = 0; e0 < wv.n_todo; e0 += 64u) {
        if (e0 + lane < wv.n_todo) {
            const uint2 e = wv.todo[e0 + lane];
            cell_pair_exact(cx, a, sT, hist, po1f.at(e.x), po2f.at(e.y & 0x7fffffffu),
                            (e.y >> 31) + 1u, c.tri);
        }
    }
    wv.n_exact += wv.n_todo;
    wv.n_todo = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Redo of one unit whose undecided pairs did not fit the list: the same float32 filter (same
// operations, hence the same classification) one wave step at a time, the list flushed whenever
// the next step might not fit; sure pairs are skipped (the fast pass has binned them).
// sJw: the wave's slab of (pre-shifted) j rows [j0, j0 + nj); u_mask: which i halves the fast
// pass ran; general: per-pair image search.
template <bool LOWER, bool EXCL, typename Hist>
__device__ inline void cell_slow_unit(const CellHot &c, const CellArgs &a, const double *sT,
                                      const Hist &hist, const float4 *sJw, int j0, int nj,
                                      unsigned u_mask, bool tags, int general, const float *geo,
                                      const float4 &p0, const float4 &p1, const CellOrig &po1f,
                                      const CellOrig &po2f, unsigned i_idx0, unsigned jbase,
                                      unsigned w, CellWave &wv)
{
    for (int jj = j0; jj < j0 + nj; ++jj) {
        const float4 q = sJw[jj];
        for (int u = 0; u < 2; ++u) {
            if (!((u_mask >> u) & 1u))
                continue;
            const float4 &p = u ? p1 : p0;
            float fx = q.x - p.x, fy = q.y - p.y, fz = q.z - p.z;
            // the hot loop's arithmetic to the bit (MDX_CELL_GENERAL): the same pairs must come out
            // undecided here as there
            if (general & 1)
                fx = fminf(fabsf(fx), c.L[0] - fabsf(fx));
            if (general & 2)
                fy = fminf(fabsf(fy), c.L[1] - fabsf(fy));
            if (general & 4)
                fz = fminf(fabsf(fz), c.L[2] - fabsf(fz));
            float pos;
            bool cand, sure;
            cell_filter<LOWER, 0>(c, fx, fy, fz, 0, 0, pos, cand, sure);
            if (EXCL && tags)
                cand = cand && (__float_as_int(p.w) != __float_as_int(q.w));
            const bool todo = cand && !sure;
            const unsigned long long m_todo = __builtin_amdgcn_ballot_w64(todo);
            if (m_todo == 0ull)
                continue;
            if (wv.n_todo > (unsigned)CELL_TODO - 64u)
                cell_flush(c, a, sT, hist, wv, po1f, po2f);
            if (todo) {
                const unsigned rank = __builtin_amdgcn_mbcnt_hi(
                    (unsigned)(m_todo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_todo, 0u));
                wv.todo[wv.n_todo + rank] =
                    make_uint2(i_idx0 + 64u * u, (jbase + jj) | ((w - 1u) << 31));
            }
            wv.n_todo += (unsigned)__popcll(m_todo);
        }
    }
}

template <bool EXCL, bool LOWER, int MODE, bool TRI = false>
// (seven waves per SIMD: 72 VGPRs with 12 more bytes of scratch in the prologue than at six and 80 — +4 % at C2(i);
// eight do not fit the LDS of seven blocks and lose to their spills)
//
// Persistent blocks (round 3): the grid is what the chip holds at once (7 blocks x 256 CUs), and a block pulls
// (frame, i tile) items from a work counter until none is left.  The LDS histogram, the threshold table and the
// statistics live as long as the block: they are zeroed / loaded once and flushed once per BLOCK — with one block
// per item each of a launch's ~250 000 blocks ended in 201 64-bit global atomics (2 MB of the 3 MB a frame moved
// through HBM) and paid the block's launch, zeroing and flush.  Workgroups are dealt round-robin to the 8 XCDs
// in linear block order, so block b serves XCD b % 8 and takes only frames of that residue: a frame's sorted
// copies still stream through ONE XCD's L2.  Nothing waits on another block: a block that becomes resident late
// finds the counter exhausted and leaves.  What an item needs from the launch (block id, counter, sizes) is
// re-derived where it is used, behind an opaque copy of the block id (cell_opaque): carried across the hot loops
// as loop invariants, those values cost 13 scalar registers the kernel does not have (SGPR spills go to VGPR
// lanes, and with 72 VGPRs the hot loop then spills to scratch).
__global__ __launch_bounds__(256, TRI ? 4 : 7) void rdf_cell_pair_kernel(CellArgs a)
{
    constexpr bool GH = MODE == 1;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    float4 *sJ = reinterpret_cast<float4 *>(smem_raw);                        // [4 waves][64]
    double *sT = reinterpret_cast<double *>(smem_raw + sizeof(float4) * 256); // [n_bins+1]
    unsigned *sh = reinterpret_cast<unsigned *>(sT + (a.n_bins + 1));         // [n_bins + 1][n_hist replicas]
    __shared__ unsigned s_exact, s_units, s_general, s_qn, s_qnext, s_next;
    __shared__ unsigned sQ[CELL_QCAP];
    __shared__ uint2 s_todo[4][CELL_TODO];
    __shared__ float s_geo[48];
    __shared__ unsigned sQimg[TRI ? CELL_QCAP : 1];   // triclinic: surviving images of a queued tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // timing runs: every block adds its span in engine-clock ticks (s_memtime) and in ticks of the
    // constant 100 MHz counter (s_memrealtime); the ratio is the clock this kernel actually ran at
    long long clk0 = 0, rt0 = 0;
    if (a.clock_counter && tid == 0) {
        clk0 = clock64();
        rt0 = wall_clock64();
    }
    const int t64_2 = a.n2p / 64;

    if (!GH) {
        for (int b = tid; b <= a.n_bins; b += 256)
            sT[b] = a.thresh[b];
        for (int b = tid; b < a.n_hist * cell_hist_stride(a.n_bins); b += 256)
            sh[b] = 0u;
    }
    if (tid == 0) {
        s_exact = 0u;
        s_units = 0u;
        s_general = 0u;
        s_next = atomicAdd(a.work + CELL_WORK_STRIDE * (blockIdx.x & 7u), 1u);
    }
    __syncthreads();

    unsigned long long *out = a.counts + int64_t(blockIdx.x % unsigned(a.n_rep)) * a.n_bins;
    const double *thr = GH ? a.thresh : sT;
    // n_bins + 1 bins of n_hist (a power of two) interleaved replicas: the extra bin absorbs a (proven
    // impossible) index n_bins
    const int rep_log = __builtin_amdgcn_readfirstlane(31 - __clz(a.n_hist));
    HistLdsRep hl{sh + (GH ? 0 : (lane & (a.n_hist - 1))), GH ? 0 : rep_log};
    HistGlobal hg{out};
    CellHot hot{};

    float4 *sJw = sJ + wave * 64;
    CellWave wv{s_todo[wave], 0u, 0u, 0u};
    unsigned n_units = 0, n_general = 0;
    // upper bound of what any 32-bit LDS bin may hold, in units of 2^14 adds (= 128 i x 64 j x weight 2, one
    // j tile of one item); at 2^18 units the bins are flushed to the 64-bit replicas
    unsigned lds_units = 0;
    for (;;) {
    // the item: frame = 8 (item / tiles) + xcd, i tile = item % tiles, over the frames of this launch with
    // frame % 8 == xcd
    const unsigned item = __builtin_amdgcn_readfirstlane(s_next);
    int frame_l, I;
    {
        const unsigned xcd = cell_opaque(blockIdx.x) & 7u;
        const auto *A = cell_args();
        const unsigned tiles = unsigned(A->n1p) >> 7;                       // 128-particle i tiles of a frame
        if (A->spread) {
            const unsigned g = item * 8u + xcd;                             // item of the launch, dealt one by one
            if (g >= unsigned(A->n_frames) * tiles)
                break;
            const unsigned fq = __builtin_amdgcn_readfirstlane(g / tiles);
            frame_l = int(fq);
            I = int(g - fq * tiles);
        } else {
            const unsigned nfx = unsigned(A->n_frames + 7 - int(xcd)) >> 3;
            if (item >= nfx * tiles)
                break;
            // (the integer division runs on the VALU: pin its block-uniform results back into SGPRs,
            // or every pointer derived from them costs two VGPRs)
            const unsigned fq = __builtin_amdgcn_readfirstlane(item / tiles);
            frame_l = int(fq * 8u + xcd);
            I = int(item - fq * tiles);
        }
        frame_l += A->frame0;
    }
    const int frame = frame_l;
    // this frame's original coordinates, for the (cold) exact passes: formed where they are used
#define MDX_PO1F()                                                                                             \
    CellOrig { a.po1 ? a.po1 + int64_t(cell_opaque(frame)) * a.n1p : nullptr, a.pw1 + int64_t(cell_opaque(frame)) * a.n1p, \
               a.in1 ? a.in1 + int64_t(cell_opaque(frame)) * a.n1_in * 3 : nullptr }
#define MDX_PO2()                                                                                              \
    CellOrig { a.po2 ? a.po2 + int64_t(cell_opaque(frame)) * a.n2p : nullptr, a.pw2 + int64_t(cell_opaque(frame)) * a.n2p, \
               a.in2 ? a.in2 + int64_t(cell_opaque(frame)) * a.n2_in * 3 : nullptr }
    const float4 *PW2, *BB2, *BB16;
    {
        const auto *A = cell_args();
        PW2 = A->pw2 + int64_t(frame) * A->n2p;
        BB2 = A->bb2 + int64_t(frame) * t64_2 * 2;
        BB16 = A->bb16_2 + int64_t(frame) * t64_2 * 2 * CELL_NCHUNK;
    }
    {
    // Per-item geometry and the frame's constants are derived by one thread (fp64 error-bound
    // arithmetic included) and parked in LDS; the hot loop keeps only five floats of them, in SGPRs.
    //   s_geo: [0..2] L, [3..5] 1/L (float32), [6] cut, [7] cut^2,
    //          [8..10] cI, [11..13] hI, [14..19] cH[2][3], [20..25] hH[2][3],
    //          [26] cand_hi, [27] cand_lo, [28] inv_w, [29] -r0/w - eta, [30] 1 - 2 eta
    //   triclinic: [31..39] cell matrix B (rows a, b, c); L and 1/L are unused
    // (the previous item's last barrier has passed: nobody reads s_geo or the queue any more)
    if (tid == 0) {
        const auto *A = cell_args();
        {
            PairCtx<true> ctx;
            if (TRI) {
                // error bound of the float32 path: the shifted difference
                // fl(fl(x_j + t) - x_i) against the contract's (double)(x_j - x_i) + t deviates by less
                // than 7 * 2^-24 * sum|B| per component; the orthorhombic formula with the pseudo
                // length 4 sum|B| gives 2^-22 (2 M + 4 sum|B|) >= 2^-20 sum|B|
                const float *B = A->tri + int64_t(frame) * 9;
                float sum = 0.f;
                for (int i = 0; i < 9; ++i) {
                    s_geo[31 + i] = B[i];
                    sum += fabsf(B[i]);
                }
                const float pseudo[3] = {4.f * sum, 4.f * sum, 4.f * sum};
                ctx.init(pseudo, A->maxabs_bits, A->r0, A->r1, A->n_bins);
            } else {
                ctx.init(A->boxes + int64_t(frame) * 6, A->maxabs_bits, A->r0, A->r1, A->n_bins);
            }
            const float Lmax = fmaxf(ctx.Lf[0], fmaxf(ctx.Lf[1], ctx.Lf[2]));
            // a tile pair is culled when its box gap exceeds r1 + error bound + slack
            const float cut = sqrtf(ctx.cand_hi) + 1e-5f * Lmax;
            for (int k = 0; k < 3; ++k) {
                s_geo[k] = ctx.Lf[k];
                s_geo[3 + k] = ctx.invf[k];
            }
            s_geo[6] = cut;
            s_geo[7] = cut * cut;
            s_geo[26] = ctx.cand_hi;
            s_geo[27] = ctx.cand_lo;
            s_geo[28] = ctx.inv_wf;
            s_geo[29] = -ctx.r0f * ctx.inv_wf - ctx.eta;
            s_geo[30] = 1.0f - 2.0f * ctx.eta;
        }
        const float4 *BB1 = A->bb1 + int64_t(frame) * (A->n1p / 64) * 2 + int64_t(I) * 4;
        const float4 l0 = BB1[0], h0 = BB1[1], l1 = BB1[2], h1 = BB1[3];
        const float lo0[3] = {l0.x, l0.y, l0.z}, hi0[3] = {h0.x, h0.y, h0.z};
        const float lo1[3] = {l1.x, l1.y, l1.z}, hi1[3] = {h1.x, h1.y, h1.z};
        for (int k = 0; k < 3; ++k) {
            const float lo = fminf(lo0[k], lo1[k]), hi = fmaxf(hi0[k], hi1[k]);
            s_geo[8 + k] = 0.5f * (lo + hi);
            s_geo[11 + k] = 0.5f * (hi - lo);
            s_geo[14 + k] = 0.5f * (lo0[k] + hi0[k]);
            s_geo[17 + k] = 0.5f * (lo1[k] + hi1[k]);
            s_geo[20 + k] = 0.5f * (hi0[k] - lo0[k]);
            s_geo[23 + k] = 0.5f * (hi1[k] - lo1[k]);
        }
        s_qn = 0u;        // the queue of the item's first round
        s_qnext = 0u;
    }
    float4 p0, p1;
    {
        const auto *A = cell_args();
        const float4 *PW1 = A->pw1 + int64_t(frame) * A->n1p + int64_t(I) * 128;
        // (the lane's byte offset is formed here, per item: hoisted out of the item loop as a 64-bit value it was
        // spilled to scratch and reloaded — and written back — once per item: 4 KB of scratch traffic per item,
        // a sixth of what the kernel moved through HBM)
        int l = lane;
        asm volatile("" : "+v"(l));
        p0 = PW1[l];
        p1 = PW1[64 + l];
    }
    const unsigned i_base0 = unsigned(I) * 128u;   // wave-uniform; + lane = this lane's i index
    const unsigned i_idx0 = i_base0 + unsigned(lane);
    __syncthreads();
    {
        hot.cand_hi = cell_uniform(s_geo[26]);
        hot.cand_lo = cell_uniform(s_geo[27]);
        hot.inv_w = cell_uniform(s_geo[28]);
        hot.pos0 = s_geo[29];
        asm volatile("" : "+v"(hot.pos0));   // stays in a VGPR
        hot.sure_w = cell_uniform(s_geo[30]);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            hot.L[k] = cell_uniform(s_geo[k]);
            hot.invL[k] = cell_uniform(s_geo[3 + k]);
        }
        hot.tri = TRI ? s_geo + 31 : nullptr;
    }
    const int Jbeg = a.self ? 2 * I : 0;
    // Rounds of up to CELL_QCAP candidate j tiles: all four waves test candidates and append
    // the survivors to one LDS queue, then pull tiles from it one at a time (LDS atomic), so
    // the waves of a block stay evenly loaded whatever the spatial distribution of survivors.
    for (int round0 = Jbeg; round0 < t64_2; round0 += CELL_QCAP) {
        const int round1 = min(t64_2, round0 + CELL_QCAP);
        // 32-bit LDS bins: one round adds at most (128 i) x (64 j) x weight 2 = 2^14 per j tile to a bin.  Before
        // a round could take a bin past 2^32 the bins are flushed to the 64-bit replicas, so no count can wrap
        // however many items a block serves and however large the second set is (it could,
        // silently, from ~1.6e7 particles with coarse bins).  Uniform decision; the adds of earlier rounds are
        // behind a barrier, the next adds come after the barrier that closes the cull below.
        if (!GH && lds_units + unsigned(round1 - round0) >= cell_args()->flush_units) {
            for (int b = tid; b < a.n_bins; b += 256) {
                unsigned long long sum = 0;
                for (int h = 0; h < a.n_hist; ++h) {
                    sum += sh[b * a.n_hist + h];
                    sh[b * a.n_hist + h] = 0u;
                }
                if (sum)
                    atomicAdd(out + b, sum);
            }
            lds_units = 0u;
        }
        lds_units += unsigned(round1 - round0);
        if (round0 != Jbeg) {   // (the first round's queue was reset with the item's geometry)
            if (tid == 0) {
                s_qn = 0u;
                s_qnext = 0u;
            }
            __syncthreads();
        }
        for (int Jb = round0; Jb < round1; Jb += 256) {
            const int J = Jb + tid;
            float g2 = __int_as_float(0x7f800000);
            unsigned code = 0u;
            unsigned img_mask = 0u;
            if (J < round1) {
                const float4 lo = BB2[2 * J], hi = BB2[2 * J + 1];
                const float cJ[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
                const float hJ[3] = {0.5f * (hi.x - lo.x), 0.5f * (hi.y - lo.y), 0.5f * (hi.z - lo.z)};
                if (TRI) {
                    // every neighbouring image of the j tile is a candidate of its own: the cut is
                    // below half the smallest cell height (host check), so at most one image of
                    // a PAIR can come within the cut and nothing is counted twice
                    const float ex = s_geo[11] + hJ[0], ey = s_geo[12] + hJ[1], ez = s_geo[13] + hJ[2];
                    const float d0x = cJ[0] - s_geo[8], d0y = cJ[1] - s_geo[9], d0z = cJ[2] - s_geo[10];
#pragma unroll 1
                    for (int img = 0; img < 27; ++img) {
                        const float ia = float(img % 3 - 1), ib = float((img / 3) % 3 - 1),
                                    ic = float(img / 9 - 1);
                        const float tx = ia * s_geo[31] + ib * s_geo[34] + ic * s_geo[37];
                        const float ty = ib * s_geo[35] + ic * s_geo[38];
                        const float tz = ic * s_geo[39];
                        const float gx = fmaxf(0.f, fabsf(d0x + tx) - ex), gy = fmaxf(0.f, fabsf(d0y + ty) - ey),
                                    gz = fmaxf(0.f, fabsf(d0z + tz) - ez);
                        if (__fmaf_rn(gz, gz, __fmaf_rn(gy, gy, gx * gx)) <= s_geo[7])
                            img_mask |= 1u << img;
                    }
                    g2 = img_mask ? 0.f : __int_as_float(0x7f800000);
                    code = unsigned(J);
                } else {
                g2 = 0.f;
                int general = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const float Lk = s_geo[k], cut = s_geo[6];
                    float d = cJ[k] - s_geo[8 + k];
                    float sft = rintf(d * s_geo[3 + k]);
                    d = __fmaf_rn(-sft, Lk, d);
                    code |= unsigned((int)sft + 1) << (22 + 2 * k);
                    float ext = s_geo[11 + k] + hJ[k];
                    float reach = fabsf(d) + ext;
                    float gap = fmaxf(0.f, fabsf(d) - ext);
                    g2 = __fmaf_rn(gap, gap, g2);
                    // The shifted separation equals the minimum image for every pair whose
                    // |separation| stays below L/2 (all of them when reach < L/2).  A pair beyond
                    // L/2 is rejected by the filter (|sep| >= L/2 >= cut) and its true image
                    // component is L - |sep| >= L - reach: when that exceeds the cut it is out
                    // of range under the contract too, so the shifted value is harmless.
                    // (per dimension: bit k of `general` = this component needs the per-pair search)
                    const float halfL = 0.4999f * Lk;
                    general |= int(!(reach < halfL) && !(cut < halfL && reach < Lk - cut - 1e-4f * Lk)) << k;
                }
                code |= unsigned(J) | (unsigned(general) << 28);
                }
            }
            const bool keep = g2 <= s_geo[7];
            const unsigned long long mask = __ballot(keep);
            if (mask) {
                unsigned base = 0;
                if (lane == 0)
                    base = atomicAdd(&s_qn, (unsigned)__popcll(mask));
                base = __builtin_amdgcn_readfirstlane(base);
                if (keep) {
                    const unsigned slot = base + __popcll(mask & ((1ull << lane) - 1ull));
                    sQ[slot] = code;
                    if (TRI)
                        sQimg[slot] = img_mask;
                }
            }
        }
        __syncthreads();
        // the next item is requested while the other waves already pull tiles (the queue is dynamic: they take
        // what this wave misses while it waits for the counter); every wave has read s_next before the item's
        // first barrier, the next read comes after the round's last one
        if (tid == 0 && round0 == Jbeg)
            s_next = atomicAdd(cell_args()->work + CELL_WORK_STRIDE * (cell_opaque(blockIdx.x) & 7u), 1u);
        const unsigned nq = s_qn;
        while (true) {
            unsigned e = 0;
            if (lane == 0)
                e = atomicAdd(&s_qnext, 1u);
            e = __builtin_amdgcn_readfirstlane(e);
            if (e >= nq)
                break;
            // (wave-uniform, but read from LDS: pinned into a scalar register, or the tile index and every j index
            // derived from it live in vector registers and cost a VALU instruction per turn of the loops below)
            const unsigned code = __builtin_amdgcn_readfirstlane(sQ[e]);
            const int Jt = int(code & 0x3fffffu);
            const float4 pj_raw = PW2[int64_t(Jt) * 64 + lane];
            // orthorhombic: one pass with the tile pair's image; triclinic: one pass per
            // surviving image of the tile
            for (unsigned imgs = TRI ? sQimg[e] : 1u; imgs; imgs &= imgs - 1u) {
            int gen = 0;
            float sx, sy, sz;
            if (TRI) {
                const int img = __builtin_ctz(imgs);
                const float ia = float(img % 3 - 1), ib = float((img / 3) % 3 - 1), ic = float(img / 9 - 1);
                sx = -(ia * s_geo[31] + ib * s_geo[34] + ic * s_geo[37]);
                sy = -(ib * s_geo[35] + ic * s_geo[38]);
                sz = -(ic * s_geo[39]);
            } else {
                gen = int(code >> 28) & 7;
                sx = float(int((code >> 22) & 3u) - 1) * s_geo[0];
                sy = float(int((code >> 24) & 3u) - 1) * s_geo[1];
                sz = float(int((code >> 26) & 3u) - 1) * s_geo[2];
            }
            // a component that needs no search carries the tile pair's image shift; one that does stays
            // as wrapped (|difference| < L) and the loop below folds it
            float4 pj = pj_raw;
            if (!(gen & 1))
                pj.x -= sx;
            if (!(gen & 2))
                pj.y -= sy;
            if (!(gen & 4))
                pj.z -= sz;
            // second-level cull: lane l tests (j chunk l>>1) x (i half l&1) — 2 CELL_NCHUNK tests
            // (CELL_CHUNK == 1: every lane tests its own row against both i halves — two masks, sub for half 0 and
            // sub1 for half 1, no exchange between lanes; 15 % fewer steps than with two-row chunks at C2(i))
            constexpr unsigned long long SUB_ALL =
                CELL_NCHUNK >= 32 ? ~0ull : ((1ull << (2 * (CELL_NCHUNK & 31))) - 1ull);
            unsigned long long sub = SUB_ALL, sub1 = CELL_CHUNK == 1 ? ~0ull : 0ull;
            if (!gen) {
                float sg2 = __int_as_float(0x7f800000);
                if (CELL_CHUNK == 1) {
                    // (fmaxf drops a NaN: the NaN padding rows are taken out by a test of their own)
                    float g0 = 0.f, g1 = 0.f;
                    const float pc[3] = {pj.x, pj.y, pj.z};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float a0 = fmaxf(0.f, fabsf(pc[k] - s_geo[14 + k]) - s_geo[20 + k]);
                        const float a1 = fmaxf(0.f, fabsf(pc[k] - s_geo[17 + k]) - s_geo[23 + k]);
                        g0 = __fmaf_rn(a0, a0, g0);
                        g1 = __fmaf_rn(a1, a1, g1);
                    }
                    const bool real_row = pj.x == pj.x;
                    sub1 = __ballot(real_row && g1 <= s_geo[7]);
                    sg2 = real_row ? g0 : __int_as_float(0x7f800000);
                } else if (CELL_CHUNK == 2) {
                    // chunk = lane pair: its box comes from the staged rows themselves (one
                    // cross-lane exchange), no box array; NaN padding drops out of fmin / fmax,
                    // a chunk of two padding rows gives NaN and fails the comparison below
                    const int h = lane & 1;
                    const float ox = __shfl_xor(pj.x, 1), oy = __shfl_xor(pj.y, 1), oz = __shfl_xor(pj.z, 1);
                    const float lo[3] = {fminf(pj.x, ox), fminf(pj.y, oy), fminf(pj.z, oz)};
                    const float hi[3] = {fmaxf(pj.x, ox), fmaxf(pj.y, oy), fmaxf(pj.z, oz)};
                    sg2 = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        float gap = fmaxf(0.f, fabsf(0.5f * (lo[k] + hi[k]) - s_geo[14 + 3 * h + k]) -
                                                   (s_geo[20 + 3 * h + k] + 0.5f * (hi[k] - lo[k])));
                        sg2 = __fmaf_rn(gap, gap, sg2);
                    }
                } else if (lane < 2 * CELL_NCHUNK) {
                    const int s = lane >> 1, h = lane & 1;
                    const float4 lo = BB16[(Jt * CELL_NCHUNK + s) * 2],
                                 hi = BB16[(Jt * CELL_NCHUNK + s) * 2 + 1];
                    const float cJ[3] = {0.5f * (lo.x + hi.x) - sx, 0.5f * (lo.y + hi.y) - sy,
                                         0.5f * (lo.z + hi.z) - sz};
                    const float hJ[3] = {0.5f * (hi.x - lo.x), 0.5f * (hi.y - lo.y), 0.5f * (hi.z - lo.z)};
                    sg2 = 0.f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        float gap = fmaxf(0.f, fabsf(cJ[k] - s_geo[14 + 3 * h + k]) -
                                                   (s_geo[20 + 3 * h + k] + hJ[k]));
                        sg2 = __fmaf_rn(gap, gap, sg2);
                    }
                }
                sub = __ballot(sg2 <= s_geo[7]) & SUB_ALL;
            }
            // wave-private slab: LDS operations of one wave execute in order
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            sJw[lane] = pj;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const bool diag = a.self && Jt <= 2 * I + 1;
            unsigned w = (a.self && !diag) ? 2u : 1u;
            asm volatile("" : "+v"(w));   // the add's data operand: one VGPR per tile, not a v_mov per step
            // bit 31 of every j index handed on = (weight - 1): formed here on the scalar unit, once per tile
            // (derived from `w`, which is pinned in a vector register, it was a v_or per chunk)
            const unsigned jbase = unsigned(Jt) * 64u | ((a.self && !diag) ? 0x80000000u : 0u);
            // exclusion tags can only collide inside the diagonal tiles when exclusion == (1, 1)
            const bool tags = EXCL && (a.tags_everywhere || diag);
            if (!gen) {
                // The surviving (chunk, half) units of this tile.  Bookkeeping is per tile, not per
                // unit: one roll-back mark, one overflow test, one popcount for the statistics.
                const unsigned mark = wv.n_todo;
                n_units += (unsigned)__popcll(sub) + (CELL_CHUNK == 1 ? (unsigned)__popcll(sub1) : 0u);
#define MDX_CELL_HALF(TG, P, IB, Q, JJ)                                                            \
    if (GH) cell_step<LOWER, TG, MODE>(hot, a, hg, Q.x - P.x, Q.y - P.y, Q.z - P.z, __float_as_int(P.w), __float_as_int(Q.w), IB, (JJ), w, wv); \
    else cell_step<LOWER, TG, MODE>(hot, a, hl, Q.x - P.x, Q.y - P.y, Q.z - P.z, __float_as_int(P.w), __float_as_int(Q.w), IB, (JJ), w, wv);
// one unit: CELL_CHUNK slab rows (whole 16-byte reads: ds_read_b96 costs 8 LDS cycles,
// ds_read_b128 4) against the i halves that survived; the global j index is scalar
#define MDX_CELL_UNITS_CHUNKS(TG)                                                                         \
    for (unsigned long long rem = sub; rem;) {                                                     \
        const int s = __builtin_ctzll(rem) >> 1;                                                   \
        const unsigned bits = unsigned(rem >> (2 * s)) & 3u;                                       \
        rem &= ~(3ull << (2 * s));                                                                 \
        const unsigned jg = jbase + unsigned(CELL_CHUNK * s);                                      \
        float4 q[CELL_CHUNK];                                                                      \
        _Pragma("unroll") for (int c = 0; c < CELL_CHUNK; ++c) q[c] = sJw[CELL_CHUNK * s + c];     \
        /* all reads issued before any is waited for; .w kept alive = whole 16-byte reads */       \
        if (CELL_CHUNK == 2) asm volatile("" ::"v"(q[0].w), "v"(q[1].w));                          \
        else _Pragma("unroll") for (int c = 0; c < CELL_CHUNK; ++c) asm volatile("" ::"v"(q[c].w)); \
        if (bits & 1u) {                                                                           \
            _Pragma("unroll") for (int c = 0; c < CELL_CHUNK; ++c)                                 \
            {                                                                                      \
                MDX_CELL_HALF(TG, p0, i_base0, q[c], jg + unsigned(c))                             \
            }                                                                                      \
        }                                                                                          \
        if (bits & 2u) {                                                                           \
            _Pragma("unroll") for (int c = 0; c < CELL_CHUNK; ++c)                                 \
            {                                                                                      \
                MDX_CELL_HALF(TG, p1, i_base0 + 64u, q[c], jg + unsigned(c))                       \
            }                                                                                      \
        }                                                                                          \
    }
// CELL_CHUNK == 1: adjacent row pairs that survived against both halves (two slab reads, four steps), single rows
// that did (one read, two steps), then the rows of one half only — loops without a per-row test of which half
// applies
#define MDX_CELL_ROW_LOOP(TG, MASK, BODY)                                                          \
    for (unsigned long long rem = (MASK); rem;) {                                                  \
        const int r = __builtin_ctzll(rem);                                                        \
        asm("s_bitset0_b64 %0, %1" : "+s"(rem) : "s"(r));                                          \
        const float4 q = sJw[r];                                                                   \
        asm volatile("" ::"v"(q.w));   /* whole 16-byte read */                                    \
        BODY                                                                                       \
    }
// two adjacent rows per turn: both reads issued before the first wait
#define MDX_CELL_ROW2_LOOP(TG, MASK, BODY)                                                         \
    for (unsigned long long rem = (MASK); rem;) {                                                  \
        const int r = __builtin_ctzll(rem);                                                        \
        asm("s_bitset0_b64 %0, %1" : "+s"(rem) : "s"(r));                                          \
        const float4 q = sJw[r], q1 = sJw[r + 1];                                                  \
        asm volatile("" ::"v"(q.w), "v"(q1.w));                                                    \
        BODY                                                                                       \
    }
#define MDX_CELL_UNITS_ROWS(TG)                                                                    \
    {                                                                                              \
        const unsigned long long both = sub & sub1;                                                \
        /* rows (2 k, 2 k + 1) that both survived against both halves: two reads, four steps per turn */ \
        const unsigned long long pairs = both & (both >> 1) & 0x5555555555555555ull;               \
        const unsigned long long paired = pairs | (pairs << 1);                                    \
        MDX_CELL_ROW2_LOOP(TG, pairs,                                                              \
                          MDX_CELL_HALF(TG, p0, i_base0, q, jbase + unsigned(r))                   \
                          MDX_CELL_HALF(TG, p0, i_base0, q1, jbase + unsigned(r) + 1u)             \
                          MDX_CELL_HALF(TG, p1, i_base0 + 64u, q, jbase + unsigned(r))             \
                          MDX_CELL_HALF(TG, p1, i_base0 + 64u, q1, jbase + unsigned(r) + 1u))      \
        MDX_CELL_ROW_LOOP(TG, both & ~paired,                                                      \
                          MDX_CELL_HALF(TG, p0, i_base0, q, jbase + unsigned(r))                   \
                          MDX_CELL_HALF(TG, p1, i_base0 + 64u, q, jbase + unsigned(r)))            \
        /* rows of one half only, adjacent pairs first */                                          \
        const unsigned long long only0 = sub & ~sub1, only1 = sub1 & ~sub;                         \
        const unsigned long long pairs0 = only0 & (only0 >> 1) & 0x5555555555555555ull;            \
        const unsigned long long pairs1 = only1 & (only1 >> 1) & 0x5555555555555555ull;            \
        MDX_CELL_ROW2_LOOP(TG, pairs0,                                                             \
                          MDX_CELL_HALF(TG, p0, i_base0, q, jbase + unsigned(r))                   \
                          MDX_CELL_HALF(TG, p0, i_base0, q1, jbase + unsigned(r) + 1u))            \
        MDX_CELL_ROW2_LOOP(TG, pairs1,                                                             \
                          MDX_CELL_HALF(TG, p1, i_base0 + 64u, q, jbase + unsigned(r))             \
                          MDX_CELL_HALF(TG, p1, i_base0 + 64u, q1, jbase + unsigned(r) + 1u))      \
        MDX_CELL_ROW_LOOP(TG, only0 & ~(pairs0 | (pairs0 << 1)),                                   \
                          MDX_CELL_HALF(TG, p0, i_base0, q, jbase + unsigned(r)))                  \
        MDX_CELL_ROW_LOOP(TG, only1 & ~(pairs1 | (pairs1 << 1)),                                   \
                          MDX_CELL_HALF(TG, p1, i_base0 + 64u, q, jbase + unsigned(r)))            \
    }
#define MDX_CELL_UNITS(TG)                                                                         \
    if (CELL_CHUNK == 1) {                                                                         \
        MDX_CELL_UNITS_ROWS(TG)                                                                    \
    } else {                                                                                       \
        MDX_CELL_UNITS_CHUNKS(TG)                                                                  \
    }
                if (tags) {
                    MDX_CELL_UNITS(1)
                } else {
                    MDX_CELL_UNITS(0)
                }
#undef MDX_CELL_UNITS
#undef MDX_CELL_UNITS_ROWS
#undef MDX_CELL_ROW_LOOP
#undef MDX_CELL_ROW2_LOOP
#undef MDX_CELL_UNITS_CHUNKS
#undef MDX_CELL_HALF
                if (__builtin_expect(wv.overflow != 0u, 0)) {
                    // the list filled up somewhere in this tile (adversarial inputs): back to the
                    // mark, then every unit again, undecided pairs only, flushing as needed
                    wv.overflow = 0u;
                    wv.n_todo = mark;
                    for (unsigned long long rem = CELL_CHUNK == 1 ? (sub | sub1) : sub; rem;) {
                        const int s = CELL_CHUNK == 1 ? __builtin_ctzll(rem) : (__builtin_ctzll(rem) >> 1);
                        const unsigned bits = CELL_CHUNK == 1
                                                  ? (unsigned((sub >> s) & 1ull) | (unsigned((sub1 >> s) & 1ull) << 1))
                                                  : (unsigned(rem >> (2 * s)) & 3u);
                        rem &= CELL_CHUNK == 1 ? ~(1ull << s) : ~(3ull << (2 * s));
                        if (GH) cell_slow_unit<LOWER, EXCL>(hot, a, thr, hg, sJw, CELL_CHUNK * s, CELL_CHUNK, bits, tags, 0, s_geo, p0, p1, MDX_PO1F(), MDX_PO2(), i_idx0, jbase, w, wv);
                        else cell_slow_unit<LOWER, EXCL>(hot, a, thr, hl, sJw, CELL_CHUNK * s, CELL_CHUNK, bits, tags, 0, s_geo, p0, p1, MDX_PO1F(), MDX_PO2(), i_idx0, jbase, w, wv);
                    }
                }
            } else {
                // Tile pair that straddles half a box in the components of `gen`: those components take the
                // per-pair minimum image, as a magnitude (only the square is used): with both coordinates
                // wrapped, |f| < L and the image's magnitude is min(|f|, L - |f|) — two instructions where
                // f - L rint(f / L) takes three; one more rounding of at most 2^-25 L, inside the 4 * 2^-24 L
                // the bound delta of DESIGN.md §4.2 sets aside for the filter's own arithmetic.  The other
                // components were shifted with the tile.  One loop per set of components (7), so that a
                // tile pair straddling in x alone — the common case — pays for x alone.
                n_units += 2 * CELL_NCHUNK;
                n_general += 2 * CELL_NCHUNK;
                const unsigned mark = wv.n_todo;
#define MDX_CELL_GENERAL(GM)                                                                       \
    _Pragma("unroll 2") for (int jj = 0; jj < 64; ++jj)                                            \
    {                                                                                              \
        float4 q = sJw[jj];                                                                        \
        asm volatile("" ::"v"(q.w)); /* whole 16-byte read (ds_read_b96 costs twice the cycles) */ \
        _Pragma("unroll") for (int u = 0; u < 2; ++u)                                              \
        {                                                                                          \
            const float4 &p = u ? p1 : p0;                                                         \
            float fx = q.x - p.x, fy = q.y - p.y, fz = q.z - p.z;                                  \
            if ((GM) & 1) fx = fminf(fabsf(fx), hot.L[0] - fabsf(fx));                             \
            if ((GM) & 2) fy = fminf(fabsf(fy), hot.L[1] - fabsf(fy));                             \
            if ((GM) & 4) fz = fminf(fabsf(fz), hot.L[2] - fabsf(fz));                             \
            if (GH) cell_step<LOWER, EXCL ? 1 : 0, MODE>(hot, a, hg, fx, fy, fz, __float_as_int(p.w), __float_as_int(q.w), i_base0 + 64u * u, jbase + jj, w, wv); \
            else cell_step<LOWER, EXCL ? 1 : 0, MODE>(hot, a, hl, fx, fy, fz, __float_as_int(p.w), __float_as_int(q.w), i_base0 + 64u * u, jbase + jj, w, wv); \
        }                                                                                          \
    }
                switch (gen) {
                case 1: MDX_CELL_GENERAL(1) break;
                case 2: MDX_CELL_GENERAL(2) break;
                case 3: MDX_CELL_GENERAL(3) break;
                case 4: MDX_CELL_GENERAL(4) break;
                case 5: MDX_CELL_GENERAL(5) break;
                case 6: MDX_CELL_GENERAL(6) break;
                default: MDX_CELL_GENERAL(7) break;
                }
#undef MDX_CELL_GENERAL
                if (__builtin_expect(wv.overflow != 0u, 0)) {
                    wv.overflow = 0u;
                    wv.n_todo = mark;
                    if (GH) cell_slow_unit<LOWER, EXCL>(hot, a, thr, hg, sJw, 0, 64, 3u, true, gen, s_geo, p0, p1, MDX_PO1F(), MDX_PO2(), i_idx0, jbase, w, wv);
                    else cell_slow_unit<LOWER, EXCL>(hot, a, thr, hl, sJw, 0, 64, 3u, true, gen, s_geo, p0, p1, MDX_PO1F(), MDX_PO2(), i_idx0, jbase, w, wv);
                }
            }
            if (wv.n_todo >= 64u) {   // enough undecided pairs for a full-width exact pass
                if (GH) cell_flush(hot, a, thr, hg, wv, MDX_PO1F(), MDX_PO2());
                else cell_flush(hot, a, thr, hl, wv, MDX_PO1F(), MDX_PO2());
            }
            }   // images
        }
        // the undecided pairs of an item are evaluated before its last barrier: their indices are the item's
        // frame's, and the exact arithmetic reads the frame's constants
        if (round0 + CELL_QCAP >= t64_2 && wv.n_todo) {
            if (GH) cell_flush(hot, a, thr, hg, wv, MDX_PO1F(), MDX_PO2());
            else cell_flush(hot, a, thr, hl, wv, MDX_PO1F(), MDX_PO2());
        }
        __syncthreads();   // the queue, s_geo and s_next are rewritten after this
    }
    }
#undef MDX_PO1F
#undef MDX_PO2
    }   // items
    if (lane == 0) {
        if (wv.n_exact) atomicAdd(&s_exact, wv.n_exact);
        if (n_units) atomicAdd(&s_units, n_units);
        if (n_general) atomicAdd(&s_general, n_general);
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned so = rdf_stat_offset(blockIdx.x);
        if (s_exact) atomicAdd(a.exact_counter + so, (unsigned long long)s_exact);
        if (s_units) atomicAdd(a.tilepair_counter + so, (unsigned long long)s_units);
        if (s_general) atomicAdd(a.tilepair_counter + so + 1, (unsigned long long)s_general);
        if (a.clock_counter) {
            atomicAdd(a.clock_counter + so, (unsigned long long)(clock64() - clk0));
            atomicAdd(a.clock_counter + so + 1, (unsigned long long)(wall_clock64() - rt0));
        }
    }
    if (!GH) {
        for (int b = tid; b < a.n_bins; b += 256) {
            unsigned long long s = 0;
            for (int h = 0; h < a.n_hist; ++h)
                s += sh[b * a.n_hist + h];
            if (s)
                atomicAdd(out + b, s);
        }
    }
}
