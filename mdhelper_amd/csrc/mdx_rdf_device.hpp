// mdx_rdf_device.hpp — device-side pair evaluation shared by the RDF kernels.
//
// See mdx_rdf.hip for the result contract.  Two evaluators:
//
//   pair_exact   the contract arithmetic (float32 subtract, exact fp64 image
//                search, fp64 squares without contraction), then the bin from
//                the squared-distance threshold table;
//   pair_filter  a float32 distance whose deviation from the contract value is
//                bounded by `delta` per component (derivation in DESIGN.md §4.2);
//                a pair whose float32 distance lies farther than that bound from
//                every bin edge and from both range ends gets its bin from the
//                float32 value — provably the bin the contract arithmetic
//                gives — and every other pair falls through to pair_exact.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

// Statistics counters (pairs sent to the exact arithmetic, units evaluated, clock ticks) are kept in
// RDF_STAT_SHARDS shards of 16 words, one 128-byte line each; a block adds to the shard of its linear id.
// One shared set of counters cost a third of the pair kernel's time and hid every other change: a launch of
// 1 000 frames retires 256 000 blocks, each with up to five atomics on ONE line, and same-line device
// atomics complete one per ~24 ns (12.5 ms per launch measured with empty blocks).
constexpr unsigned RDF_STAT_SHARDS = 512;
constexpr unsigned RDF_STAT_STRIDE = 16;   // 64-bit words per shard
constexpr size_t RDF_STAT_BYTES = size_t(RDF_STAT_SHARDS) * RDF_STAT_STRIDE * 8;
__device__ inline unsigned rdf_stat_offset(unsigned linear_block)
{
    return (linear_block & (RDF_STAT_SHARDS - 1u)) * RDF_STAT_STRIDE;
}

struct RdfArgs {
    const float4 *p1;            // [frames][n1p]  packed xyz + exclusion tag
    const float4 *p2;            // [frames][n2p]
    const float *boxes;          // [frames][6] or nullptr
    const double *thresh;        // [n_bins+1]  T(edge_k) in the squared-distance domain
    unsigned long long *counts;  // [n_rep][n_bins]
    const unsigned *maxabs_bits; // max |coordinate| of the batch (float bits)
    unsigned long long *exact_counter;   // word 0 of the shards (rdf_stat_offset)
    double t_lo, t_hi;           // in range  <=>  t_lo <= rsq < t_hi
    double r0, r1;
    int n1p, n2p, nt1, nt2;
    int n_bins, n_hist, n_rep;
    int chunk, self, frame0;
};

template <bool PBC> struct PairCtx {
    // contract constants
    double Ld[3], invd[3];
    // float32 filter constants
    float Lf[3], invf[3];
    float r0f, inv_wf;       // bin estimate: (d - r0) * inv_w
    float cand_hi, cand_lo;  // candidate window on the float32 squared distance
    float eta;               // edge proximity (in bin units) that forces the exact path
    float posmax;            // n_bins - eta

    __device__ inline void init(const RdfArgs &a, int frame)
    {
        init(PBC ? a.boxes + int64_t(frame) * 6 : nullptr, a.maxabs_bits, a.r0, a.r1, a.n_bins);
    }

    // b: the frame's box (lx, ly, lz, ...) or nullptr without periodic boundaries
    __device__ inline void init(const float *b, const unsigned *maxabs_bits, double r0, double r1,
                                int n_bins)
    {
        double Lmax = 0.0;
        if (PBC) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Lf[k] = b[k];
                Ld[k] = (double)b[k];
                float inv = (float)(1.0 / (double)b[k]);
                invf[k] = inv;
                invd[k] = (double)inv;
                Lmax = fmax(Lmax, Ld[k]);
            }
        }
        const double M = (double)__uint_as_float(*maxabs_bits);
        // per-component bound on |w_contract - w_float32|, safety included
        const double delta = 0x1p-22 * (2.0 * M + Lmax);
        const double width = (r1 - r0) / n_bins;
        const double margin_d = 1.7320508075688772 * delta + r1 * 0x1p-21;
        r0f = (float)r0;
        inv_wf = (float)(1.0 / width);
        // Bin-coordinate budget.  Q = r1 / width is the largest magnitude the
        // float32 bin arithmetic handles (n_bins only when r0 = 0).  Per unit of Q:
        //   <= 0.625 * 2^-21  evaluating pos in float32 (sqrt 1 ulp, fma, 1/width, -r0/width),
        //   <= 1.07  * 2^-21  how far the candidate window reaches past r1 + margin (and below
        //                     r0 - margin) because of its own (1 +- 2^-20) safety factor,
        // so Q * 2^-20 covers both at once: a candidate beyond either range end is never "sure".
        const double Q = r1 / width;
        double e = margin_d / width + Q * 0x1p-20 + 0x1p-20;
        eta = (float)fmin(e, 1.0);
        posmax = (float)n_bins - eta;
        double hi = (r1 + margin_d);
        cand_hi = (float)(hi * hi * (1.0 + 0x1p-20));
        double lo = r0 - margin_d;
        cand_lo = lo > 0.0 ? (float)(lo * lo * (1.0 - 0x1p-20)) : -1.0f;
        if (!(M < 3.0e38))   // inf/NaN coordinates: everything goes to the exact path
            eta = 1.0f;
    }
};

// Bin of an in-range squared distance: float32 estimate, then exact correction
// against the threshold table (equals searchsorted(edges, d, 'right') - 1, last bin closed).
__device__ inline int rdf_bin_exact(double rsq, const double *sT, int n_bins, float r0f, float inv_wf)
{
    float d = __fsqrt_rn((float)rsq);
    int k = (int)((d - r0f) * inv_wf);
    k = max(0, min(k, n_bins - 1));
    while (k > 0 && rsq < sT[k])
        --k;
    while (k < n_bins - 1 && rsq >= sT[k + 1])
        ++k;
    return k;
}

template <bool PBC>
__device__ inline double rdf_rsq_contract(const PairCtx<PBC> &c, const float4 &pi, const float4 &pj)
{
    float fx = pj.x - pi.x, fy = pj.y - pi.y, fz = pj.z - pi.z;
    double dx = (double)fx, dy = (double)fy, dz = (double)fz;
    if (PBC) {
        // rint instead of C round(): they differ only on exact half-integers, where the
        // two images give w = +-L/2 and the same square
        double sx = __dmul_rn(c.invd[0], dx), sy = __dmul_rn(c.invd[1], dy),
               sz = __dmul_rn(c.invd[2], dz);
        dx = __dmul_rn(c.Ld[0], __dsub_rn(sx, rint(sx)));
        dy = __dmul_rn(c.Ld[1], __dsub_rn(sy, rint(sy)));
        dz = __dmul_rn(c.Ld[2], __dsub_rn(sz, rint(sz)));
    }
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz));
}

// per-wave LDS histogram (32-bit) or, for very many bins, the global 64-bit replica
struct HistLds {
    unsigned *h;
    __device__ inline void add(int k, unsigned w) const { atomicAdd(h + k, w); }
};
struct HistGlobal {
    unsigned long long *h;
    __device__ inline void add(int k, unsigned w) const { atomicAdd(h + k, (unsigned long long)w); }
};

template <bool PBC, bool EXCL, typename Hist>
__device__ inline void pair_exact(const PairCtx<PBC> &c, const RdfArgs &a, const double *sT,
                                  const Hist &hist, const float4 &pi, const float4 &pj, unsigned w)
{
    double rsq = rdf_rsq_contract<PBC>(c, pi, pj);
    bool in = (rsq >= a.t_lo) && (rsq < a.t_hi);
    if (EXCL)
        in = in && (__float_as_int(pi.w) != __float_as_int(pj.w));
    if (in) {
        int k = rdf_bin_exact(rsq, sT, a.n_bins, c.r0f, c.inv_wf);
        hist.add(k, w);
    }
}

template <bool PBC, bool EXCL, typename Hist>
__device__ inline void pair_filter(const PairCtx<PBC> &c, const RdfArgs &a, const double *sT,
                                   const Hist &hist, const float4 &pi, const float4 &pj, unsigned w,
                                   unsigned &n_exact)
{
    float fx = pj.x - pi.x, fy = pj.y - pi.y, fz = pj.z - pi.z;
    if (PBC) {
        fx = __fmaf_rn(-rintf(fx * c.invf[0]), c.Lf[0], fx);
        fy = __fmaf_rn(-rintf(fy * c.invf[1]), c.Lf[1], fy);
        fz = __fmaf_rn(-rintf(fz * c.invf[2]), c.Lf[2], fz);
    }
    float r2 = __fmaf_rn(fz, fz, __fmaf_rn(fy, fy, fx * fx));
    // NaN (padding) fails the first comparison
    bool cand = (r2 < c.cand_hi) && (r2 >= c.cand_lo);
    if (EXCL)
        cand = cand && (__float_as_int(pi.w) != __float_as_int(pj.w));
    if (cand) {
        float pos = (__fsqrt_rn(r2) - c.r0f) * c.inv_wf;
        float kf = floorf(pos);
        float fr = pos - kf;
        bool sure = (fr > c.eta) && (fr < 1.0f - c.eta) && (pos > c.eta) && (pos < c.posmax);
        int k = (int)kf;
        if (!sure) {
            ++n_exact;
            double rsq = rdf_rsq_contract<PBC>(c, pi, pj);
            if ((rsq >= a.t_lo) && (rsq < a.t_hi))
                k = rdf_bin_exact(rsq, sT, a.n_bins, c.r0f, c.inv_wf);
            else
                k = -1;
        }
        if (k >= 0)
            hist.add(k, w);
    }
}
