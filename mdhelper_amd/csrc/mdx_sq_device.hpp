// mdx_sq_device.hpp — device code shared by the structure-factor and ISF engines:
// fp64 sincos and the fused exp(i q.r) accumulation kernel (see mdx_sq.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <vector>

namespace mdx_sq_dev {
namespace {   // internal linkage: the header is compiled into two translation units

constexpr int SQ_THREADS = 256;
constexpr int SQ_QPT = 2;            // wavevectors per thread
constexpr int SQ_QPB = SQ_THREADS * SQ_QPT;
constexpr int SQ_TILE = 1024;        // particles per LDS stage

// sin and cos of x in fp64: Cody-Waite reduction by pi/2 (three-term, FMA), then the
// fdlibm minimax kernels on [-pi/4, pi/4].  |x| up to ~1e9 keeps ~1e-15 absolute error.
__device__ inline void sincos_f64(double x, double &s, double &c)
{
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_1 = 1.57079632679489655800e+00;   // pi/2 rounded to double
    const double PIO2_2 = 6.12323399573676603587e-17;   // pi/2 - PIO2_1
    const double PIO2_3 = -1.49738490485916983e-33;     // next 53 bits
    double kd = rint(x * TWO_OVER_PI);
    double r = fma(-kd, PIO2_1, x);
    r = fma(-kd, PIO2_2, r);
    r = fma(-kd, PIO2_3, r);
    int q = (int)(long long)kd;
    double z = r * r;
    // sin kernel
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double ps = fma(z, fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2), S1);
    double sn = fma(z * r, ps, r);
    // cos kernel
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double pc = fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    double cs = fma(z * z, pc, fma(-0.5, z, 1.0));
    // quadrant
    double so = (q & 1) ? cs : sn;
    double co = (q & 1) ? sn : cs;
    s = (q & 2) ? -so : so;
    c = ((q + 1) & 2) ? -co : co;
}

// The constants of sincos_f64 as wave-uniform values pinned to scalar registers (pin() once, outside the loops
// that call eval()): fp64 literals are not inline constants, and left to itself the compiler keeps all sixteen of
// them in vector registers across a whole kernel — 32 VGPRs of a register-blocked kernel that has none to spare —
// and copies one into place before every FMA of the polynomials.  eval() is sincos_f64 operation for operation
// (same bits); an FMA takes its constant straight from the scalar file.
struct SinCosScalars {
    double two_over_pi = 6.36619772367581382433e-01, p1 = 1.57079632679489655800e+00,
           p2 = 6.12323399573676603587e-17, p3 = -1.49738490485916983e-33;
    double s1 = -1.66666666666666324348e-01, s2 = 8.33333333332248946124e-03, s3 = -1.98412698298579493134e-04,
           s4 = 2.75573137070700676789e-06, s5 = -2.50507602534068634195e-08, s6 = 1.58969099521155010221e-10;
    double c1 = 4.16666666666666019037e-02, c2 = -1.38888888888741095749e-03, c3 = 2.48015872894767294178e-05,
           c4 = -2.75573143513906633035e-07, c5 = 2.08757232129817482790e-09, c6 = -1.13596475577881948265e-11;
    __device__ __forceinline__ void pin()
    {
        asm volatile("" : "+s"(two_over_pi), "+s"(p1), "+s"(p2), "+s"(p3));
        asm volatile("" : "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6));
        asm volatile("" : "+s"(c1), "+s"(c2), "+s"(c3), "+s"(c4), "+s"(c5), "+s"(c6));
    }
    __device__ __forceinline__ void eval(double x, double &s, double &c) const
    {
        double kd = rint(x * two_over_pi);
        double r = fma(-kd, p1, x);
        r = fma(-kd, p2, r);
        r = fma(-kd, p3, r);
        int q = (int)(long long)kd;
        double z = r * r;
        double ps = fma(z, fma(z, fma(z, fma(z, fma(z, s6, s5), s4), s3), s2), s1);
        double sn = fma(z * r, ps, r);
        double pc = fma(z, fma(z, fma(z, fma(z, fma(z, c6, c5), c4), c3), c2), c1);
        double cs = fma(z * z, pc, fma(-0.5, z, 1.0));
        double so = (q & 1) ? cs : sn;
        double co = (q & 1) ? sn : cs;
        s = (q & 2) ? -so : so;
        c = ((q + 1) & 2) ? -co : co;
    }
};

// rho[frame][group][split][q] (re, im)
__global__ __launch_bounds__(SQ_THREADS) void sq_rho_kernel(
    const float *__restrict__ pos, int64_t n_atoms, const double *__restrict__ qv, int n_q,
    const int64_t *__restrict__ group_offsets, int n_groups, int n_split,
    double2 *__restrict__ rho)
{
    __shared__ float sx[SQ_TILE], sy[SQ_TILE], sz[SQ_TILE];
    const int tid = threadIdx.x;
    const int qb = blockIdx.x;
    const int g = blockIdx.y / n_split, sp = blockIdx.y % n_split;
    const int frame = blockIdx.z;

    double q0[SQ_QPT], q1[SQ_QPT], q2[SQ_QPT], ac[SQ_QPT], as[SQ_QPT];
#pragma unroll
    for (int u = 0; u < SQ_QPT; ++u) {
        int qi = qb * SQ_QPB + u * SQ_THREADS + tid;
        bool ok = qi < n_q;
        q0[u] = ok ? qv[3 * int64_t(qi) + 0] : 0.0;
        q1[u] = ok ? qv[3 * int64_t(qi) + 1] : 0.0;
        q2[u] = ok ? qv[3 * int64_t(qi) + 2] : 0.0;
        ac[u] = 0.0;
        as[u] = 0.0;
    }
    const int64_t g_lo = group_offsets[g], g_hi = group_offsets[g + 1];
    const int64_t per = (g_hi - g_lo + n_split - 1) / n_split;
    const int64_t lo = g_lo + sp * per, hi = min(g_hi, lo + per);
    const float *P = pos + int64_t(frame) * n_atoms * 3;

    for (int64_t base = lo; base < hi; base += SQ_TILE) {
        const int cnt = (int)min<int64_t>(SQ_TILE, hi - base);
        __syncthreads();
        for (int e = tid; e < cnt * 3; e += SQ_THREADS) {
            float v = P[base * 3 + e];
            int a = e / 3, k = e - 3 * a;
            (k == 0 ? sx : k == 1 ? sy : sz)[a] = v;
        }
        __syncthreads();
        for (int a = 0; a < cnt; ++a) {
            const double x = (double)sx[a], y = (double)sy[a], z = (double)sz[a];
#pragma unroll
            for (int u = 0; u < SQ_QPT; ++u) {
                // a[0]*b[0] + a[1]*b[1] + a[2]*b[2]   (accelerated.py:43; fastmath there)
                double ph = fma(q2[u], z, fma(q1[u], y, q0[u] * x));
                double s, c;
                sincos_f64(ph, s, c);
                ac[u] += c;
                as[u] += s;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < SQ_QPT; ++u) {
        int qi = qb * SQ_QPB + u * SQ_THREADS + tid;
        if (qi < n_q)
            rho[((int64_t(frame) * n_groups + g) * n_split + sp) * n_q + qi] =
                make_double2(ac[u], as[u]);
    }
}


// ---------------------------------------------------------------------------------------
// Lattice fast path.  When every wavevector is an integer combination q = (m_x g_x, m_y g_y,
// m_z g_z) of one base vector per axis — the reference's default reciprocal grid
// 2 pi n / L (structure.py:1376-1381, 1404-1409) and any q_max-filtered subset of it —
//     exp(i q . r) = E_x(m_x) E_y(m_y) E_z(m_z),   E_k(m) = exp(i m g_k r_k)
// so one fp64 sincos per (particle, axis) and a complex recurrence fill small LDS tables,
// and each (q, particle) term costs two complex multiplies (8 fp64 FMA-class ops) instead of a
// sincos (~40).  Tables: tab_k[a][m - m_min_k], complex double.
struct SqLattice {
    double base[3];
    int mmin[3];
    int R[3];      // table length per axis
    int tile;      // particles per LDS stage
};

// Detects the lattice structure of a wavevector set (host).  trip: short[n_q][4] integer
// multiples (m_x, m_y, m_z, 0).  MDX_SQ_NO_LATTICE=1 disables the fast path.
inline bool sq_detect_lattice(const double *q, int64_t n_q, SqLattice &lat, std::vector<short> &trip)
{
    if (getenv("MDX_SQ_NO_LATTICE"))
        return false;
    trip.assign(size_t(4) * n_q, 0);
    int total = 0;
    for (int k = 0; k < 3; ++k) {
        double g = 0.0, big = 0.0;
        for (int64_t i = 0; i < n_q; ++i)
            big = std::max(big, std::fabs(q[3 * i + k]));
        for (int64_t i = 0; i < n_q; ++i) {
            double v = std::fabs(q[3 * i + k]);
            if (v > 1e-12 * std::max(big, 1e-300) && (g == 0.0 || v < g))
                g = v;
        }
        int mmin = 0, mmax = 0;
        if (g > 0.0) {
            for (int64_t i = 0; i < n_q; ++i) {
                double m = q[3 * i + k] / g, r = std::nearbyint(m);
                if (std::fabs(m - r) > 1e-9 * std::max(1.0, std::fabs(r)) || std::fabs(r) > 512)
                    return false;
                trip[4 * i + k] = (short)r;
                mmin = std::min(mmin, (int)r);
                mmax = std::max(mmax, (int)r);
            }
        }
        lat.base[k] = g;
        lat.mmin[k] = mmin;
        lat.R[k] = mmax - mmin + 1;
        total += lat.R[k];
    }
    // particles per LDS stage: as many as fit ~26 KiB of tables (6 blocks per CU), between 8 and 64
    int tile = int((26 * 1024) / (size_t(16) * total));
    if (tile < 8)
        return false;
    lat.tile = std::min(tile, 64);
    return true;
}

// row[m - mmin] = exp(i m theta) for m in [mmin, mmin + R): E(1) by sincos, the rest by recurrence
__device__ inline void sq_lattice_fill_row(double2 *row, double theta, int mmin, int R)
{
    double s1, c1;
    sincos_f64(theta, s1, c1);
    const int mmax = mmin + R - 1;
    double er = 1.0, ei = 0.0;                 // E(0)
    for (int m = 0; m <= mmax; ++m) {
        if (m >= mmin)
            row[m - mmin] = make_double2(er, ei);
        const double nr = fma(er, c1, -ei * s1), ni = fma(er, s1, ei * c1);
        er = nr;
        ei = ni;
    }
    er = c1;
    ei = -s1;                                  // E(-1)
    for (int m = -1; m >= mmin; --m) {
        if (m <= mmax)
            row[m - mmin] = make_double2(er, ei);
        const double nr = fma(er, c1, ei * s1), ni = fma(ei, c1, -er * s1);
        er = nr;
        ei = ni;
    }
}

// Column form of the lattice path.  A lattice wavevector set is a union of "columns": all q
// that share (m_x, m_y) and differ in m_z.  One thread owns up to SQ_ZPT wavevectors of one
// column: per particle it reads E_x(m_x), E_y(m_y) once, forms their product once, and then only
// needs E_z(m_z) — the same table entry for every lane of a wave when the columns carry the same
// m_z lists (a full grid), i.e. an LDS broadcast — and one complex multiply-add per wavevector:
// 1.25 LDS reads and 4.5 FMA-class operations per term instead of 3 and 8.
constexpr int SQ_ZPT = 8;
struct SqColumnItem {
    short i0, i1;          // table offsets m_x - mmin_x, m_y - mmin_y
    short nz, pad;
    short z[SQ_ZPT];       // table offsets m_z - mmin_z (unused entries: 0)
    int q[SQ_ZPT];         // wavevector index of each entry
};

__global__ __launch_bounds__(256) void sq_rho_columns_kernel(
    const float *__restrict__ pos, int64_t n_atoms, const SqColumnItem *__restrict__ items,
    int n_items, int n_q, SqLattice lat, const int64_t *__restrict__ group_offsets, int n_groups,
    int n_split, double2 *__restrict__ rho)
{
    extern __shared__ double2 lat_tab[];
    const int tid = threadIdx.x, T = blockDim.x;
    const int gid = blockIdx.x * T + tid;
    const int g = blockIdx.y / n_split, sp = blockIdx.y % n_split;
    const int frame = blockIdx.z;
    const int A = lat.tile;
    double2 *tab[3] = {lat_tab, lat_tab + size_t(A) * lat.R[0],
                       lat_tab + size_t(A) * (lat.R[0] + lat.R[1])};
    const SqColumnItem it = items[min(gid, n_items - 1)];
    double ar[SQ_ZPT], ai[SQ_ZPT];
#pragma unroll
    for (int j = 0; j < SQ_ZPT; ++j)
        ar[j] = ai[j] = 0.0;
    const int64_t g_lo = group_offsets[g], g_hi = group_offsets[g + 1];
    const int64_t per = (g_hi - g_lo + n_split - 1) / n_split;
    const int64_t lo = g_lo + sp * per, hi = min(g_hi, lo + per);
    const float *P = pos + int64_t(frame) * n_atoms * 3;

    for (int64_t base = lo; base < hi; base += A) {
        const int cnt = (int)min<int64_t>(A, hi - base);
        __syncthreads();
        for (int t = tid; t < cnt * 3; t += T) {
            const int a = t / 3, k = t - 3 * a;
            const double theta = lat.base[k] * (double)P[(base + a) * 3 + k];
            sq_lattice_fill_row(tab[k] + size_t(a) * lat.R[k], theta, lat.mmin[k], lat.R[k]);
        }
        __syncthreads();
        for (int a = 0; a < cnt; ++a) {
            const double2 ex = tab[0][size_t(a) * lat.R[0] + it.i0],
                          ey = tab[1][size_t(a) * lat.R[1] + it.i1];
            const double tr = fma(ex.x, ey.x, -ex.y * ey.y), ti = fma(ex.x, ey.y, ex.y * ey.x);
            const double2 *r2 = tab[2] + size_t(a) * lat.R[2];
#pragma unroll
            for (int j = 0; j < SQ_ZPT; ++j) {
                const double2 ez = r2[it.z[j]];
                ar[j] = fma(tr, ez.x, fma(-ti, ez.y, ar[j]));
                ai[j] = fma(tr, ez.y, fma(ti, ez.x, ai[j]));
            }
        }
    }
    if (gid < n_items) {
        double2 *out = rho + ((int64_t(frame) * n_groups + g) * n_split + sp) * n_q;
#pragma unroll
        for (int j = 0; j < SQ_ZPT; ++j)
            if (j < it.nz)
                out[it.q[j]] = make_double2(ar[j], ai[j]);
    }
}

// Register-blocked form of the column kernel: one thread owns SQ_QCOLS columns (m_x, m_y) that
// share a list of SQ_ZPT m_z values — a 4 x 8 block of accumulators fed by 8 + 8 table reads per
// particle instead of 2 + 8 per column (the column kernel is bound by its LDS reads: 1.25 16-byte
// reads per term against four fp64 FMAs).  Threads beyond the item count take further copies of
// the items for interleaved particles (sub = thread / items per block) and are summed at the end.
// Dynamic LDS: max(two table sets, 32 KB).
constexpr int SQ_QCOLS = 4;
constexpr int SQ_QUAD_THREADS = 256;

struct SqQuadItem {
    short i0[SQ_QCOLS], i1[SQ_QCOLS];   // table offsets of the columns
    short z[SQ_ZPT];                    // table offsets m_z - mmin_z
    int q[SQ_QCOLS][SQ_ZPT];            // wavevector index of each entry, -1: none
};

// Tables of a tile: tab[axis][m][particle] (particle fastest, row stride `lat.tile + SQ_QUAD_PAD`
// entries), two sets so that the next tile is filled while the current one is consumed: one
// barrier per tile.  A thread's particles of a tile are consecutive, which turns the particle
// index into the immediate offset of the ds_read instructions.
constexpr int SQ_QUAD_PAD = 1;

// theta = base_k * x (Pprev == nullptr) or base_k * (x - x_prev), the float32 coordinates widened
// first so that the difference is exact in fp64 (the ISF's displacement phases)
__device__ __forceinline__ void sq_quad_fill(double2 *set, const SqLattice &lat, int stride,
                                             const float *P, const float *Pprev, int64_t base, int cnt,
                                             int tid)
{
    for (int t = tid; t < cnt * 3; t += SQ_QUAD_THREADS) {
        const int a = t / 3, k = t - 3 * a;
        double x = (double)P[(base + a) * 3 + k];
        if (Pprev)
            x -= (double)Pprev[(base + a) * 3 + k];
        const double theta = lat.base[k] * x;
        double2 *col = set + size_t(k == 0 ? 0 : k == 1 ? lat.R[0] : lat.R[0] + lat.R[1]) * stride + a;
        double s1, c1;
        sincos_f64(theta, s1, c1);
        const int mmin = lat.mmin[k], mmax = mmin + lat.R[k] - 1;
        double er = 1.0, ei = 0.0;                 // E(0) upwards
        for (int m = 0; m <= mmax; ++m) {
            if (m >= mmin)
                col[size_t(m - mmin) * stride] = make_double2(er, ei);
            const double nr = fma(er, c1, -ei * s1), ni = fma(er, s1, ei * c1);
            er = nr;
            ei = ni;
        }
        er = c1;
        ei = -s1;                                  // E(-1) downwards
        for (int m = -1; m >= mmin; --m) {
            if (m <= mmax)
                col[size_t(m - mmin) * stride] = make_double2(er, ei);
            const double nr = fma(er, c1, ei * s1), ni = fma(ei, c1, -er * s1);
            er = nr;
            ei = ni;
        }
    }
}

// A thread's view of the quad tables: its item and copy, byte offsets of its 16 read streams
struct SqQuadThread {
    int item, sub;
    bool live;
    int stride, set_len, chunk;
    int o0[SQ_QCOLS], o1[SQ_QCOLS], oz[SQ_ZPT];
};

__device__ __forceinline__ SqQuadThread sq_quad_thread(const SqQuadItem *items, int n_items,
                                                       int ipb, int n_sub, const SqLattice &lat)
{
    // block b owns items [b ipb, (b + 1) ipb); thread = (copy, item of the block); threads past the last copy
    // (ipb n_sub < 256) and past the last item idle: they get no particles and write nothing
    SqQuadThread t;
    const int local = int(threadIdx.x) % ipb;
    t.item = blockIdx.x * ipb + local;
    t.sub = int(threadIdx.x) / ipb;
    t.live = t.item < n_items && t.sub < n_sub;
    t.stride = lat.tile + SQ_QUAD_PAD;
    t.set_len = (lat.R[0] + lat.R[1] + lat.R[2]) * t.stride;   // entries per table set
    t.chunk = lat.tile / n_sub;                                // particles per thread and tile
    const SqQuadItem *it = items + min(t.item, n_items - 1);
    const int first = min(t.sub, n_sub - 1) * t.chunk;   // (idle copies read where the last copy reads)
#pragma unroll
    for (int c = 0; c < SQ_QCOLS; ++c) {
        t.o0[c] = (it->i0[c] * t.stride + first) * 16;
        t.o1[c] = ((lat.R[0] + it->i1[c]) * t.stride + first) * 16;
    }
#pragma unroll
    for (int j = 0; j < SQ_ZPT; ++j)
        t.oz[j] = ((lat.R[0] + lat.R[1] + it->z[j]) * t.stride + first) * 16;
    return t;
}

// The regular form of sq_quad_frame (RS > 0: see there).  Beyond the item shape the host guarantees a SIMPLE
// lattice: every axis has m = 0 ... R - 1 with one R (the reference's grids, n = arange(n_points):
// structure.py:1376-1381), and lat.tile = RS - SQ_QUAD_PAD.  That makes the table fill branch-free and the same
// for every lane: a task is one (particle, axis) of the tile, rows are written at byte strides of ROW, and the coordinates are FETCHED A TILE AHEAD — the fill of tile n + 1 runs on values
// loaded before tile n - 1's particle loop, so no wave waits for memory between two tiles (the general fill loads
// its coordinate, its axis' base, mmin and R per lane and runs per-lane trip counts: ~175 instructions and one
// memory latency per tile and thread against ~100 and none here).  E(m) is the same chain of products as in
// sq_quad_fill (E(1) = 1 * (c, s) exactly), so both forms give the same bits.
template <bool REAL_ONLY, int RS>
__device__ __forceinline__ void sq_quad_frame_regular(double2 *lat_tab, const SqLattice &lat,
                                                      const SqQuadThread &t, const float *P, const float *Pprev,
                                                      int64_t lo, int64_t hi, double (&ar)[SQ_QCOLS][SQ_ZPT],
                                                      double (&ai)[SQ_QCOLS][SQ_ZPT])
{
    constexpr int A = RS - SQ_QUAD_PAD;                                          // particles per tile
    constexpr int NT = (3 * A + SQ_QUAD_THREADS - 1) / SQ_QUAD_THREADS;         // fill tasks per thread
    static_assert(NT == 1, "tiles of the regular form have at most SQ_QUAD_THREADS / 3 particles");
    constexpr int ROW = RS * 16;                                                // bytes between rows m, m + 1
    const int tid = threadIdx.x, R = lat.R[0];
    char *const tab = reinterpret_cast<char *>(lat_tab);
    // a thread's fill tasks (row offset in a set, its axis' base) and read streams are formed once and pinned: left
    // to rematerialise them inside the tile loop the compiler pairs the prefetched coordinate's register with an
    // address product and waits for the load a tile early
    // the fill task of a thread: particle fa of the tile, axis fk — AXIS-major (task = fk * A + fa), so that the
    // lanes of a ds_write_b128 group write consecutive entries of one row: particle-major tasks (fa = task / 3)
    // put the three axes of a particle on one bank group, a three-way conflict on every write (nearly all of the
    // 0.9 conflict cycles per 64 terms the counters showed)
    const int fk = min(tid / A, 2), fa = tid - (tid / A) * A;
    const bool filler = tid < 3 * A;
    int fo = (fk * R * RS + fa) * 16;
    double fb = fk == 0 ? lat.base[0] : fk == 1 ? lat.base[1] : lat.base[2];
    asm volatile("" : "+v"(fo));
    asm volatile("" : "+v"(fb));
    int px0 = t.o0[0], py0 = t.o1[0], pz0 = t.oz[0];
    asm volatile("" : "+v"(px0), "+v"(py0), "+v"(pz0));
    SinCosScalars sincos;
    sincos.pin();
    float xa, xb = 0.0f;
    // the thread's coordinate of the tile at `base` (cnt >= 1 particles); lanes past its end re-read the last particle
    auto fetch = [&](int64_t base, int cnt) {
        const int64_t at = (base + min(fa, cnt - 1)) * 3 + fk;
        xa = P[at];
        if (Pprev)
            xb = Pprev[at];
    };
    auto fill = [&](int set_bytes, int cnt) {
        if (filler && fa < cnt) {
            double x = (double)xa;
            if (Pprev)
                x -= (double)xb;
            const double theta = fb * x;
            double s1, c1;
            sincos.eval(theta, s1, c1);
            char *row = tab + (set_bytes + fo);
            *reinterpret_cast<double2 *>(row) = make_double2(1.0, 0.0);
            double er = c1, ei = s1;
            row += ROW;
            *reinterpret_cast<double2 *>(row) = make_double2(er, ei);
#pragma unroll 2
            for (int m = 2; m < R; ++m) {
                const double nr = fma(er, c1, -ei * s1), ni = fma(er, s1, ei * c1);
                er = nr;
                ei = ni;
                row += ROW;
                *reinterpret_cast<double2 *>(row) = make_double2(er, ei);
            }
        }
    };
    if (lo < hi) {
        fetch(lo, (int)min<int64_t>(A, hi - lo));
        fill(0, (int)min<int64_t>(A, hi - lo));
        if (lo + A < hi)
            fetch(lo + A, (int)min<int64_t>(A, hi - lo - A));
    }
    __syncthreads();
    int cur = 0;
    for (int64_t base = lo; base < hi; base += A, cur ^= 1) {
        const int cnt = (int)min<int64_t>(A, hi - base);
        if (base + A < hi) {
            fill((cur ^ 1) * t.set_len * 16, (int)min<int64_t>(A, hi - base - A));
            if (base + 2 * A < hi)
                fetch(base + 2 * A, (int)min<int64_t>(A, hi - base - 2 * A));
        }
        const int mine = t.live ? max(0, min(t.chunk, cnt - t.sub * t.chunk)) : 0;
        auto at = [&](int stream_bytes, int ib) {
            return *reinterpret_cast<const double2 *>(tab + (stream_bytes + ib));
        };
        // Software pipeline: the reads of the next particle's (e_x, e_y) and of the next e_z are issued before
        // the FMAs that hide their latency (the scheduler, short of registers, otherwise waits for every read
        // right after issuing it).  The read past a thread's last particle stays inside the allocation.
        const int ib0 = cur * t.set_len * 16;
        int px = px0 + ib0, py = py0 + ib0, pz = pz0 + ib0;
        double2 ex = at(px, 0), ey[SQ_QCOLS];
#pragma unroll
        for (int c = 0; c < SQ_QCOLS; ++c)
            ey[c] = at(py, c * ROW);
        // one particle; `io` = its byte offset from the three stream addresses (an immediate of the reads)
        auto particle = [&](int io) {
            // the first e_z is asked for before the column products, which hide its latency; LDS answers in order,
            // so the wait before the first terms is for this read alone
            double2 ez = at(pz, io);
            __builtin_amdgcn_sched_barrier(0);
            double tr[SQ_QCOLS], ti[SQ_QCOLS];
#pragma unroll
            for (int c = 0; c < SQ_QCOLS; ++c) {
                tr[c] = fma(ex.x, ey[c].x, -ex.y * ey[c].y);
                ti[c] = fma(ex.x, ey[c].y, ex.y * ey[c].x);
            }
            __builtin_amdgcn_sched_barrier(0);
            ex = at(px, io + 16);
#pragma unroll
            for (int c = 0; c < SQ_QCOLS; ++c)
                ey[c] = at(py, io + 16 + c * ROW);
#pragma unroll
            for (int j = 0; j < SQ_ZPT; ++j) {
                double2 ezn = ez;
                if (j + 1 < SQ_ZPT)
                    ezn = at(pz, io + (j + 1) * ROW);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < SQ_QCOLS; ++c) {
                    ar[c][j] = fma(tr[c], ez.x, fma(-ti[c], ez.y, ar[c][j]));
                    if (!REAL_ONLY)
                        ai[c][j] = fma(tr[c], ez.y, fma(ti[c], ez.x, ai[c][j]));
                }
                __builtin_amdgcn_sched_barrier(0);
                ez = ezn;
            }
        };
        // U particles per trip (a tile of 80 gives every thread a multiple of five, the other tiles of four):
        // one set of address updates and one branch per U particles
        constexpr int U = A % 5 == 0 ? 5 : 4;
        int i = 0;
#pragma unroll 1
        for (; i + U <= mine; i += U) {
#pragma unroll
            for (int u = 0; u < U; ++u)
                particle(16 * u);
            px += 16 * U;
            py += 16 * U;
            pz += 16 * U;
        }
#pragma unroll 1
        for (; i < mine; ++i) {
            particle(0);
            px += 16;
            py += 16;
            pz += 16;
        }
        __syncthreads();
    }
}

// Particles [lo, hi) of one frame (or of one pair of frames: Pprev) into the thread's 4 x 8
// accumulators; REAL_ONLY keeps Re(E_x E_y E_z) only (two FMAs per term).  Ends on a barrier.
// RS > 0: every item is REGULAR — its four columns share m_x and have consecutive m_y, its eight m_z are
// consecutive — and the table rows are RS entries apart (a compile-time constant: tile + SQ_QUAD_PAD).  A thread
// then reads ONE e_x, four e_y and eight e_z per particle, the latter two at immediate offsets c * RS * 16 from
// one address each: 13 LDS reads and three address registers where the general form has 16 and 16 (the address
// arithmetic was 0.5 of the 6.3 VALU instructions per 64 terms, the reads 0.5 per term).  Full grids — the
// reference's default wavevector sets, structure.py:1376-1381 — are regular; anything else takes RS = 0.
template <bool REAL_ONLY, int RS>
__device__ __forceinline__ void sq_quad_frame(double2 *lat_tab, const SqLattice &lat,
                                              const SqQuadThread &t, const float *P, const float *Pprev,
                                              int64_t lo, int64_t hi, double (&ar)[SQ_QCOLS][SQ_ZPT],
                                              double (&ai)[SQ_QCOLS][SQ_ZPT])
{
    if constexpr (RS > 0) {
        sq_quad_frame_regular<REAL_ONLY, RS>(lat_tab, lat, t, P, Pprev, lo, hi, ar, ai);
        return;
    }
    const int tid = threadIdx.x, A = lat.tile;
    if (lo < hi)
        sq_quad_fill(lat_tab, lat, t.stride, P, Pprev, lo, (int)min<int64_t>(A, hi - lo), tid);
    __syncthreads();
    int cur = 0;
    for (int64_t base = lo; base < hi; base += A, cur ^= 1) {
        const int cnt = (int)min<int64_t>(A, hi - base);
        if (base + A < hi)
            sq_quad_fill(lat_tab + size_t(cur ^ 1) * t.set_len, lat, t.stride, P, Pprev, base + A,
                         (int)min<int64_t>(A, hi - base - A), tid);
        const int mine = t.live ? max(0, min(t.chunk, cnt - t.sub * t.chunk)) : 0;
        // byte address of a read = stream offset (per thread) + ib (set and particle)
        auto at = [&](int stream_bytes, int ib) {
            return *reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(lat_tab) +
                                                      (stream_bytes + ib));
        };
        // Software pipeline: the reads of the next particle's (e_x, e_y) and of the next e_z are
        // issued before the FMAs that hide their latency (the scheduler, short of registers,
        // otherwise waits for every read right after issuing it).  The read past a thread's
        // last particle stays inside the allocation and is discarded.
        const int ib0 = cur * t.set_len * 16;
        double2 ex[SQ_QCOLS], ey[SQ_QCOLS];
#pragma unroll
        for (int c = 0; c < SQ_QCOLS; ++c) {
            ex[c] = at(t.o0[c], ib0);
            ey[c] = at(t.o1[c], ib0);
        }
#pragma unroll 1
        for (int i = 0; i < mine; ++i) {
            const int ib = ib0 + i * 16;
            double tr[SQ_QCOLS], ti[SQ_QCOLS];
#pragma unroll
            for (int c = 0; c < SQ_QCOLS; ++c) {
                tr[c] = fma(ex[c].x, ey[c].x, -ex[c].y * ey[c].y);
                ti[c] = fma(ex[c].x, ey[c].y, ex[c].y * ey[c].x);
            }
            double2 ez = at(t.oz[0], ib);
#pragma unroll
            for (int c = 0; c < SQ_QCOLS; ++c) {
                ex[c] = at(t.o0[c], ib + 16);
                ey[c] = at(t.o1[c], ib + 16);
            }
#pragma unroll
            for (int j = 0; j < SQ_ZPT; ++j) {
                double2 ezn = ez;
                if (j + 1 < SQ_ZPT)
                    ezn = at(t.oz[j + 1], ib);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < SQ_QCOLS; ++c) {
                    ar[c][j] = fma(tr[c], ez.x, fma(-ti[c], ez.y, ar[c][j]));
                    if (!REAL_ONLY)
                        ai[c][j] = fma(tr[c], ez.y, fma(ti[c], ez.x, ai[c][j]));
                }
                __builtin_amdgcn_sched_barrier(0);
                ez = ezn;
            }
        }
        __syncthreads();
    }
}

// The copies of an item are summed in a fixed order through LDS (the table space, one column of
// the block at a time: 16 doubles per thread) and written by copy 0: out_c[q] (complex) or
// out_r[q] (REAL_ONLY).
template <bool REAL_ONLY>
__device__ __forceinline__ void sq_quad_store(double2 *lat_tab, const SqQuadThread &t,
                                              const SqQuadItem *items, int n_items, int ipb,
                                              int n_sub, const double (&ar)[SQ_QCOLS][SQ_ZPT],
                                              const double (&ai)[SQ_QCOLS][SQ_ZPT], double2 *out_c,
                                              double *out_r)
{
    constexpr int T = SQ_QUAD_THREADS;
    const int tid = threadIdx.x;
    const SqQuadItem *it = items + min(t.item, n_items - 1);
    double *red = reinterpret_cast<double *>(lat_tab);
#pragma unroll
    for (int c = 0; c < SQ_QCOLS; ++c) {
        if (n_sub > 1) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < SQ_ZPT; ++j) {
                red[(2 * j) * T + tid] = ar[c][j];
                if (!REAL_ONLY)
                    red[(2 * j + 1) * T + tid] = ai[c][j];
            }
            __syncthreads();
        }
        if (t.live && t.sub == 0) {
#pragma unroll
            for (int j = 0; j < SQ_ZPT; ++j) {
                const int q = it->q[c][j];
                if (q < 0)
                    continue;
                double re = ar[c][j], im = REAL_ONLY ? 0.0 : ai[c][j];
                for (int s2 = 1; s2 < n_sub; ++s2) {
                    re += red[(2 * j) * T + s2 * ipb + tid];
                    if (!REAL_ONLY)
                        im += red[(2 * j + 1) * T + s2 * ipb + tid];
                }
                if (REAL_ONLY)
                    out_r[q] = re;
                else
                    out_c[q] = make_double2(re, im);
            }
        }
    }
}

template <int RS>
__global__ __launch_bounds__(SQ_QUAD_THREADS, 2) void sq_rho_quads_kernel(
    const float *__restrict__ pos, int64_t n_atoms, const SqQuadItem *__restrict__ items,
    int n_items, int ipb, int n_sub, int n_q, SqLattice lat,
    const int64_t *__restrict__ group_offsets, int n_groups, int n_split, double2 *__restrict__ rho)
{
    extern __shared__ double2 lat_tab[];
    const SqQuadThread t = sq_quad_thread(items, n_items, ipb, n_sub, lat);
    const int g = blockIdx.y / n_split, sp = blockIdx.y % n_split;
    const int frame = blockIdx.z;
    double ar[SQ_QCOLS][SQ_ZPT], ai[SQ_QCOLS][SQ_ZPT];
#pragma unroll
    for (int c = 0; c < SQ_QCOLS; ++c)
#pragma unroll
        for (int j = 0; j < SQ_ZPT; ++j)
            ar[c][j] = ai[c][j] = 0.0;
    const int64_t g_lo = group_offsets[g], g_hi = group_offsets[g + 1];
    const int64_t per = (g_hi - g_lo + n_split - 1) / n_split;
    const int64_t lo = g_lo + sp * per, hi = min(g_hi, lo + per);
    sq_quad_frame<false, RS>(lat_tab, lat, t, pos + int64_t(frame) * n_atoms * 3, nullptr, lo, hi, ar, ai);
    sq_quad_store<false>(lat_tab, t, items, n_items, ipb, n_sub, ar, ai,
                         rho + ((int64_t(frame) * n_groups + g) * n_split + sp) * n_q, nullptr);
}

// Incoherent ISF through the same blocking: part[split][lag][slot][q] = sum over the block's new
// frames f >= lag and the slot's particles of cos q.(r(f) - r(f - lag)) — only the real part of
// E_x E_y E_z is accumulated, two FMAs per term.  grid (item blocks, slots x splits, lags).
template <int RS>
__global__ __launch_bounds__(SQ_QUAD_THREADS, 2) void isf_incoherent_quads_kernel(
    const float *__restrict__ pos_ring, int ring_slots, int64_t n_atoms,
    const SqQuadItem *__restrict__ items, int n_items, int ipb, int n_sub, int n_q,
    SqLattice lat, const int64_t *__restrict__ ranges /*[n_slots][2]*/, int n_slots, int n_split,
    int n_lags, long long f_first, int n_new, double *__restrict__ part)
{
    extern __shared__ double2 lat_tab[];
    const SqQuadThread t = sq_quad_thread(items, n_items, ipb, n_sub, lat);
    const int slot = blockIdx.y / n_split, sp = blockIdx.y % n_split;
    const int lag = blockIdx.z;
    double ar[SQ_QCOLS][SQ_ZPT], ai[SQ_QCOLS][SQ_ZPT];
#pragma unroll
    for (int c = 0; c < SQ_QCOLS; ++c)
#pragma unroll
        for (int j = 0; j < SQ_ZPT; ++j)
            ar[c][j] = ai[c][j] = 0.0;
    const int64_t g_lo = ranges[2 * slot], g_hi = ranges[2 * slot + 1];
    const int64_t per = (g_hi - g_lo + n_split - 1) / n_split;
    const int64_t lo = g_lo + sp * per, hi = min(g_hi, lo + per);
    for (int i = 0; i < n_new; ++i) {
        const long long f = f_first + i;
        if (f < lag)
            continue;
        sq_quad_frame<true, RS>(lat_tab, lat, t, pos_ring + int64_t(f % ring_slots) * n_atoms * 3,
                            pos_ring + int64_t((f - lag) % ring_slots) * n_atoms * 3, lo, hi, ar, ai);
    }
    sq_quad_store<true>(lat_tab, t, items, n_items, ipb, n_sub, ar, ai, nullptr,
                        part + ((int64_t(sp) * n_lags + lag) * n_slots + slot) * n_q);
}

// The instantiated row strides (tile + SQ_QUAD_PAD) of the regular form; 0 = general items.  Tiles of at most 80
// particles: every thread has at most one fill task per tile (static_assert in sq_quad_frame_regular).
#define MDX_SQ_QUAD_STRIDES(X) X(17) X(33) X(49) X(65) X(81)
using SqRhoQuadsFn = void (*)(const float *, int64_t, const SqQuadItem *, int, int, int, int, SqLattice,
                              const int64_t *, int, int, double2 *);
using IsfIncQuadsFn = void (*)(const float *, int, int64_t, const SqQuadItem *, int, int, int, int, SqLattice,
                               const int64_t *, int, int, int, long long, int, double *);
inline SqRhoQuadsFn sq_rho_quads_pick(int regular_stride)
{
    switch (regular_stride) {
#define MDX_CASE(S) case S: return sq_rho_quads_kernel<S>;
        MDX_SQ_QUAD_STRIDES(MDX_CASE)
#undef MDX_CASE
    default: return sq_rho_quads_kernel<0>;
    }
}
inline IsfIncQuadsFn isf_incoherent_quads_pick(int regular_stride)
{
    switch (regular_stride) {
#define MDX_CASE(S) case S: return isf_incoherent_quads_kernel<S>;
        MDX_SQ_QUAD_STRIDES(MDX_CASE)
#undef MDX_CASE
    default: return isf_incoherent_quads_kernel<0>;
    }
}

// Host: quad items of a detected lattice set.  Columns are cut into chunks of SQ_ZPT consecutive
// m_z; chunks with the same m_z list are grouped four at a time (the last group of a list is
// padded with a repeated column that writes nothing).  Returns false when less than 60 % of the
// accumulators would be used (scattered sets: the column kernel serves those).
inline bool sq_build_quads(const std::vector<short> &trip, int64_t n_q, const SqLattice &lat,
                           std::vector<SqQuadItem> &items)
{
    struct Entry { int mx, my, mz; int q; };
    std::vector<Entry> e((size_t)n_q);
    for (int64_t i = 0; i < n_q; ++i)
        e[(size_t)i] = {trip[4 * i], trip[4 * i + 1], trip[4 * i + 2], (int)i};
    std::sort(e.begin(), e.end(), [](const Entry &a, const Entry &b) {
        if (a.mx != b.mx) return a.mx < b.mx;
        if (a.my != b.my) return a.my < b.my;
        if (a.mz != b.mz) return a.mz < b.mz;
        return a.q < b.q;
    });
    struct Chunk { std::vector<short> z; short i0, i1; std::vector<int> q; };
    std::vector<Chunk> chunks;
    size_t i = 0;
    while (i < e.size()) {
        size_t j = i;
        while (j < e.size() && e[j].mx == e[i].mx && e[j].my == e[i].my)
            ++j;
        for (size_t c = i; c < j; c += SQ_ZPT) {
            Chunk ch;
            ch.i0 = (short)(e[i].mx - lat.mmin[0]);
            ch.i1 = (short)(e[i].my - lat.mmin[1]);
            for (size_t k = c; k < std::min(j, c + SQ_ZPT); ++k) {
                ch.z.push_back((short)(e[k].mz - lat.mmin[2]));
                ch.q.push_back(e[k].q);
            }
            chunks.push_back(std::move(ch));
        }
        i = j;
    }
    std::stable_sort(chunks.begin(), chunks.end(),
                     [](const Chunk &a, const Chunk &b) { return a.z < b.z; });
    items.clear();
    for (size_t c = 0; c < chunks.size();) {
        size_t d = c;
        while (d < chunks.size() && chunks[d].z == chunks[c].z)
            ++d;
        for (size_t k = c; k < d; k += SQ_QCOLS) {
            SqQuadItem it{};
            for (int col = 0; col < SQ_QCOLS; ++col) {
                const bool have = k + col < d;
                const Chunk &ch = chunks[have ? k + col : k];
                it.i0[col] = ch.i0;
                it.i1[col] = ch.i1;
                for (int z = 0; z < SQ_ZPT; ++z)
                    it.q[col][z] = (have && z < (int)ch.q.size()) ? ch.q[(size_t)z] : -1;
            }
            for (int z = 0; z < SQ_ZPT; ++z)
                it.z[z] = z < (int)chunks[c].z.size() ? chunks[c].z[(size_t)z] : chunks[c].z[0];
            items.push_back(it);
        }
        c = d;
    }
    return !items.empty() && double(n_q) >= 0.6 * double(items.size()) * SQ_QCOLS * SQ_ZPT;
}

// Host: REGULAR quad items — the set covered by aligned blocks of one m_x, four consecutive m_y (from a multiple of
// four) and eight consecutive m_z (from a multiple of eight); entries of a block that are not wavevectors of the set
// are computed and dropped (q = -1).  Needs m >= 0 on every axis (the reference's grids, n = arange(n_points),
// structure.py:1376-1381, and their q_max-filtered subsets, structure.py:1412-1414) and no wavevector twice.  The
// tables then run m = 0 ... R - 1 with ONE R for the three axes (`out`: the largest index, rounded up so that every
// block's rows exist) — what sq_quad_frame_regular's fill assumes.  A full grid gives the items the general builder
// gives it; a sphere octant uses 60-85 % of its accumulators.  `used` = that fraction.
inline bool sq_build_quad_blocks(const std::vector<short> &trip, int64_t n_q, const SqLattice &lat,
                                 std::vector<SqQuadItem> &items, SqLattice &out, double &used)
{
    if (lat.mmin[0] < 0 || lat.mmin[1] < 0 || lat.mmin[2] < 0)
        return false;
    const int top[3] = {lat.mmin[0] + lat.R[0], lat.mmin[1] + lat.R[1], lat.mmin[2] + lat.R[2]};   // max m + 1
    const int r = std::max(top[0], std::max((top[1] + SQ_QCOLS - 1) / SQ_QCOLS * SQ_QCOLS,
                                            (top[2] + SQ_ZPT - 1) / SQ_ZPT * SQ_ZPT));
    if (r > 1024)
        return false;
    struct Key { int zb, mx, yb; };
    auto less = [](const Key &a, const Key &b) {
        return a.zb != b.zb ? a.zb < b.zb : a.mx != b.mx ? a.mx < b.mx : a.yb < b.yb;
    };
    std::map<Key, SqQuadItem, decltype(less)> blocks(less);
    for (int64_t i = 0; i < n_q; ++i) {
        const int mx = trip[4 * i], my = trip[4 * i + 1], mz = trip[4 * i + 2];
        const Key key{mz / SQ_ZPT, mx, my / SQ_QCOLS};
        auto found = blocks.find(key);
        if (found == blocks.end()) {
            SqQuadItem it{};
            for (int c = 0; c < SQ_QCOLS; ++c) {
                it.i0[c] = (short)mx;
                it.i1[c] = (short)(key.yb * SQ_QCOLS + c);
                for (int z = 0; z < SQ_ZPT; ++z)
                    it.q[c][z] = -1;
            }
            for (int z = 0; z < SQ_ZPT; ++z)
                it.z[z] = (short)(key.zb * SQ_ZPT + z);
            found = blocks.emplace(key, it).first;
        }
        int &slot = found->second.q[my % SQ_QCOLS][mz % SQ_ZPT];
        if (slot >= 0)
            return false;   // the same wavevector twice
        slot = (int)i;
    }
    items.clear();
    for (auto &kv : blocks)
        items.push_back(kv.second);
    out = lat;
    for (int k = 0; k < 3; ++k) {
        out.mmin[k] = 0;
        out.R[k] = r;
    }
    used = double(n_q) / (double(items.size()) * SQ_QCOLS * SQ_ZPT);
    return !items.empty();
}

// Host: launch shape of the quad kernels for a lattice set — items, copies per item (threads of a
// block beyond the item count), table tile (two blocks per CU, two table sets per block: ~36 KB per
// set, a multiple of the copies) and dynamic LDS.  false: the set does not suit the quad form.
struct SqQuadShape {
    int n_items = 0, ipb = 0, n_sub = 1;   // items, items per block, copies of an item in its block
    int regular_stride = 0;   // > 0: every item is regular and the table rows are this many entries apart
    SqLattice lat{};
    size_t lds = 0;
    int blocks() const { return (n_items + ipb - 1) / ipb; }
};

inline bool sq_quad_plan(const std::vector<short> &trip, int64_t n_q, const SqLattice &base,
                         std::vector<SqQuadItem> &items, SqQuadShape &sh, bool allow_regular = true)
{
    // regular items (aligned blocks) where they use their accumulators about as well as the general items: the
    // regular form of the kernel does ~1.1 x the slots per second (measured on full 10^3 / 20^3 grids and sphere
    // octants, scripts/run/diag_sq_forms.py)
    SqLattice lat_use = base;
    bool regular = false;
    {
        std::vector<SqQuadItem> blk;
        SqLattice bl{};
        double used_blk = 0.0;
        if (allow_regular && sq_build_quad_blocks(trip, n_q, base, blk, bl, used_blk)) {
            const bool general = sq_build_quads(trip, n_q, base, items);
            const double used_gen = general ? double(n_q) / (double(items.size()) * SQ_QCOLS * SQ_ZPT) : 0.0;
            if (used_blk >= 0.45 && used_blk >= 0.92 * used_gen) {
                items.swap(blk);
                lat_use = bl;
                regular = true;
            } else if (!general) {
                return false;
            }
        } else if (!sq_build_quads(trip, n_q, base, items)) {
            return false;
        }
    }
    sh.n_items = (int)items.size();
    const int total_r = lat_use.R[0] + lat_use.R[1] + lat_use.R[2];
    const size_t fit = size_t(36) * 1024 / (size_t(16) * total_r);
    auto general_instead = [&]() {   // the padded tables of the regular items do not fit: the general items
        if (!regular)
            return false;
        SqQuadShape again;
        std::vector<SqQuadItem> gen;
        const bool ok = sq_quad_plan(trip, n_q, base, gen, again, false);
        if (ok) {
            items.swap(gen);
            sh = again;
        }
        return ok;
    };
    if (fit <= size_t(SQ_QUAD_PAD))
        return general_instead();
    const int tile_max = (int)std::min<size_t>(512, fit - SQ_QUAD_PAD);
    // the tile for `c` copies of an item: a multiple of c.  Regular items (sq_quad_frame_regular): the row stride is
    // a template argument, so the tile is one of the instantiated sizes
    auto tile_for = [&](int c) {
        if (regular) {
            for (int cand : {80, 64, 48, 32, 16})   // 3 * tile fill tasks <= one per thread (96, 128: measured slower)
                if (cand <= tile_max && cand % c == 0)
                    return cand;
            return 0;
        }
        const int t = tile_max / c * c;
        return t >= 16 ? t : 0;
    };
    // Items per block and copies per item.  A block's threads run in lockstep: per tile of the particles a thread
    // spends ~149 instructions on each of its tile / c particles (idle threads cost what busy ones do) and ~150 on
    // the tile itself (its share of the table fill, the pipeline's prologue, the barrier) — C3's counters: 5.37
    // instructions per 64 terms against 4.66 in the particle loop, 5 particles per tile.  Take the split of the
    // items over blocks that minimises blocks * (149 / c + 150 / tile): e.g. 95 items with tiles of 32 go to 3
    // blocks x 32 items x 8 copies instead of one block of 128 slots x 2 copies with 33 of them idle.
    double best = 0.0;
    sh.ipb = 0;
    for (int nb = 1; nb <= sh.n_items; ++nb) {
        const int ipb = (sh.n_items + nb - 1) / nb;
        if (ipb > SQ_QUAD_THREADS)
            continue;
        const int blocks = (sh.n_items + ipb - 1) / ipb;
        int c = SQ_QUAD_THREADS / ipb, tile = 0;
        while (c >= 1 && !(tile = tile_for(c)))
            --c;
        if (c >= 1) {
            const double cost = blocks * (149.0 / c + 150.0 / tile);
            if (!sh.ipb || cost < best * (1.0 - 1e-9)) {
                best = cost;
                sh.ipb = ipb;
                sh.n_sub = c;
                sh.lat = lat_use;
                sh.lat.tile = tile;
            }
        }
        if (ipb == 1)
            break;
    }
    if (!sh.ipb)
        return general_instead();
    sh.regular_stride = regular ? sh.lat.tile + SQ_QUAD_PAD : 0;
    sh.lds = std::max<size_t>(size_t(32) * (sh.lat.tile + SQ_QUAD_PAD) * total_r + 256, size_t(32) * 1024);
    return true;
}

// Host: the column items of a detected lattice set (trip: short[n_q][4]).  Items are ordered
// chunk-of-z major, column minor, so that the lanes of a wave hold the same m_z lists whenever
// the columns do.  Returns false when the set is too scattered for columns to pay off.
inline bool sq_build_columns(const std::vector<short> &trip, int64_t n_q, const SqLattice &lat,
                             std::vector<SqColumnItem> &items)
{
    struct Entry { int mx, my, mz; int q; };
    std::vector<Entry> e((size_t)n_q);
    for (int64_t i = 0; i < n_q; ++i)
        e[(size_t)i] = {trip[4 * i], trip[4 * i + 1], trip[4 * i + 2], (int)i};
    std::sort(e.begin(), e.end(), [](const Entry &a, const Entry &b) {
        if (a.mx != b.mx) return a.mx < b.mx;
        if (a.my != b.my) return a.my < b.my;
        if (a.mz != b.mz) return a.mz < b.mz;
        return a.q < b.q;
    });
    // columns -> chunks of SQ_ZPT
    std::vector<std::vector<SqColumnItem>> by_chunk;
    size_t i = 0;
    while (i < e.size()) {
        size_t j = i;
        while (j < e.size() && e[j].mx == e[i].mx && e[j].my == e[i].my)
            ++j;
        for (size_t c = i, ord = 0; c < j; c += SQ_ZPT, ++ord) {
            SqColumnItem it{};
            it.i0 = (short)(e[i].mx - lat.mmin[0]);
            it.i1 = (short)(e[i].my - lat.mmin[1]);
            it.nz = (short)std::min<size_t>(SQ_ZPT, j - c);
            for (int k = 0; k < it.nz; ++k) {
                it.z[k] = (short)(e[c + k].mz - lat.mmin[2]);
                it.q[k] = e[c + k].q;
            }
            if (by_chunk.size() <= ord)
                by_chunk.resize(ord + 1);
            by_chunk[ord].push_back(it);
        }
        i = j;
    }
    items.clear();
    for (const auto &v : by_chunk)
        items.insert(items.end(), v.begin(), v.end());
    // worthwhile when the items are at least half full on average
    return !items.empty() && n_q >= int64_t(items.size()) * (SQ_ZPT / 2);
}

// 6 waves/SIMD: 78 VGPRs without spills (unbounded the table-build code takes 120 = 4 waves)
__global__ __launch_bounds__(SQ_THREADS, 6) void sq_rho_lattice_kernel(
    const float *__restrict__ pos, int64_t n_atoms, const short4 *__restrict__ mtrip, int n_q,
    SqLattice lat, const int64_t *__restrict__ group_offsets, int n_groups, int n_split,
    double2 *__restrict__ rho)
{
    extern __shared__ double2 lat_tab[];
    const int tid = threadIdx.x;
    const int qb = blockIdx.x;
    const int g = blockIdx.y / n_split, sp = blockIdx.y % n_split;
    const int frame = blockIdx.z;
    const int A = lat.tile;
    double2 *tab[3] = {lat_tab, lat_tab + size_t(A) * lat.R[0],
                       lat_tab + size_t(A) * (lat.R[0] + lat.R[1])};

    int i0[SQ_QPT], i1[SQ_QPT], i2[SQ_QPT];
    double ac[SQ_QPT], as[SQ_QPT];
#pragma unroll
    for (int u = 0; u < SQ_QPT; ++u) {
        int qi = qb * SQ_QPB + u * SQ_THREADS + tid;
        short4 m = mtrip[min(qi, n_q - 1)];
        i0[u] = m.x - lat.mmin[0];
        i1[u] = m.y - lat.mmin[1];
        i2[u] = m.z - lat.mmin[2];
        ac[u] = 0.0;
        as[u] = 0.0;
    }
    const int64_t g_lo = group_offsets[g], g_hi = group_offsets[g + 1];
    const int64_t per = (g_hi - g_lo + n_split - 1) / n_split;
    const int64_t lo = g_lo + sp * per, hi = min(g_hi, lo + per);
    const float *P = pos + int64_t(frame) * n_atoms * 3;

    for (int64_t base = lo; base < hi; base += A) {
        const int cnt = (int)min<int64_t>(A, hi - base);
        __syncthreads();
        // one (particle, axis) per thread: E(1) by sincos, the rest by recurrence
        for (int t = tid; t < cnt * 3; t += SQ_THREADS) {
            const int a = t / 3, k = t - 3 * a;
            const double theta = lat.base[k] * (double)P[(base + a) * 3 + k];
            sq_lattice_fill_row(tab[k] + size_t(a) * lat.R[k], theta, lat.mmin[k], lat.R[k]);
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
                as[u] += fma(tr, ez.y, ti * ez.x);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < SQ_QPT; ++u) {
        int qi = qb * SQ_QPB + u * SQ_THREADS + tid;
        if (qi < n_q)
            rho[((int64_t(frame) * n_groups + g) * n_split + sp) * n_q + qi] =
                make_double2(ac[u], as[u]);
    }
}

}  // namespace
}  // namespace mdx_sq_dev
