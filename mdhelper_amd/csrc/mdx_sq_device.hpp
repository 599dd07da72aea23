// mdx_sq_device.hpp — device code shared by the structure-factor and ISF engines:
// fp64 sincos and the fused exp(i q.r) accumulation kernel (see mdx_sq.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
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
                ar[j] += fma(tr, ez.x, -ti * ez.y);
                ai[j] += fma(tr, ez.y, ti * ez.x);
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
