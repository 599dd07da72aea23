/*
 * oracle/c/rdf_oracle.c — plain-C restatement of the reference's radial
 * histogram, used as the fast CPU checker and as bench.py's cpu_baseline.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the product library
 * (mdhelper_amd/csrc) never links or loads this file.
 *
 * Follows reference src/mdhelper/analysis/structure.py:92-104:
 *   :93-96   capped_distance(pos1, pos2, range[1], range[0] - eps, box=dims)
 *            -> all ordered pairs with  min < d <= max; the distance
 *            arithmetic is MDAnalysis' brute-force orthorhombic path
 *            (third party, absent, unpinned: "parity unpinned" for this
 *            step, contract in SURVEY.md §8 a-1), restated in pair_distance()
 *   :100-102 exclusion  i / e0 != j / e1
 *   :104     numpy.histogram(dist, bins=n_bins, range=range) — uniform-bin
 *            fast path of numpy/lib/_histograms_impl.py:841-877 restated in
 *            numpy_bin().
 *
 * Build:  make -C oracle/c      (gcc -O2 -ffp-contract=off -fopenmp)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* MDAnalysis calc_distances.h: minimum_image() + squared norm + sqrt. */
static inline double pair_distance(const float *ri, const float *rj,
                                   const float *box, const float *inv_box)
{
    double dx[3];
    for (int k = 0; k < 3; ++k) {
        float d32 = rj[k] - ri[k];          /* float32 subtract */
        dx[k] = (double)d32;
        if (box) {
            double s = (double)inv_box[k] * dx[k];
            dx[k] = (double)box[k] * (s - round(s));
        }
    }
    double rsq = (dx[0] * dx[0] + dx[1] * dx[1]) + dx[2] * dx[2];
    return sqrt(rsq);
}

/*
 * Triclinic cells (angles != 90): MDAnalysis' brute-force triclinic path, restated from its
 * published algorithm (MDAnalysis/lib/src/calc_distances.h: _calc_distance_array_triclinic =
 * _triclinic_pbc on both sets + minimum_image_triclinic per pair; third party, absent here,
 * version unpinned: "parity unpinned" like the orthorhombic step).  Contract:
 *   1. box matrix B (rows a, b, c; lower triangular) from (lx, ly, lz, alpha, beta, gamma) as
 *      MDAnalysis.lib.mdamath.triclinic_vectors computes it, float64 arithmetic, entries stored
 *      as float32, exact zeros for right angles;
 *   2. both coordinate sets are moved into the central cell, c then b then a axis, in double:
 *      s = floor(r[k] / B[k][k]); r -= s * B[k]; the result is stored as float32;
 *   3. dx = (double)(conf[j] - ref[i])  (float32 subtract of the wrapped coordinates);
 *   4. the 27 images dx + ix a + iy b + iz c, ix, iy, iz in {-1, 0, 1} (ix outermost), are
 *      scanned in double and the first one with the strictly smallest squared length wins;
 *      rsq = (x*x + y*y) + z*z without contraction.
 */
void rdf_oracle_triclinic_vectors(const float *box6, float *B /* [9] row-major */)
{
    const double lx = box6[0], ly = box6[1], lz = box6[2];
    const double deg = 3.14159265358979323846 / 180.0;
    const double ca = box6[3] == 90.0f ? 0.0 : cos((double)box6[3] * deg);
    const double cb = box6[4] == 90.0f ? 0.0 : cos((double)box6[4] * deg);
    const double cg = box6[5] == 90.0f ? 0.0 : cos((double)box6[5] * deg);
    const double sg = box6[5] == 90.0f ? 1.0 : sin((double)box6[5] * deg);
    memset(B, 0, 9 * sizeof(float));
    B[0] = (float)lx;
    B[3] = (float)(ly * cg);
    B[4] = (float)(ly * sg);
    const double cx = lz * cb;
    const double cy = lz * (ca - cb * cg) / sg;
    B[6] = (float)cx;
    B[7] = (float)cy;
    B[8] = (float)sqrt(lz * lz - cx * cx - cy * cy);
}

void rdf_oracle_triclinic_wrap(const float *pos, long n, const float *B, float *out)
{
    for (long i = 0; i < n; ++i) {
        double r[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
        for (int k = 2; k >= 0; --k) {
            double s = floor(r[k] / (double)B[4 * k]);
            for (int c = 0; c <= k; ++c)
                r[c] -= s * (double)B[3 * k + c];
        }
        out[3 * i] = (float)r[0];
        out[3 * i + 1] = (float)r[1];
        out[3 * i + 2] = (float)r[2];
    }
}

static inline double pair_distance_triclinic(const float *ri, const float *rj, const float *B)
{
    double dx[3];
    for (int k = 0; k < 3; ++k) {
        float d32 = rj[k] - ri[k];
        dx[k] = (double)d32;
    }
    double best = 1.0e300;
    for (int ix = -1; ix < 2; ++ix) {
        double rx = dx[0] + (double)B[0] * ix;
        for (int iy = -1; iy < 2; ++iy) {
            double ry0 = rx + (double)B[3] * iy;
            double ry1 = dx[1] + (double)B[4] * iy;
            for (int iz = -1; iz < 2; ++iz) {
                double rz0 = ry0 + (double)B[6] * iz;
                double rz1 = ry1 + (double)B[7] * iz;
                double rz2 = dx[2] + (double)B[8] * iz;
                double dsq = (rz0 * rz0 + rz1 * rz1) + rz2 * rz2;
                if (dsq < best)
                    best = dsq;
            }
        }
    }
    return sqrt(best);
}

/* numpy.histogram uniform-bin index; returns -1 when d is outside range. */
static inline int numpy_bin(double d, const double *edges, int n_bins,
                            double first_edge, double last_edge)
{
    if (!(d >= first_edge) || !(d <= last_edge))
        return -1;
    double norm_denom = last_edge - first_edge;
    double f = ((d - first_edge) / norm_denom) * (double)n_bins;
    long idx = (long)f;
    if (idx == n_bins)
        idx -= 1;
    if (d < edges[idx])
        idx -= 1;
    if (d >= edges[idx + 1] && idx != n_bins - 1)
        idx += 1;
    return (int)idx;
}

/*
 * counts[n_bins] += histogram of one frame.
 * box6 == NULL -> no periodic boundaries.  e0 == 0 -> no exclusion.
 * Returns 0 (angles != 90 select the triclinic contract above).
 */
int rdf_oracle_histogram(const float *pos1, long n1, const float *pos2, long n2,
                         const float *box6, const double *edges, int n_bins,
                         double r0, double r1, long e0, long e1,
                         long long *counts, int n_threads)
{
    float box[3], inv_box[3];
    float B[9];
    float *w1 = NULL, *w2 = NULL;
    const int tri = box6 && (box6[3] != 90.0f || box6[4] != 90.0f || box6[5] != 90.0f);
    if (tri) {
        rdf_oracle_triclinic_vectors(box6, B);
        w1 = (float *)malloc(sizeof(float) * 3 * (size_t)n1);
        w2 = (float *)malloc(sizeof(float) * 3 * (size_t)n2);
        if (!w1 || !w2)
            return -3;
        rdf_oracle_triclinic_wrap(pos1, n1, B, w1);
        rdf_oracle_triclinic_wrap(pos2, n2, B, w2);
        pos1 = w1;
        pos2 = w2;
    } else if (box6) {
        for (int k = 0; k < 3; ++k) {
            box[k] = box6[k];
            inv_box[k] = (float)(1.0 / box6[k]);
        }
    }
    const double max_cut = r1;
    const double min_cut = r0 - 2.220446049250313e-16;
    const double first_edge = r0, last_edge = r1;
#ifdef _OPENMP
    if (n_threads > 0)
        omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        long long *priv = (long long *)calloc((size_t)n_bins, sizeof(long long));
#pragma omp for schedule(dynamic, 16)
        for (long i = 0; i < n1; ++i) {
            const float *ri = pos1 + 3 * i;
            for (long j = 0; j < n2; ++j) {
                double d = tri ? pair_distance_triclinic(ri, pos2 + 3 * j, B)
                               : pair_distance(ri, pos2 + 3 * j, box6 ? box : NULL, inv_box);
                if (!(d <= max_cut && d > min_cut))
                    continue;
                if (e0 > 0 && (i / e0) == (j / e1))
                    continue;
                int b = numpy_bin(d, edges, n_bins, first_edge, last_edge);
                if (b >= 0)
                    priv[b] += 1;
            }
        }
#pragma omp critical
        for (int b = 0; b < n_bins; ++b)
            counts[b] += priv[b];
        free(priv);
    }
    free(w1);
    free(w2);
    return 0;
}

int rdf_oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
