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
 * Returns 0, or -2 for a non-orthorhombic box.
 */
int rdf_oracle_histogram(const float *pos1, long n1, const float *pos2, long n2,
                         const float *box6, const double *edges, int n_bins,
                         double r0, double r1, long e0, long e1,
                         long long *counts, int n_threads)
{
    float box[3], inv_box[3];
    if (box6) {
        if (box6[3] != 90.0f || box6[4] != 90.0f || box6[5] != 90.0f)
            return -2;
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
                double d = pair_distance(ri, pos2 + 3 * j, box6 ? box : NULL, inv_box);
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
