// mdx_internal.hpp — cross-translation-unit hooks (not part of the C-ABI).
#pragma once

#include <hip/hip_runtime.h>

#include "../../include/mdx.h"

extern "C" {
// RDF: reduce the replicas into one device vector / adopt an all-reduced total
int mdx_rdf_internal_total(mdx_rdf_t h, unsigned long long **d_total, hipStream_t *stream);
int mdx_rdf_internal_adopt_total(mdx_rdf_t h);
int mdx_rdf_internal_nbins(mdx_rdf_t h);
// S(q): the float64 accumulator [n_pairs][n_q]
int mdx_sq_internal_buffer(mdx_sq_t h, double **d_acc, int64_t *n, hipStream_t *stream);
// MSD: the power-spectrum / D_k accumulators and the summed trajectories
int mdx_msd_internal_buffers(mdx_msd_t h, double **d_a, int64_t *na, double **d_b, int64_t *nb,
                             hipStream_t *stream);
}

// trajectory ingest: the C++ object behind an mdx_traj_t (mdx_traj.hpp)
namespace mdx {
struct Trajectory;
}
mdx::Trajectory *mdx_traj_internal(mdx_traj_t h);
