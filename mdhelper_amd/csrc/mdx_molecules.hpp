// mdx_molecules.hpp — centre-of-mass stage shared by the RDF, S(q) and ISF engines
// (SURVEY.md §8f row 3): incoming rows are particles sorted molecule by molecule, the analysis
// kernels see one float32 point per molecule.  Restates `center_of_mass`
// (reference src/mdhelper/algorithm/molecule.py:300-306) as the analyses call it for
// groupings="residues" / "segments" (analysis/structure.py:753-756, :1484-1486).
#pragma once

#include "mdx_common.hpp"

#include <vector>

namespace mdx {
namespace {   // internal linkage: compiled into several translation units

// out[frame][g][k] = (float)(sum_a m_a x_a / M_g) over the particles a in [offsets[g], offsets[g+1])
// of the incoming order, accumulated in double in that order with separate multiply and add — the
// operations of the reference's host path (numpy: weights m * x, sequential sums, one division),
// so the float32 centres are the ones the reference feeds its histogram / Fourier sums.
__global__ __launch_bounds__(256) void molecule_com_kernel(const float *__restrict__ pos, int64_t n_atoms,
                                                           const int64_t *__restrict__ offsets,
                                                           const double *__restrict__ masses,
                                                           const double *__restrict__ total_mass,
                                                           int64_t n_groups, float *__restrict__ out)
{
    const int64_t i = int64_t(blockIdx.x) * 256 + threadIdx.x;   // (group, k)
    const int64_t frame = blockIdx.y;
    if (i >= n_groups * 3)
        return;
    const int64_t g = i / 3;
    const int k = int(i - 3 * g);
    const float *p = pos + frame * n_atoms * 3 + k;
    double acc = 0.0;
    for (int64_t a = offsets[g]; a < offsets[g + 1]; ++a)
        acc = __dadd_rn(acc, __dmul_rn(masses[a], (double)p[3 * a]));
    out[(frame * n_groups + g) * 3 + k] = (float)(acc / total_mass[g]);
}

}  // namespace

struct MoleculeStage {
    int64_t n_atoms = 0, n_groups = 0;   // n_groups == 0: plain particles, stage inactive
    DeviceBuffer d_offsets, d_masses, d_total, d_com;

    bool active() const { return n_groups > 0; }

    // offsets int64[n_groups + 1] (CSR over the incoming rows), masses float64[offsets[n_groups]];
    // n_groups <= 0 switches the stage off.  The caller has synchronised the consuming stream.
    int set(int64_t groups, const int64_t *offsets, const double *masses)
    {
        if (groups <= 0) {
            n_groups = n_atoms = 0;
            return MDX_OK;
        }
        MDX_REQUIRE(offsets && masses, "NULL argument");
        MDX_REQUIRE(offsets[0] == 0, "offsets must start at 0");
        const int64_t atoms = offsets[groups];
        std::vector<double> total((size_t)groups);
        for (int64_t g = 0; g < groups; ++g) {
            MDX_REQUIRE(offsets[g + 1] > offsets[g], "group %lld is empty", (long long)g);
            double m = 0.0;
            for (int64_t a = offsets[g]; a < offsets[g + 1]; ++a)
                m += masses[a];   // sequential, as numpy.bincount sums the weights
            MDX_REQUIRE(m > 0.0, "group %lld has no mass", (long long)g);
            total[(size_t)g] = m;
        }
        MDX_TRY(d_offsets.ensure(size_t(8) * (groups + 1)));
        MDX_TRY(d_masses.ensure(size_t(8) * atoms));
        MDX_TRY(d_total.ensure(size_t(8) * groups));
        MDX_HIP(hipMemcpy(d_offsets.ptr, offsets, size_t(8) * (groups + 1), hipMemcpyHostToDevice));
        MDX_HIP(hipMemcpy(d_masses.ptr, masses, size_t(8) * atoms, hipMemcpyHostToDevice));
        MDX_HIP(hipMemcpy(d_total.ptr, total.data(), size_t(8) * groups, hipMemcpyHostToDevice));
        n_groups = groups;
        n_atoms = atoms;
        return MDX_OK;
    }

    // src float32[n_frames][n_atoms][3] -> dst float32[n_frames][n_groups][3] (dst == nullptr: d_com)
    int run(hipStream_t stream, const float *src, int64_t n_frames, float *dst, const float **out)
    {
        if (!dst) {
            MDX_TRY(d_com.ensure(size_t(12) * n_groups * n_frames));
            dst = d_com.as<float>();
        }
        if (n_frames > 0)
            hipLaunchKernelGGL(molecule_com_kernel,
                               dim3((unsigned)ceil_div(n_groups * 3, 256), (unsigned)n_frames), dim3(256), 0,
                               stream, src, n_atoms, d_offsets.as<int64_t>(), d_masses.as<double>(),
                               d_total.as<double>(), n_groups, dst);
        if (out)
            *out = dst;
        return MDX_OK;
    }

    void release()
    {
        for (DeviceBuffer *b : {&d_offsets, &d_masses, &d_total, &d_com})
            b->release();
    }
    // destroy paths, after the handle's streams have been synchronised: the blocks go to the per-device cache
    void recycle()
    {
        for (DeviceBuffer *b : {&d_offsets, &d_masses, &d_total, &d_com})
            b->recycle();
    }
};

}  // namespace mdx
