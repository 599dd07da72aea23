"""
oracle.polymer — CPU restatement of the end-to-end vector ACF analysis.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

Follows reference ``src/mdhelper/analysis/polymer.py`` frame by frame:

* ``end_to_end_run_ref``         ``EndToEndVector._prepare`` :677-737, ``_single_frame`` :739-763
  (explicit ``n_chains`` / ``n_monomers`` branch: ``positions.reshape(M, N_p, -1, 3)[:, (0, -1)]``,
  first atom for ``"atoms"``, centre of mass for ``"residues"``; ``unwrap`` with the image
  flags of ``algorithm/topology.py:366-376``; ``np.diff`` of the two ends) and ``_conclude``
  :765-781 (unit vectors, ``correlation_fft(..., average=True, vector=True)``).
* ``unwrap_edge_chain_ref``      ``algorithm/topology.py:385-529`` for linear chains: atom by
  atom along the chain with the minimum image of the bond vector, then the chain is moved
  so that its centre of mass lies in the cell.
* ``relaxation_time_ref``        ``calculate_relaxation_time`` :59-108.

Pinning: the ACF itself goes through ``oracle.correlation.correlation_fft_ref``, which is
pinned against the reference's own file (``tests/golden/correlation_ref.npz``).  The reference
holds no test or fixture for ``EndToEndVector`` (``tests/test_analysis_polymer.py`` covers
``Gyradius`` only) and the class cannot be imported here (MDAnalysis absent), so the
class-level restatement is **parity unpinned**; it is checked against closed forms instead
(rigid rotors: ACF = cos(omega t); freely rotating dumbbells: exponential decay).
"""

from __future__ import annotations

import numpy as np
from scipy import optimize, special

from .correlation import correlation_fft_ref, correlation_shift_ref


def unwrap_edge_chain_ref(positions, n_chains, dimensions, masses):
    """positions float[M * n, 3] of M linear chains of n atoms -> whole chains, COM in the cell."""
    pos = np.array(positions, dtype=float)
    L = np.asarray(dimensions[:3], dtype=float)
    n = len(pos) // n_chains
    for c in range(n_chains):
        for j in range(1, n):
            i = c * n + j
            d = pos[i] - pos[i - 1]
            pos[i] = pos[i - 1] + (d - L * np.round(d / L))
        sl = slice(c * n, (c + 1) * n)
        m = np.asarray(masses[sl], dtype=float)
        com = (m[:, None] * pos[sl]).sum(axis=0) / m.sum()
        wrapped = com.copy()
        outside = (wrapped < 0) | (wrapped > L)
        wrapped[outside] -= (np.floor(wrapped / L) * L)[outside]
        pos[sl] += wrapped - com
    return pos


def _ends(frame_positions, M, N_p, grouping, masses):
    ends = np.asarray(frame_positions, dtype=float).reshape(M, N_p, -1, 3)[:, (0, -1)]
    if grouping == "atoms":
        return ends[:, :, 0]
    m = np.asarray(masses, dtype=float).reshape(M, N_p, -1)[:, (0, -1)]
    return np.einsum("cea,cead->ced", m, ends) / m.sum(axis=-1)[..., None]


def end_to_end_run_ref(positions, group_indices, n_chains, n_monomers, groupings, *, masses=None,
                       dimensions=None, n_blocks=1, unwrap=False, fft=True):
    """
    positions float[T, N, 3]; group_indices: list of index arrays (one per group).
    Returns ``acf[n_groups, n_blocks, T // n_blocks]`` and ``e2e[T, sum(n_chains), 3]``.
    """
    positions = np.asarray(positions)
    T = positions.shape[0]
    masses = np.ones(positions.shape[1]) if masses is None else np.asarray(masses, dtype=float)
    n_groups = len(group_indices)
    slices, index = [], 0
    for M in n_chains:
        slices.append(slice(index, index + M))
        index += M
    e2e = np.empty((T, index, 3))
    if unwrap:
        L = np.asarray(dimensions[:3], dtype=float)
        old = np.empty((index, 2, 3))
        for idx, gr, s, M, N_p in zip(group_indices, groupings, slices, n_chains, n_monomers):
            whole = unwrap_edge_chain_ref(positions[0][idx], M, L, masses[idx])
            old[s] = _ends(whole, M, N_p, gr, masses[idx])
        images = np.zeros((index, 2, 3), dtype=int)
        thresholds = L / 2
    for f in range(T):
        for idx, gr, s, M, N_p in zip(group_indices, groupings, slices, n_chains, n_monomers):
            ends = _ends(positions[f][idx], M, N_p, gr, masses[idx])
            if unwrap:
                dpos = ends - old[s]
                mask = np.abs(dpos) >= thresholds
                img = images[s]
                img[mask] -= np.sign(dpos[mask]).astype(int)
                old[s] = ends
                ends = ends + img * L
            e2e[f, s] = np.diff(ends, axis=1)[:, 0]
    Tb = T // n_blocks
    used = e2e[:n_blocks * Tb]
    acf = np.empty((n_groups, n_blocks, Tb))
    corr = correlation_fft_ref if fft else correlation_shift_ref
    for i, (s, M) in enumerate(zip(slices, n_chains)):
        u = used[:, s] / np.linalg.norm(used[:, s], axis=-1, keepdims=True)
        acf[i] = corr(u.reshape(n_blocks, -1, M, 3), average=True, vector=True)
    return acf, e2e


def relaxation_time_ref(time, acf):
    def stretched(x, alpha, beta):
        return np.exp(-(x / alpha) ** beta)
    tau, beta = optimize.curve_fit(stretched, time / time[1], acf, bounds=(0, np.inf))[0]
    return tau * time[1] * special.gamma(1 + beta ** -1)
