"""
oracle.fourier — CPU restatement of the structure-factor path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

Follows:

* ``fourier_sum_ref``   reference ``src/mdhelper/algorithm/accelerated.py:81-122``
                        (``F[q] = sum_j exp(i q . r_j)``; ``dot_1d_1d :12-43``,
                        ``delta_fourier_transform_1d_1d :45-79``)
* ``inner_ref``         ``accelerated.py:167-206`` (``q . r`` table)
* ``ssf_frame_ref``     ``src/mdhelper/analysis/structure.py:1481-1527``
                        (``_single_frame``: exp and trig forms, modes None/pair/partial)
* ``grid_wavevectors``  ``structure.py:1376-1381, 1404-1410`` (meshgrid, default 'xy' indexing)
* ``ssf_run_ref``       ``_prepare :1456-1479`` + ``_conclude :1529-1550``
"""

from __future__ import annotations

from itertools import combinations_with_replacement

import numpy as np


def fourier_sum_ref(qs: np.ndarray, rs: np.ndarray, q_chunk: int = 64) -> np.ndarray:
    """complex128[N_q] = sum over particles of exp(i q.r), float64 throughout."""
    qs = np.asarray(qs, dtype=np.float64).reshape(-1, 3)
    rs = np.asarray(rs, dtype=np.float64).reshape(-1, 3)
    out = np.empty(qs.shape[0], dtype=np.complex128)
    for lo in np.arange(0, qs.shape[0], q_chunk):
        q = qs[lo:lo + q_chunk]
        # a[0]*b[0] + a[1]*b[1] + a[2]*b[2]  (accelerated.py:43)
        phase = (q[:, None, 0] * rs[None, :, 0] + q[:, None, 1] * rs[None, :, 1]) \
            + q[:, None, 2] * rs[None, :, 2]
        out[lo:lo + q_chunk] = np.exp(1j * phase).sum(axis=1)
    return out


def inner_ref(qs, rs):
    qs = np.asarray(qs, dtype=np.float64).reshape(-1, 3)
    rs = np.asarray(rs, dtype=np.float64).reshape(-1, 3)
    return (qs[:, None, 0] * rs[None, :, 0] + qs[:, None, 1] * rs[None, :, 1]) \
        + qs[:, None, 2] * rs[None, :, 2]


def ssf_pairs(n_groups: int, mode):
    """``_prepare`` (structure.py:1459-1464)."""
    if mode == "partial":
        return tuple(combinations_with_replacement(range(n_groups), 2))
    if mode == "pair":
        return ((0, n_groups - 1),)
    return ((None, None),)


def ssf_frame_ref(wavevectors, positions, slices, pairs, mode, form="exp"):
    """One frame's un-normalised contribution, float64[n_pairs, N_q]."""
    out = np.zeros((len(pairs), len(wavevectors)))
    if form == "exp":
        if mode is None:
            rho = fourier_sum_ref(wavevectors, positions)
            out[0] = (rho * rho.conj()).real
        else:
            for i, (j, k) in enumerate(pairs):
                rho_j = fourier_sum_ref(wavevectors, positions[slices[j]])
                if j == k:
                    out[i] = (rho_j * rho_j.conj()).real
                else:
                    rho_k = fourier_sum_ref(wavevectors, positions[slices[k]])
                    out[i] = 2 * (rho_j * rho_k.conj()).real
    elif form == "trig":
        # accelerated.py:249-321 (Pythagorean identities on the q.r table)
        if mode is None:
            qr = inner_ref(wavevectors, positions)
            out[0] = np.cos(qr).sum(1) ** 2 + np.sin(qr).sum(1) ** 2
        else:
            for i, (j, k) in enumerate(pairs):
                qr_j = inner_ref(wavevectors, positions[slices[j]])
                if j == k:
                    out[i] = np.cos(qr_j).sum(1) ** 2 + np.sin(qr_j).sum(1) ** 2
                else:
                    qr_k = inner_ref(wavevectors, positions[slices[k]])
                    out[i] = 2 * (np.cos(qr_j).sum(1) * np.cos(qr_k).sum(1)
                                  + np.sin(qr_j).sum(1) * np.sin(qr_k).sum(1))
    else:
        raise ValueError("Invalid form.")
    return out


def grid_wavevectors(dimensions, n_points):
    """Cubic / non-cubic reciprocal grid, rows ordered as numpy.meshgrid's default 'xy'."""
    dimensions = np.asarray(dimensions, dtype=np.float64)
    if np.allclose(dimensions, dimensions[0]):
        grid = 2 * np.pi * np.arange(n_points) / dimensions[0]
        return np.stack(np.meshgrid(grid, grid, grid), -1).reshape(-1, 3)
    return np.stack(
        np.meshgrid(*[2 * np.pi * np.arange(n_points) / L for L in dimensions]), axis=-1
    ).reshape(-1, 3)


def ssf_run_ref(frames, group_sizes, wavevectors, *, mode=None, form="exp",
                sort=True, unique=True):
    """
    frames : float[F, N, 3] with the groups laid out consecutively
    (``self._positions[s]`` of structure.py:1484-1486).
    """
    wavevectors = np.asarray(wavevectors, dtype=np.float64)
    wavenumbers = np.linalg.norm(wavevectors, axis=1)
    slices, idx = [], 0
    for n in group_sizes:
        slices.append(slice(idx, idx + n))
        idx += n
    n_total = idx
    pairs = ssf_pairs(len(group_sizes), mode)
    ssf = np.zeros((len(pairs), len(wavenumbers)))
    positions = np.empty((n_total, 3))
    for f in np.arange(len(frames)):
        positions[:] = frames[f][:n_total]
        ssf += ssf_frame_ref(wavevectors, positions, slices, pairs, mode, form)
    ssf /= len(frames) * n_total
    out_q = np.unique(wavenumbers.round(11)) if unique else wavenumbers
    if unique:
        ssf = np.hstack([ssf[:, np.isclose(q, wavenumbers)].mean(axis=1, keepdims=True)
                         for q in out_q])
    if sort:
        order = np.argsort(out_q)
        out_q = out_q[order]
        ssf = ssf[:, order]
    return {"pairs": pairs, "wavenumbers": out_q, "ssf": ssf}


def isf_run_ref(frames, group_sizes, wavevectors, n_lags=None, *, mode=None, incoherent=False,
                sort=True, unique=True):
    """
    Intermediate scattering functions, restating
    ``IntermediateScatteringFunction._prepare/_single_frame/_conclude``
    (reference structure.py:1904-2127, form="exp"; the "trig" form computes the same numbers).

    frames : float[F, N, 3] with the groups laid out consecutively.
    Returns cisf[n_lags, n_pairs or 1, N_q'], iisf[n_lags, n_groups or 1, N_q'] (or None).
    """
    frames = np.asarray(frames, dtype=np.float64)
    n_frames = len(frames)
    n_lags = n_lags or n_frames
    wavevectors = np.asarray(wavevectors, dtype=np.float64)
    wavenumbers = np.linalg.norm(wavevectors, axis=1)
    slices, idx = [], 0
    for n in group_sizes:
        slices.append(slice(idx, idx + n))
        idx += n
    n_total = idx
    n_groups = len(group_sizes)
    pairs = ssf_pairs(n_groups, mode)
    n_q = len(wavevectors)
    cisf = np.zeros((n_lags, 1 if mode is None else len(pairs), n_q))
    iisf = np.zeros((n_lags, 1 if mode is None else n_groups, n_q)) if incoherent else None
    ring_pos = np.zeros((n_lags, n_total, 3))
    ring_rho = np.empty((n_lags, 1 if mode is None else n_groups, n_q), dtype=complex)
    for f in range(n_frames):
        cur = f % n_lags
        ring_pos[cur] = frames[f][:n_total]
        if mode is None:
            ring_rho[cur, 0] = fourier_sum_ref(wavevectors, ring_pos[cur])
        else:
            for g in range(n_groups):
                ring_rho[cur, g] = fourier_sum_ref(wavevectors, ring_pos[cur, slices[g]])
        for lag in range(min(n_lags, f + 1)):
            old = (f - lag) % n_lags
            if mode is None:
                cisf[lag, 0] += (ring_rho[old, 0] * ring_rho[cur, 0].conj()).real
                if incoherent:
                    iisf[lag, 0] += fourier_sum_ref(wavevectors, ring_pos[cur] - ring_pos[old]).real
            else:
                for i, (j, k) in enumerate(pairs):
                    if j == k:
                        cisf[lag, i] += (ring_rho[old, j] * ring_rho[cur, j].conj()).real
                        if incoherent:
                            iisf[lag, j] += fourier_sum_ref(
                                wavevectors, ring_pos[cur, slices[j]] - ring_pos[old, slices[j]]).real
                    else:
                        cisf[lag, i] += ((ring_rho[old, j] * ring_rho[cur, k].conj()).real
                                         + (ring_rho[old, k] * ring_rho[cur, j].conj()).real)
    norm = n_total * np.arange(n_frames, n_frames - n_lags, -1)[:, None, None]
    cisf /= norm
    if incoherent:
        iisf /= norm
    out_q = np.unique(wavenumbers.round(11)) if unique else wavenumbers
    if unique:
        cisf = np.stack([cisf[:, :, np.isclose(q, wavenumbers)].mean(axis=2) for q in out_q], axis=-1)
        if incoherent:
            iisf = np.stack([iisf[:, :, np.isclose(q, wavenumbers)].mean(axis=2) for q in out_q], axis=-1)
    if sort:
        order = np.argsort(out_q)
        out_q = out_q[order]
        cisf = cisf[:, :, order]
        if incoherent:
            iisf = iisf[:, :, order]
    return {"pairs": pairs, "wavenumbers": out_q, "cisf": cisf, "iisf": iisf}
