"""
Function-level drop-ins for ``mdhelper.algorithm.accelerated`` (reference
src/mdhelper/algorithm/accelerated.py:12-627): the reference's Numba micro-kernels, by name, on the MI355X.

The analysis classes do not call these — ``StructureFactor`` / ``IntermediateScatteringFunction`` run fused
kernels that never materialise ``q . r`` (one kernel family serves ``form="exp"`` and ``form="trig"``) — they exist
so that user code written against the reference's module keeps working, computed on the device through
``libmdx.so`` (``mdx_fourier_sum``, ``mdx_inner``, ``mdx_trig_rowsums``).  Arrays are float64 as in the reference;
the ``*_parallel_*`` names are the same device kernels (the reference's ``prange`` variants).  The scalar helpers
``dot_1d_1d`` / ``delta_fourier_transform_1d_1d`` are three multiplications: host arithmetic.

There is no CPU fallback: without a HIP device every array function raises ``RuntimeError``.
"""

from __future__ import annotations

import numpy as np

from .. import _core


def dot_1d_1d(a, b) -> float:
    """``a[0] b[0] + a[1] b[1] + a[2] b[2]`` (reference :12-43)."""
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def delta_fourier_transform_1d_1d(q, r) -> complex:
    """``exp(i q . r)`` for one wavevector and one position (reference :45-79)."""
    return np.exp(1j * dot_1d_1d(q, r))


def delta_fourier_transform_sum_2d_2d(qs, rs) -> np.ndarray:
    """``F[i] = sum_j exp(i q_i . r_j)``, complex128[N_q] (reference :81-122)."""
    return _core.fourier_sum_device(qs, rs)


delta_fourier_transform_sum_parallel_2d_2d = delta_fourier_transform_sum_2d_2d     # reference :124-165


def inner_2d_2d(qs, rs) -> np.ndarray:
    """``s[i, j] = q_i . r_j``, float64[N_q, N_r] (reference :167-206)."""
    return _core.inner_device(qs, rs)


inner_parallel_2d_2d = inner_2d_2d                                                 # reference :208-247


def cosine_sum_2d(xs) -> np.ndarray:
    """Row-wise ``sum_j cos(xs[i, j])`` (reference :353-383)."""
    return _core.trig_rowsums_device(xs, sin=False)[0]


def sine_sum_2d(xs) -> np.ndarray:
    """Row-wise ``sum_j sin(xs[i, j])`` (reference :506-536)."""
    return _core.trig_rowsums_device(xs, cos=False)[1]


cosine_sum_parallel_2d = cosine_sum_2d                                             # reference :385-415
sine_sum_parallel_2d = sine_sum_2d                                                 # reference :538-568


def cosine_sum_1d(x) -> float:
    """``sum_j cos(x[j])`` (reference :323-351)."""
    return float(cosine_sum_2d(np.asarray(x, dtype=np.float64)[None])[0])


def sine_sum_1d(x) -> float:
    """``sum_j sin(x[j])`` (reference :476-504)."""
    return float(sine_sum_2d(np.asarray(x, dtype=np.float64)[None])[0])


def cosine_sum_inplace_2d(xs, s) -> None:
    """``s[i] = sum_j cos(xs[i, j])`` into the caller's array (reference :417-444)."""
    assert s.shape[0] == np.shape(xs)[0]
    s[:] = cosine_sum_2d(xs)


def sine_sum_inplace_2d(xs, s) -> None:
    """``s[i] = sum_j sin(xs[i, j])`` into the caller's array (reference :570-597)."""
    assert s.shape[0] == np.shape(xs)[0]
    s[:] = sine_sum_2d(xs)


cosine_sum_inplace_parallel_2d = cosine_sum_inplace_2d                             # reference :446-474
sine_sum_inplace_parallel_2d = sine_sum_inplace_2d                                 # reference :599-627


def pythagorean_trigonometric_identity_1d(r) -> float:
    """``(sum cos r_i)^2 + (sum sin r_i)^2`` (reference :249-279)."""
    c, s = _core.trig_rowsums_device(np.asarray(r, dtype=np.float64)[None])
    return float(c[0] ** 2 + s[0] ** 2)


def pythagorean_trigonometric_identity_1d_1d(r, s) -> float:
    """``2 (sum cos r_i sum cos s_j + sum sin r_i sum sin s_j)`` (reference :281-321)."""
    c1, s1 = _core.trig_rowsums_device(np.asarray(r, dtype=np.float64)[None])
    c2, s2 = _core.trig_rowsums_device(np.asarray(s, dtype=np.float64)[None])
    return float(2 * (c1[0] * c2[0] + s1[0] * s2[0]))
