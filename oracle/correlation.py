"""
oracle.correlation — CPU restatement of the FFT / direct time-correlation path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

Follows reference ``src/mdhelper/algorithm/correlation.py``:

* ``correlation_fft_ref``   ``:17-226``  (zero-padded rfft, ``f f*``, irfft, lag normalisation)
* ``correlation_shift_ref`` ``:228-459`` (sliding-window definition)
* ``msd_fft_ref``           ``:461-668`` (``MSD_m = S_m - 2 A_m`` with the cumsum recurrence)
* ``msd_shift_ref``         ``:670-850`` (Einstein definition)

Pinned against the reference file itself (importable here: numpy + scipy only)
through ``tests/golden/correlation_*.npz`` and against the closed-form cases of
``tests/test_algorithm_correlation.py:438-561``.  Real input only; the axis
conventions (axis 0 or 1; optional leading block axis; optional entity axis;
optional trailing vector axis) are the reference's.
"""

from __future__ import annotations

import warnings

import numpy as np
from scipy import fft as _fft


def _check(arr1, arr2, axis, lo_dim, what):
    arr1 = np.asarray(arr1)
    if arr1.size == 0:
        raise ValueError(f"The {what} must not be empty.")
    if not lo_dim <= arr1.ndim <= 4:
        raise ValueError(f"The {what} have an unsupported dimensionality.")
    if arr2 is not None:
        arr2 = np.asarray(arr2)
        if arr1.shape != arr2.shape:
            raise ValueError(f"The {what} must have the same dimensions.")
    if axis is None:
        if arr1.ndim == 4:
            axis = 1
        else:
            axis = 0
            if arr1.ndim > lo_dim:
                warnings.warn("Ambiguous time axis; using the first axis.")
    elif axis not in (0, 1):
        raise ValueError("The time axis must be the first or second axis.")
    return arr1, arr2, axis


def _lag_weights(n_t, ndim_after, axis):
    """N_t, N_t-1, ..., 1 shaped to broadcast along ``axis``."""
    shape = [1] * ndim_after
    shape[axis] = n_t
    return np.arange(n_t, 0, -1, dtype=np.float64).reshape(shape)


def correlation_fft_ref(arr1, arr2=None, axis=None, *, average=False, double=False,
                        vector=False):
    arr1, arr2, axis = _check(arr1, arr2, axis, 1, "arrays")
    ndim = arr1.ndim
    n_t = arr1.shape[axis]
    n_fft = 2 * _fft.next_fast_len(n_t, real=True)
    take = [slice(None)] * ndim
    take[axis] = slice(0, n_t)
    take = tuple(take)
    if arr2 is None:
        f = _fft.rfft(arr1, n=n_fft, axis=axis)
        corr = _fft.irfft(f * f.conj(), axis=axis)
        corr = (double + 1) * corr[take]
    else:
        f1 = _fft.rfft(arr1, n=n_fft, axis=axis)
        f2 = _fft.rfft(arr2, n=n_fft, axis=axis)
        f = f1.conj() * f2
        if double:
            corr = _fft.irfft(f + f1 * f2.conj(), axis=axis)[take]
        else:
            corr = _fft.irfft(f, axis=axis)
    if vector:
        corr = corr.sum(axis=-1)
    nd = corr.ndim
    if corr.shape[axis] == n_t:
        corr = corr / _lag_weights(n_t, nd, axis)
    else:
        # full CCF: positive lags 0..N_t-1 at the front, negative lags at the back
        pos = [slice(None)] * nd
        pos[axis] = slice(0, n_t)
        neg = [slice(None)] * nd
        neg[axis] = slice(corr.shape[axis] - (n_t - 1), None)
        w_pos = _lag_weights(n_t, nd, axis)
        shape = [1] * nd
        shape[axis] = n_t - 1
        w_neg = np.arange(1, n_t, dtype=np.float64).reshape(shape)
        corr = np.concatenate((corr[tuple(neg)] / w_neg, corr[tuple(pos)] / w_pos), axis=axis)
    if average:
        axis_avg = ndim - vector - 1
        if axis != axis_avg:
            return corr.mean(axis=axis_avg)
    return corr


def correlation_shift_ref(arr1, arr2=None, axis=None, *, average=False, double=False,
                          vector=False):
    arr1, arr2, axis = _check(arr1, arr2, axis, 1, "arrays")
    ndim = arr1.ndim
    n_t = arr1.shape[axis]
    a = np.moveaxis(arr1, axis, 0).astype(np.float64)
    b = a if arr2 is None else np.moveaxis(arr2, axis, 0).astype(np.float64)

    def window(lag):
        # sum_k a[k] * b[k + lag] over the overlap, optionally summed over the vector axis
        if lag >= 0:
            prod = a[:n_t - lag] * b[lag:]
        else:
            prod = a[-lag:] * b[:n_t + lag]
        out = prod.sum(axis=0)
        return out.sum(axis=-1) if vector else out

    if arr2 is None:
        corr = np.stack([window(m) for m in range(n_t)])
        if double:
            corr = corr * 2
        lags = np.arange(n_t)
    elif double:
        corr = np.stack([window(m) + window(-m) for m in range(n_t)])
        lags = np.arange(n_t)
    else:
        lags = np.arange(-(n_t - 1), n_t)
        corr = np.stack([window(m) for m in lags])
    weights = (n_t - np.abs(lags)).astype(np.float64)
    corr = corr / weights.reshape((-1,) + (1,) * (corr.ndim - 1))
    corr = np.moveaxis(corr, 0, axis)
    if average:
        axis_avg = ndim - 1 - vector
        if axis != axis_avg:
            return corr.mean(axis=axis_avg)
    return corr


def msd_fft_ref(pos1, pos2=None, axis=None, *, average=True):
    pos1, pos2, axis = _check(pos1, pos2, axis, 2, "position arrays")
    ndim = pos1.ndim
    n_t = pos1.shape[axis]
    s2 = correlation_fft_ref(pos1, pos2, axis, average=False, double=True, vector=True)
    d = (pos1 * (pos1 if pos2 is None else pos2)).sum(axis=-1)
    has_entities = (ndim - axis == 3)
    if has_entities and average:
        s2 = s2.mean(axis=ndim - 2)
        d = d.mean(axis=ndim - 2)
    # S_m (N_t - m) = 2 sum_k D_k - sum_{k=1..m} (D_{k-1} + D_{N_t-k})
    dm = np.moveaxis(d, axis, 0)
    tail = dm[:n_t - 1] + dm[:0:-1]
    run = np.concatenate((np.zeros((1,) + dm.shape[1:]), np.cumsum(tail, axis=0)), axis=0)
    ssum = 2 * dm.sum(axis=0)[None] - run
    ssum = ssum / np.arange(n_t, 0, -1, dtype=np.float64).reshape((-1,) + (1,) * (dm.ndim - 1))
    return np.moveaxis(ssum, 0, axis) - s2


def msd_shift_ref(pos1, pos2=None, axis=None, *, average=True):
    pos1, pos2, axis = _check(pos1, pos2, axis, 2, "position arrays")
    ndim = pos1.ndim
    n_t = pos1.shape[axis]
    a = np.moveaxis(pos1, axis, 0).astype(np.float64)
    b = a if pos2 is None else np.moveaxis(pos2, axis, 0).astype(np.float64)
    rows = []
    for m in range(n_t):
        da = a[m:] - a[:n_t - m]
        db = b[m:] - b[:n_t - m]
        rows.append((da * db).sum(axis=-1).mean(axis=0))
    disp = np.moveaxis(np.stack(rows), 0, axis)
    if ndim - axis == 3 and average:
        disp = disp.mean(axis=ndim - 2)
    return disp
