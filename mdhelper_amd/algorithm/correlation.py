"""
Time correlation functions and mean squared displacements (operator surface of
``mdhelper.algorithm.correlation``).

Mirrors reference ``src/mdhelper/algorithm/correlation.py``:

* ``correlation_fft``   :17-226   — GPU: batched rocFFT R2C / spectrum product / C2R
                                    (``mdx_correlate``), lag normalisation on the host
* ``correlation_shift`` :228-459  — direct O(T^2) definition, host NumPy (the
                                    reference's own cross-check, not a GPU target)
* ``msd_fft``           :461-668  — ``MSD_m = S_m - 2 A_m`` on top of ``correlation_fft``
* ``msd_shift``         :670-850  — Einstein definition, host NumPy

Argument names, axis conventions (time along axis 0 or 1; optional leading block
axis, entity axis, trailing vector axis), keyword flags, warnings and error
behaviour follow the reference.  ``Onsager`` does not go through ``msd_fft`` for
its per-particle term: it feeds particles to the MSD engine (``mdx_msd_*``), which
keeps one power spectrum per group instead of one inverse transform per particle.
"""

from __future__ import annotations

import warnings

import numpy as np

from .. import _core


def _validate(arr1, arr2, axis, min_dim, noun, warn_dim):
    arr1 = np.asarray(arr1)
    if arr1.size == 0:
        raise ValueError(f"The {noun} must not be empty.")
    ndim = arr1.ndim
    if not min_dim <= ndim <= 4:
        names = {1: "one-, two-, three-, or four-", 2: "two-, three-, or four-"}[min_dim]
        raise ValueError(f"The {noun} must be {names}dimensional.")
    if arr2 is not None:
        arr2 = np.asarray(arr2)
        if arr1.shape != arr2.shape:
            raise ValueError(f"The {noun} must have the same dimensions.")
    if axis is None:
        if ndim == 4:
            axis = 1
        else:
            axis = 0
            if warn_dim(ndim):
                warnings.warn("The axis along which to evaluate the correlation was not "
                              "specified and is ambiguous for a multidimensional array. As such, "
                              "it has been set to the first axis by default.")
    elif axis not in {0, 1}:
        raise ValueError("The correlation can only be evaluated along the first or second axis.")
    return arr1, arr2, axis


def _series(arr, axis):
    """Time axis last, everything else flattened: float64[n_series, n_t]."""
    moved = np.moveaxis(arr, axis, -1)
    dtype = np.complex128 if np.iscomplexobj(moved) else np.float64
    return np.ascontiguousarray(moved.reshape(-1, moved.shape[-1]), dtype=dtype), moved.shape


def _unseries(flat, shape, axis):
    return np.moveaxis(flat.reshape(shape[:-1] + (flat.shape[-1],)), -1, axis)


def _raw_correlation(a, b, negative):
    """sum_k conj(a[k]) b[k+m] for real or complex series via real GPU correlations."""
    if not (np.iscomplexobj(a) or (b is not None and np.iscomplexobj(b))):
        return _core.correlate_device(a, b, negative=negative)
    bb = a if b is None else b
    ar, ai = np.ascontiguousarray(a.real), np.ascontiguousarray(a.imag)
    br, bi = np.ascontiguousarray(bb.real), np.ascontiguousarray(bb.imag)
    parts = [_core.correlate_device(x, y, negative=negative)
             for x, y in ((ar, br), (ai, bi), (ar, bi), (ai, br))]
    if negative:
        pos = (parts[0][0] + parts[1][0]) + 1j * (parts[2][0] - parts[3][0])
        neg = (parts[0][1] + parts[1][1]) + 1j * (parts[2][1] - parts[3][1])
        return pos, neg
    return (parts[0] + parts[1]) + 1j * (parts[2] - parts[3])


def correlation_fft(arr1, arr2=None, axis: int = None, *, average: bool = False,
                    double: bool = False, vector: bool = False) -> np.ndarray:
    r"""
    Autocorrelation (ACF) or cross-correlation (CCF) of time series by the Fast
    Correlation Algorithm (Wiener–Khinchin), evaluated on the GPU.

    .. math:: A(\tau)=\mathrm{FFT}^{-1}\left[\hat r(\xi)\hat r^*(\xi)\right],\qquad
              \hat r=\mathrm{FFT}(r\ \text{zero-padded to}\ 2\,\mathrm{next\_fast\_len}(N_t))

    Parameters
    ----------
    arr1, arr2 : array-like, 1- to 4-D (``arr2=None`` → ACF)
        ``(N_t,)``, ``(N_t, N)``, ``(N_b, N_t)``, ``(N_t, N, d)``, ``(N_b, N_t, N)``,
        ``(N_b, N_t, N, d)`` … with the time axis given by ``axis``.
    axis : {0, 1}, optional
    average : bool — average over the entity axis
    double : bool — double the ACF / overlap positive and negative CCF lags
    vector : bool — the last axis holds vector components (summed)

    Returns
    -------
    corr : numpy.ndarray — normalised by :math:`N_t-|m|`; a non-``double`` CCF has
        :math:`2N_t-1` lags ordered negative → positive.
    """
    arr1, arr2, axis = _validate(arr1, arr2, axis, 1, "arrays", lambda nd: nd > 1)
    ndim = arr1.ndim
    n_t = arr1.shape[axis]
    a, shape = _series(arr1, axis)
    if arr2 is None:
        corr = _unseries(_raw_correlation(a, None, False), shape, axis)
        corr = (double + 1) * corr
    else:
        b, _ = _series(arr2, axis)
        pos, neg = _raw_correlation(a, b, True)
        if double:
            corr = _unseries(pos + neg, shape, axis)
        else:
            full = np.concatenate((neg[:, :0:-1], pos), axis=1)
            corr = _unseries(full, shape, axis)
    if vector:
        corr = corr.sum(axis=-1)
    lags = corr.shape[axis]
    weights = (n_t - np.abs(np.arange(lags) - (lags - n_t))).astype(np.float64)
    wshape = [1] * corr.ndim
    wshape[axis] = lags
    corr = corr / weights.reshape(wshape)
    if average:
        axis_avg = ndim - vector - 1
        if axis != axis_avg:
            return corr.mean(axis=axis_avg)
    return corr


def correlation_shift(arr1, arr2=None, axis: int = None, *, average: bool = False,
                      double: bool = False, vector: bool = False) -> np.ndarray:
    r"""
    ACF / CCF straight from the definition with sliding windows along the time
    axis, :math:`C(m)=\frac{1}{N_t-|m|}\sum_k r_1(k)\,r_2(k+m)` — O(:math:`N_t^2`),
    host NumPy; same arguments and output layout as :func:`correlation_fft`.
    """
    arr1, arr2, axis = _validate(arr1, arr2, axis, 1, "arrays", lambda nd: nd > 1)
    ndim = arr1.ndim
    n_t = arr1.shape[axis]
    a = np.moveaxis(np.asarray(arr1, dtype=np.result_type(arr1, np.float64)), axis, 0)
    b = a if arr2 is None else np.moveaxis(
        np.asarray(arr2, dtype=np.result_type(arr2, np.float64)), axis, 0)

    def window(lag):
        prod = (np.conj(a[:n_t - lag]) * b[lag:]) if lag >= 0 else (np.conj(a[-lag:]) * b[:n_t + lag])
        out = prod.sum(axis=0)
        return out.sum(axis=-1) if vector else out

    if arr2 is None:
        lags = np.arange(n_t)
        corr = np.stack([window(m) for m in lags]) * (double + 1)
    elif double:
        lags = np.arange(n_t)
        corr = np.stack([window(m) + window(-m) for m in lags])
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


def _msd_from_parts(d, s2, axis, n_t):
    """``S_m - 2 A_m`` given D_k = r1.r2 and the doubled, normalised correlation s2."""
    dm = np.moveaxis(d, axis, 0)
    tail = dm[:n_t - 1] + dm[:0:-1]
    run = np.concatenate((np.zeros((1,) + dm.shape[1:]), np.cumsum(tail, axis=0)), axis=0)
    ssum = 2 * dm.sum(axis=0)[None] - run
    ssum = ssum / np.arange(n_t, 0, -1, dtype=np.float64).reshape((-1,) + (1,) * (dm.ndim - 1))
    return np.moveaxis(ssum, 0, axis) - s2


def msd_fft(pos1, pos2=None, axis: int = None, *, average: bool = True) -> np.ndarray:
    r"""
    Mean squared displacement (``pos2=None``) or cross displacement by FFT:

    .. math:: \mathrm{MSD}_m=S_m-2A_m,\quad
              S_m(N_t-m)=2\sum_k D_k-\sum_{k=1}^{m}\left(D_{k-1}+D_{N_t-k}\right),\quad
              D_k=\mathbf r_1(k)\cdot\mathbf r_2(k)

    with :math:`A_m` from :func:`correlation_fft` (GPU).  ``pos``: ``(N_t, d)``,
    ``(N_t, N, d)``, ``(N_b, N_t, d)`` or ``(N_b, N_t, N, d)``; ``average`` averages
    over the particle axis when there is one.
    """
    pos1, pos2, axis = _validate(pos1, pos2, axis, 2, "position arrays", lambda nd: nd == 3)
    ndim = pos1.ndim
    n_t = pos1.shape[axis]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s2 = correlation_fft(pos1, pos2, axis, average=False, double=True, vector=True)
    d = (pos1 * (pos1 if pos2 is None else pos2)).sum(axis=-1)
    if ndim - axis == 3 and average:
        s2 = s2.mean(axis=ndim - 2)
        d = d.mean(axis=ndim - 2)
    return _msd_from_parts(d, s2, axis, n_t)


def msd_shift(pos1, pos2=None, axis: int = None, *, average: bool = True) -> np.ndarray:
    r"""
    MSD / cross displacement from the Einstein definition,
    :math:`\langle[\mathbf r_1(t_0+\tau)-\mathbf r_1(t_0)]\cdot[\mathbf r_2(t_0+\tau)-\mathbf r_2(t_0)]\rangle_{t_0}`
    — O(:math:`N_t^2`), host NumPy; same arguments as :func:`msd_fft`.
    """
    pos1, pos2, axis = _validate(pos1, pos2, axis, 2, "position arrays", lambda nd: nd == 3)
    ndim = pos1.ndim
    n_t = pos1.shape[axis]
    a = np.moveaxis(np.asarray(pos1, dtype=np.float64), axis, 0)
    b = a if pos2 is None else np.moveaxis(np.asarray(pos2, dtype=np.float64), axis, 0)
    rows = []
    for m in range(n_t):
        da = a[m:] - a[:n_t - m]
        db = da if pos2 is None else b[m:] - b[:n_t - m]
        rows.append((da * db).sum(axis=-1).mean(axis=0))
    disp = np.moveaxis(np.stack(rows), 0, axis)
    if ndim - axis == 3 and average:
        disp = disp.mean(axis=ndim - 2)
    return disp
