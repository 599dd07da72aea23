#!/usr/bin/env python3
"""
Generate the golden input/output vectors under tests/golden/ by running the
reference's own Python where it is importable in the build container.

Run once, here (``/root/reference`` does not exist on the GPU box); the
``.npz`` outputs are committed, the reference never travels.

* ``src/mdhelper/algorithm/correlation.py`` needs only numpy + scipy and is
  loaded by file path (the package ``__init__`` needs ``pint``, absent here).
* ``src/mdhelper/algorithm/accelerated.py`` needs ``numba`` (absent): it is
  loaded with an inert stand-in module whose ``njit`` is the identity
  decorator and whose ``prange`` is ``range``, so the reference's loop bodies
  run as plain Python (slow, hence the small sizes).

The analysis classes (``structure.py``, ``transport.py``) import MDAnalysis at
module top and cannot be imported here; no fixture is made for them.
"""

import importlib.util
import pathlib
import sys
import types
import warnings

import numpy as np

REF = pathlib.Path("/root/reference/src/mdhelper/algorithm")
OUT = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden"


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def numba_stand_in():
    nb = types.ModuleType("numba")

    def njit(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda fn: fn

    nb.njit = njit
    nb.prange = range
    return nb


def make_correlation(corr):
    rng = np.random.default_rng(20261003)
    # drawn first so that SURVEY.md §8c's anchor values apply to ``walk``
    walk = np.cumsum(rng.normal(size=(2, 64, 5, 3)), axis=1)
    walk2 = np.cumsum(rng.normal(size=(2, 64, 5, 3)), axis=1)
    a = rng.normal(size=(3, 37, 5, 3))
    b = rng.normal(size=(3, 37, 5, 3))
    out = {"a": a, "b": b, "walk": walk, "walk2": walk2}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        cases = {
            # name: (function, args, kwargs)
            "acf_1d": ("correlation_fft", (a[0, :, 0, 0],), {}),
            "acf_2d_axis0": ("correlation_fft", (a[0, :, :, 0],), {"axis": 0}),
            "acf_2d_axis1": ("correlation_fft", (a[:, :, 0, 0],), {"axis": 1}),
            "acf_vec_axis0": ("correlation_fft", (a[0, :, 0],), {"axis": 0, "vector": True}),
            "acf_3d_vec": ("correlation_fft", (a[0],), {"axis": 0, "vector": True}),
            "acf_3d_vec_avg": ("correlation_fft", (a[0],), {"axis": 0, "vector": True, "average": True}),
            "acf_4d_vec": ("correlation_fft", (a,), {"axis": 1, "vector": True}),
            "acf_4d_vec_dbl_avg": ("correlation_fft", (a,), {"axis": 1, "vector": True, "double": True, "average": True}),
            "acf_4d_scalar": ("correlation_fft", (a,), {"axis": 1}),
            "ccf_1d": ("correlation_fft", (a[0, :, 0, 0], b[0, :, 0, 0]), {}),
            "ccf_1d_dbl": ("correlation_fft", (a[0, :, 0, 0], b[0, :, 0, 0]), {"double": True}),
            "ccf_2d_axis1": ("correlation_fft", (a[:, :, 0, 0], b[:, :, 0, 0]), {"axis": 1}),
            "ccf_3d_vec": ("correlation_fft", (a[0], b[0]), {"axis": 0, "vector": True}),
            "ccf_4d_vec": ("correlation_fft", (a, b), {"axis": 1, "vector": True}),
            "ccf_4d_vec_dbl": ("correlation_fft", (a, b), {"axis": 1, "vector": True, "double": True}),
            "shift_acf_4d_vec": ("correlation_shift", (a,), {"axis": 1, "vector": True}),
            "shift_ccf_4d_vec": ("correlation_shift", (a, b), {"axis": 1, "vector": True}),
            "shift_ccf_1d_dbl": ("correlation_shift", (a[0, :, 0, 0], b[0, :, 0, 0]), {"axis": 0, "double": True}),
            "shift_acf_2d_avg": ("correlation_shift", (a[0, :, :, 0],), {"axis": 0, "average": True}),
            "msd_self": ("msd_fft", (walk,), {"axis": 1, "average": False}),
            "msd_avg": ("msd_fft", (walk,), {"axis": 1}),
            "msd_coll": ("msd_fft", (walk.sum(axis=2),), {"axis": 1}),
            "msd_cross": ("msd_fft", (walk.sum(axis=2), walk2.sum(axis=2)), {"axis": 1}),
            "msd_cross_particles": ("msd_fft", (walk, walk2), {"axis": 1, "average": False}),
            "msd_tn3_axis0": ("msd_fft", (walk[0],), {"axis": 0, "average": False}),
            "msd_t3_axis0": ("msd_fft", (walk[0, :, 0],), {"axis": 0}),
            "msd_shift_self": ("msd_shift", (walk,), {"axis": 1, "average": False}),
            "msd_shift_cross": ("msd_shift", (walk.sum(axis=2), walk2.sum(axis=2)), {"axis": 1}),
        }
        for name, (fn, args, kwargs) in cases.items():
            out["out_" + name] = getattr(corr, fn)(*args, **kwargs)
    # the reference tests' closed-form trajectories (test_algorithm_correlation.py:438-443)
    traj_1 = np.array(((0, 0, 0), (1, 1, 1), (2, 2, 2), (3, 3, 3)))
    traj_2 = np.array(((0, 1, 2), (2, 3, 4), (4, 5, 6), (6, 7, 8)))
    out["traj_1"], out["traj_2"] = traj_1, traj_2
    out["out_traj_1"] = corr.msd_fft(traj_1)
    out["out_traj_2"] = corr.msd_fft(traj_2)
    out["out_traj_cd"] = corr.msd_fft(traj_1, traj_2)
    np.savez_compressed(OUT / "correlation_ref.npz", **out)
    print("correlation:", len(out), "arrays")


def make_fourier(acc):
    rng = np.random.default_rng(20261004)
    L = 12.5
    qs = 2 * np.pi / L * rng.integers(-4, 5, size=(24, 3)).astype(np.float64)
    qs[0] = 0.0
    rs = L * rng.random((300, 3))
    rs2 = L * rng.random((200, 3))
    out = {"qs": qs, "rs": rs, "rs2": rs2}
    out["out_fourier_sum"] = acc.delta_fourier_transform_sum_2d_2d(qs, rs)
    out["out_fourier_sum_parallel"] = acc.delta_fourier_transform_sum_parallel_2d_2d(qs, rs2)
    qr = acc.inner_2d_2d(qs, rs)
    qr2 = acc.inner_2d_2d(qs, rs2)
    out["out_inner"] = qr
    out["out_pythag"] = np.array([acc.pythagorean_trigonometric_identity_1d(row) for row in qr])
    out["out_pythag_cross"] = np.array(
        [acc.pythagorean_trigonometric_identity_1d_1d(r1, r2) for r1, r2 in zip(qr, qr2)])
    # the row sums the trigonometric forms and the ISF are built from (accelerated.py:323-627), every variant
    out["out_inner_parallel"] = acc.inner_parallel_2d_2d(qs, rs2)
    out["out_cosine_sum_1d"] = np.array(acc.cosine_sum_1d(qr[3]))
    out["out_sine_sum_1d"] = np.array(acc.sine_sum_1d(qr[3]))
    out["out_cosine_sum_2d"] = acc.cosine_sum_2d(qr)
    out["out_sine_sum_2d"] = acc.sine_sum_2d(qr)
    out["out_cosine_sum_parallel_2d"] = acc.cosine_sum_parallel_2d(qr2)
    out["out_sine_sum_parallel_2d"] = acc.sine_sum_parallel_2d(qr2)
    for name in ("cosine_sum_inplace_2d", "cosine_sum_inplace_parallel_2d", "sine_sum_inplace_2d",
                 "sine_sum_inplace_parallel_2d"):
        hold = np.full(len(qs), np.nan)
        getattr(acc, name)(qr2, hold)
        out["out_" + name] = hold
    out["out_dot_1d_1d"] = np.array(acc.dot_1d_1d(qs[5], rs[7]))
    out["out_delta_1d_1d"] = np.array(acc.delta_fourier_transform_1d_1d(qs[5], rs[7]))
    np.savez_compressed(OUT / "fourier_ref.npz", **out)
    print("fourier:", len(out), "arrays")


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    corr = load("ref_correlation", REF / "correlation.py")
    make_correlation(corr)
    sys.modules["numba"] = numba_stand_in()
    try:
        acc = load("ref_accelerated", REF / "accelerated.py")
        make_fourier(acc)
    finally:
        del sys.modules["numba"]


if __name__ == "__main__":
    main()
