"""
Oracle pinning: oracle.correlation against the reference's own
correlation.py outputs (tests/golden/correlation_ref.npz, made by
scripts/make_golden.py) and the closed-form cases of the reference's
tests/test_algorithm_correlation.py.
"""
import warnings

import numpy as np
import pytest

from oracle import correlation as oc


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(golden_dir / "correlation_ref.npz")


def _cases(g):
    a, b, walk, walk2 = g["a"], g["b"], g["walk"], g["walk2"]
    return {
        "acf_1d": (oc.correlation_fft_ref, (a[0, :, 0, 0],), {}),
        "acf_2d_axis0": (oc.correlation_fft_ref, (a[0, :, :, 0],), {"axis": 0}),
        "acf_2d_axis1": (oc.correlation_fft_ref, (a[:, :, 0, 0],), {"axis": 1}),
        "acf_vec_axis0": (oc.correlation_fft_ref, (a[0, :, 0],), {"axis": 0, "vector": True}),
        "acf_3d_vec": (oc.correlation_fft_ref, (a[0],), {"axis": 0, "vector": True}),
        "acf_3d_vec_avg": (oc.correlation_fft_ref, (a[0],), {"axis": 0, "vector": True, "average": True}),
        "acf_4d_vec": (oc.correlation_fft_ref, (a,), {"axis": 1, "vector": True}),
        "acf_4d_vec_dbl_avg": (oc.correlation_fft_ref, (a,), {"axis": 1, "vector": True, "double": True, "average": True}),
        "acf_4d_scalar": (oc.correlation_fft_ref, (a,), {"axis": 1}),
        "ccf_1d": (oc.correlation_fft_ref, (a[0, :, 0, 0], b[0, :, 0, 0]), {}),
        "ccf_1d_dbl": (oc.correlation_fft_ref, (a[0, :, 0, 0], b[0, :, 0, 0]), {"double": True}),
        "ccf_2d_axis1": (oc.correlation_fft_ref, (a[:, :, 0, 0], b[:, :, 0, 0]), {"axis": 1}),
        "ccf_3d_vec": (oc.correlation_fft_ref, (a[0], b[0]), {"axis": 0, "vector": True}),
        "ccf_4d_vec": (oc.correlation_fft_ref, (a, b), {"axis": 1, "vector": True}),
        "ccf_4d_vec_dbl": (oc.correlation_fft_ref, (a, b), {"axis": 1, "vector": True, "double": True}),
        "shift_acf_4d_vec": (oc.correlation_shift_ref, (a,), {"axis": 1, "vector": True}),
        "shift_ccf_4d_vec": (oc.correlation_shift_ref, (a, b), {"axis": 1, "vector": True}),
        "shift_ccf_1d_dbl": (oc.correlation_shift_ref, (a[0, :, 0, 0], b[0, :, 0, 0]), {"axis": 0, "double": True}),
        "shift_acf_2d_avg": (oc.correlation_shift_ref, (a[0, :, :, 0],), {"axis": 0, "average": True}),
        "msd_self": (oc.msd_fft_ref, (walk,), {"axis": 1, "average": False}),
        "msd_avg": (oc.msd_fft_ref, (walk,), {"axis": 1}),
        "msd_coll": (oc.msd_fft_ref, (walk.sum(axis=2),), {"axis": 1}),
        "msd_cross": (oc.msd_fft_ref, (walk.sum(axis=2), walk2.sum(axis=2)), {"axis": 1}),
        "msd_cross_particles": (oc.msd_fft_ref, (walk, walk2), {"axis": 1, "average": False}),
        "msd_tn3_axis0": (oc.msd_fft_ref, (walk[0],), {"axis": 0, "average": False}),
        "msd_t3_axis0": (oc.msd_fft_ref, (walk[0, :, 0],), {"axis": 0}),
        "msd_shift_self": (oc.msd_shift_ref, (walk,), {"axis": 1, "average": False}),
        "msd_shift_cross": (oc.msd_shift_ref, (walk.sum(axis=2), walk2.sum(axis=2)), {"axis": 1}),
    }


def test_oracle_matches_reference_outputs(g):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, (fn, args, kwargs) in _cases(g).items():
            got = fn(*args, **kwargs)
            want = g["out_" + name]
            assert got.shape == want.shape, name
            # MSD values at lag 0 are pure round-off -> absolute tolerance
            assert np.allclose(got, want, rtol=1e-10, atol=1e-9), name


def test_survey_anchor_values(g):
    """SURVEY.md §8c anchor values of the reference msd_fft."""
    assert np.allclose(g["out_msd_self"][0, 1:4, 0], [3.07981827, 6.21258610, 8.63888675])
    assert np.isclose(g["out_msd_self"][1, 63, 4], 432.6780414037383)
    assert np.allclose(g["out_msd_coll"][0, 1:4], [14.27461227, 28.30115629, 41.9702773])


def test_closed_form_msd(g):
    """tests/test_algorithm_correlation.py:438-472 (reference) closed forms."""
    traj_1, traj_2 = g["traj_1"], g["traj_2"]
    assert np.allclose(oc.msd_fft_ref(traj_1), [0, 3, 12, 27])
    assert np.allclose(oc.msd_fft_ref(traj_2), [0, 12, 48, 108])
    assert np.allclose(oc.msd_fft_ref(traj_1, traj_2), [0, 6, 24, 54])
    assert np.allclose(oc.msd_shift_ref(traj_1), [0, 3, 12, 27])
    assert np.allclose(oc.msd_shift_ref(traj_1, traj_2), [0, 6, 24, 54])
    assert np.allclose(g["out_traj_1"], [0, 3, 12, 27])
    assert np.allclose(g["out_traj_cd"], [0, 6, 24, 54])


def test_fft_equals_shift():
    rng = np.random.default_rng(7)
    x = np.cumsum(rng.normal(size=(2, 50, 4, 3)), axis=1)
    y = np.cumsum(rng.normal(size=(2, 50, 4, 3)), axis=1)
    assert np.allclose(oc.msd_fft_ref(x, axis=1, average=False),
                       oc.msd_shift_ref(x, axis=1, average=False), atol=1e-9)
    assert np.allclose(oc.msd_fft_ref(x, y, axis=1), oc.msd_shift_ref(x, y, axis=1), atol=1e-9)
    assert np.allclose(oc.correlation_fft_ref(x, y, axis=1, vector=True),
                       oc.correlation_shift_ref(x, y, axis=1, vector=True), atol=1e-9)


def test_errors():
    with pytest.raises(ValueError):
        oc.correlation_fft_ref(np.empty(0))
    with pytest.raises(ValueError):
        oc.correlation_fft_ref(np.empty((2, 2, 2, 2, 2)))
    with pytest.raises(ValueError):
        oc.correlation_fft_ref(np.ones((2, 2, 2)), axis=2)
    with pytest.raises(ValueError):
        oc.msd_fft_ref(np.ones((4, 3)), np.ones((1, 3)))
