"""
Oracle pinning: oracle.fourier against the reference's accelerated.py loop
bodies (tests/golden/fourier_ref.npz) and analytic lattice answers.
"""
import numpy as np
import pytest

from oracle import fourier as of


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(golden_dir / "fourier_ref.npz")


def test_fourier_sum_matches_reference(g):
    assert np.allclose(of.fourier_sum_ref(g["qs"], g["rs"]), g["out_fourier_sum"], rtol=1e-12, atol=1e-10)
    assert np.allclose(of.fourier_sum_ref(g["qs"], g["rs2"]), g["out_fourier_sum_parallel"], rtol=1e-12, atol=1e-10)
    assert np.allclose(of.inner_ref(g["qs"], g["rs"]), g["out_inner"], rtol=0, atol=1e-12)


def test_trig_form_matches_reference(g):
    n1, n2 = len(g["rs"]), len(g["rs2"])
    pos = np.vstack((g["rs"], g["rs2"]))
    slices = [slice(0, n1), slice(n1, n1 + n2)]
    pairs = of.ssf_pairs(2, "partial")
    trig = of.ssf_frame_ref(g["qs"], pos, slices, pairs, "partial", "trig")
    expf = of.ssf_frame_ref(g["qs"], pos, slices, pairs, "partial", "exp")
    assert np.allclose(trig[0], g["out_pythag"], rtol=1e-10, atol=1e-8)
    assert np.allclose(trig[1], g["out_pythag_cross"], rtol=1e-10, atol=1e-8)
    assert np.allclose(trig, expf, rtol=1e-9, atol=1e-7)


def test_simple_cubic_bragg_peaks():
    """S(q) of a perfect simple-cubic lattice: N at reciprocal-lattice vectors, 0 elsewhere on the grid."""
    n, a = 4, 1.5
    L = n * a
    idx = np.arange(n)
    pos = a * np.stack(np.meshgrid(idx, idx, idx, indexing="ij"), -1).reshape(-1, 3)
    q = of.grid_wavevectors([L, L, L], 2 * n)
    res = of.ssf_run_ref(pos[None].astype(np.float64), [n ** 3], q, sort=False, unique=False)
    m = np.rint(q * L / (2 * np.pi)).astype(int)
    bragg = np.all(m % n == 0, axis=1)
    assert np.allclose(res["ssf"][0][bragg], n ** 3)
    assert np.allclose(res["ssf"][0][~bragg], 0, atol=1e-9)


def test_meshgrid_row_order():
    q = of.grid_wavevectors([10.0, 10.0, 10.0], 3)
    g1 = 2 * np.pi / 10.0
    # numpy.meshgrid default 'xy' indexing: row index = (j, i, k)
    assert np.allclose(q[1], [0, 0, g1])
    assert np.allclose(q[3], [g1, 0, 0])
    assert np.allclose(q[9], [0, g1, 0])
