"""
GPU parity tests of the operator surface (RadialDistributionFunction,
StructureFactor, Onsager, correlation_fft / msd_fft, radial_histogram) against
the oracle's restatement of the reference drivers and against the golden
vectors produced by the reference's own correlation.py / accelerated.py.
"""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import mdhelper_amd  # noqa: E402
from mdhelper_amd.algorithm import correlation  # noqa: E402
from mdhelper_amd.analysis import Onsager, RadialDistributionFunction, StructureFactor, structure  # noqa: E402
from oracle import correlation as oc  # noqa: E402
from oracle import fourier as of  # noqa: E402
from oracle import rdf as orf  # noqa: E402


def lj_melt(F=100, n_side=10, seed=1):
    """C1: 1000-atom jittered simple-cubic lattice, rho* = 0.8442."""
    a = 1.0577
    L = n_side * a
    rng = np.random.default_rng(seed)
    idx = np.arange(n_side)
    lattice = a * np.stack(np.meshgrid(idx, idx, idx, indexing="ij"), -1).reshape(-1, 3)
    frames = lattice[None] + rng.normal(scale=0.1 * a, size=(F, n_side ** 3, 3))
    return np.mod(frames, L).astype(np.float32), np.float32(L)


def test_c1_rdf_lj_melt_matches_reference_driver():
    """BASELINE config C1: RDF on a 1000-atom LJ melt, 100 frames."""
    frames, L = lj_melt()
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    u = mdhelper_amd.ArrayUniverse(frames, dims)
    rng_range = (0.0, float(L) / 2)
    rdf = RadialDistributionFunction(u.atoms, n_bins=201, range=rng_range, exclusion=(1, 1)).run()
    ref = orf.rdf_run_ref(frames, dims, 201, rng_range, exclusion=(1, 1))
    assert rdf.results.counts.dtype == np.int64
    assert np.array_equal(rdf.results.counts, ref["counts"])
    assert np.allclose(rdf.results.rdf, ref["rdf"], rtol=1e-6)
    assert np.array_equal(rdf.results.edges, ref["edges"]) and np.array_equal(rdf.results.bins, ref["bins"])
    # generic per-frame path (what a real MDAnalysis universe takes) gives the same counts
    slow = RadialDistributionFunction(u.atoms, n_bins=201, range=rng_range, exclusion=(1, 1))
    slow.run(start=0, stop=100, step=3)
    ref3 = orf.rdf_run_ref(frames[::3], dims, 201, rng_range, exclusion=(1, 1))
    assert np.array_equal(slow.results.counts, ref3["counts"])
    # post-processing runs on the result
    rdf.calculate_coordination_numbers(0.8442 / 1.0577 ** 3 * 1.0577 ** 3)
    rdf.calculate_pmf(300)
    rdf.calculate_structure_factor(0.8442, n_q=64)
    assert rdf.results.ssf.shape == (64,) and np.isfinite(rdf.results.pmf[50:]).all()


def test_rdf_two_groups_norms_and_drop_axis():
    rng = np.random.default_rng(2)
    L = np.float32(24.0)
    frames = (rng.random((7, 900, 3)) * L).astype(np.float32)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    u = mdhelper_amd.ArrayUniverse(frames, dims, resids=np.arange(900) // 3)
    a, b = u.atoms[:400], u.atoms[400:]
    for norm in ("rdf", "density", None):
        r = RadialDistributionFunction(a, b, 50, (0.5, 10.0), norm=norm).run()
        ref = orf.rdf_run_ref(frames, dims, 50, (0.5, 10.0), sel1=slice(0, 400), sel2=slice(400, 900), norm=norm)
        assert np.array_equal(r.results.counts, ref["counts"])
        assert np.allclose(r.results.rdf, ref["rdf"], rtol=1e-6)
        assert np.allclose(r._get_rdf(), orf.rdf_run_ref(frames, dims, 50, (0.5, 10.0), sel1=slice(0, 400),
                                                         sel2=slice(400, 900))["rdf"], rtol=1e-6)
    # residue centres of mass: formed on the device (mdx_rdf_set_grouping), float32 like the host path
    r = RadialDistributionFunction(u.atoms, n_bins=30, range=(0.0, 8.0), groupings="residues",
                                   exclusion=(1, 1)).run()
    com = frames.reshape(7, 300, 3, 3).astype(np.float64).mean(axis=2).astype(np.float32)
    ref = orf.rdf_run_ref(com, dims, 30, (0.0, 8.0), exclusion=(1, 1))
    assert np.array_equal(r.results.counts, ref["counts"])
    # 2-D mode: drop z
    r = RadialDistributionFunction(u.atoms, n_bins=30, range=(0.0, 8.0), drop_axis="z", exclusion=(1, 1)).run()
    flat = frames.copy()
    flat[..., 2] = 0
    want = np.zeros(30, dtype=np.int64)
    for f in range(7):
        want += orf.radial_histogram_ref(flat[f], flat[f], 30, (0.0, 8.0), dims, exclusion=(1, 1))
    assert np.array_equal(r.results.counts, want)
    assert np.isfinite(r.results.rdf).all()


def test_radial_histogram_function():
    rng = np.random.default_rng(11)
    L = 20
    half_L = L // 2
    dims = np.array((L, L, L, 90, 90, 90), dtype=int)
    origin = half_L * np.ones(3)
    norm = half_L * rng.random(1000)
    neighbors = rng.random((1000, 3))
    neighbors *= norm[:, None] / np.linalg.norm(neighbors, axis=1, keepdims=True)
    neighbors += dims[:3] / 2
    got = structure.radial_histogram(origin, neighbors, n_bins=half_L, range=(0, half_L + 1), dims=dims)
    want = orf.radial_histogram_ref(origin, neighbors, half_L, (0, half_L + 1), dims)
    assert np.array_equal(got, want)
    tri = [20, 20, 20, 90, 60, 90]
    assert np.array_equal(structure.radial_histogram(origin, neighbors, 4, (0, 5), tri),
                          orf.radial_histogram_ref(origin, neighbors, 4, (0, 5), tri))


@pytest.mark.parametrize("mode", [None, "pair", "partial"])
@pytest.mark.parametrize("form", ["exp", "trig"])
def test_structure_factor_matches_reference_driver(mode, form):
    rng = np.random.default_rng(3)
    L = 18.0
    frames = (rng.random((5, 600, 3)) * L).astype(np.float32)
    u = mdhelper_amd.ArrayUniverse(frames, [L, L, L, 90, 90, 90])
    groups = (u.atoms[:350], u.atoms[350:])
    sf = StructureFactor(groups, mode=mode, form=form, n_points=5).run()
    q = of.grid_wavevectors([L, L, L], 5)
    ref = of.ssf_run_ref(frames.astype(np.float64), [350, 250], q, mode=mode, form=form)
    assert sf.results.pairs == ref["pairs"]
    assert np.allclose(sf.results.wavenumbers, ref["wavenumbers"])
    assert np.allclose(sf.results.ssf, ref["ssf"], rtol=1e-6, atol=1e-8)
    raw = StructureFactor(groups, mode=mode, n_points=3, sort=False, unique=False).run(step=2)
    ref = of.ssf_run_ref(frames[::2].astype(np.float64), [350, 250], of.grid_wavevectors([L, L, L], 3),
                         mode=mode, sort=False, unique=False)
    assert np.allclose(raw.results.ssf, ref["ssf"], rtol=1e-6, atol=1e-8)


def test_structure_factor_reference_default_grid_of_32_points():
    """The reference's default configuration: ``n_points=32`` => 32 768 grid wavevectors
    (structure.py:1324, 1376-1381): 32-long m_z lists in the register-blocked column kernel, several
    column chunks, and the round(11) fold of 32 768 rows into the unique wavenumbers.  ~4 000 atoms in
    two groups, 2 frames, ``mode="partial"``, sort/unique on and off, against oracle/fourier.py."""
    rng = np.random.default_rng(33)
    L = float(np.float32(34.2))          # the cell as the universe stores it (float32 dimensions)
    n1, n2 = 2300, 1700
    frames = (rng.random((2, n1 + n2, 3)) * L).astype(np.float32)
    u = mdhelper_amd.ArrayUniverse(frames, [L, L, L, 90, 90, 90])
    groups = (u.atoms[:n1], u.atoms[n1:])
    q = of.grid_wavevectors([L, L, L], 32)
    assert q.shape == (32768, 3)
    raw_ref = of.ssf_run_ref(frames.astype(np.float64), [n1, n2], q, mode="partial", sort=False, unique=False)
    raw = StructureFactor(groups, mode="partial", sort=False, unique=False).run()       # n_points default
    assert raw._wavevectors.shape == (32768, 3) and np.array_equal(raw._wavevectors, q)
    assert raw.results.pairs == raw_ref["pairs"] and raw.results.ssf.shape == (3, 32768)
    # element-wise: the diagonal columns relative, the cross column against sqrt(S_00 S_11) (its
    # entries pass through zero); q = 0 (N_j N_k / N) does not enter any norm
    ref, got = raw_ref["ssf"], raw.results.ssf
    ntot = n1 + n2
    for i, (j, k) in enumerate(raw_ref["pairs"]):
        bound = (1e-6 * np.abs(ref[i]) if j == k else 1e-6 * np.sqrt(ref[0] * ref[2])) + 1e-9
        assert np.all(np.abs(got[i] - ref[i]) <= bound), (i, float(np.abs(got[i] - ref[i]).max()))
    assert np.isclose(got[0, 0], n1 * n1 / ntot, rtol=1e-12) and np.isclose(got[1, 0], 2 * n1 * n2 / ntot, rtol=1e-12)
    # the folded form (the default): unique wavenumbers by round(11), columns averaged, sorted
    ref = of.ssf_run_ref(frames.astype(np.float64), [n1, n2], q, mode="partial")
    sf = StructureFactor(groups, mode="partial").run()
    assert sf.results.wavenumbers.shape == ref["wavenumbers"].shape and len(ref["wavenumbers"]) > 1500
    assert np.allclose(sf.results.wavenumbers, ref["wavenumbers"], rtol=1e-12, atol=0)
    for i, (j, k) in enumerate(ref["pairs"]):
        bound = (1e-6 * np.abs(ref["ssf"][i]) if j == k
                 else 1e-6 * np.sqrt(np.abs(ref["ssf"][0] * ref["ssf"][2]))) + 1e-9
        assert np.all(np.abs(sf.results.ssf[i] - ref["ssf"][i]) <= bound), i


def test_structure_factor_q_max_keeps_a_sphere_octant_of_the_grid():
    """``q_max`` (structure.py:1412-1414) keeps the grid wavevectors with |q| <= q_max — a sphere octant: the
    regular quad items built from aligned blocks, entries outside the sphere dropped.  Element-wise against
    oracle/fourier.py on the kept wavevectors, ``mode=None`` and ``"partial"``, raw and folded."""
    rng = np.random.default_rng(35)
    L = float(np.float32(29.7))
    n1, n2 = 1500, 1100
    frames = (rng.random((3, n1 + n2, 3)) * L).astype(np.float32)
    u = mdhelper_amd.ArrayUniverse(frames, [L, L, L, 90, 90, 90])
    groups = (u.atoms[:n1], u.atoms[n1:])
    q_all = of.grid_wavevectors([L, L, L], 24)
    q_max = 0.55 * np.linalg.norm(q_all, axis=1).max() / np.sqrt(3.0) * 1.7     # inside the cube's faces and beyond
    q = q_all[np.linalg.norm(q_all, axis=1) <= q_max]
    assert 1000 < len(q) < len(q_all)
    raw_ref = of.ssf_run_ref(frames.astype(np.float64), [n1, n2], q, mode="partial", sort=False, unique=False)
    raw = StructureFactor(groups, n_points=24, q_max=q_max, mode="partial", sort=False, unique=False).run()
    assert np.array_equal(raw._wavevectors, q)
    ref, got = raw_ref["ssf"], raw.results.ssf
    for i, (j, k) in enumerate(raw_ref["pairs"]):
        bound = (1e-6 * np.abs(ref[i]) if j == k else 1e-6 * np.sqrt(ref[0] * ref[2])) + 1e-9
        assert np.all(np.abs(got[i] - ref[i]) <= bound), (i, float(np.abs(got[i] - ref[i]).max()))
    ref = of.ssf_run_ref(frames.astype(np.float64), [n1 + n2], q, mode=None)
    sf = StructureFactor(u.atoms, n_points=24, q_max=q_max).run()
    assert np.allclose(sf.results.wavenumbers, ref["wavenumbers"], rtol=1e-12, atol=0)
    assert np.all(np.abs(sf.results.ssf[0] - ref["ssf"][0]) <= 1e-6 * np.abs(ref["ssf"][0]) + 1e-9)


def test_structure_factor_bragg_peaks():
    n, a = 6, 1.5
    L = n * a
    idx = np.arange(n)
    pos = (a * np.stack(np.meshgrid(idx, idx, idx, indexing="ij"), -1).reshape(-1, 3)).astype(np.float32)
    u = mdhelper_amd.ArrayUniverse(pos[None], [L, L, L, 90, 90, 90])
    sf = StructureFactor(u.atoms, n_points=2 * n, sort=False, unique=False).run()
    m = np.rint(sf._wavevectors * L / (2 * np.pi)).astype(int)
    bragg = np.all(m % n == 0, axis=1)
    assert np.allclose(sf.results.ssf[0][bragg], n ** 3, rtol=1e-6)
    assert np.allclose(sf.results.ssf[0][~bragg], 0, atol=1e-6)


def test_correlation_fft_matches_reference_golden(golden_dir):
    g = np.load(golden_dir / "correlation_ref.npz")
    a, b, walk, walk2 = g["a"], g["b"], g["walk"], g["walk2"]
    cases = {
        "acf_1d": (correlation.correlation_fft, (a[0, :, 0, 0],), {}),
        "acf_2d_axis0": (correlation.correlation_fft, (a[0, :, :, 0],), {"axis": 0}),
        "acf_2d_axis1": (correlation.correlation_fft, (a[:, :, 0, 0],), {"axis": 1}),
        "acf_vec_axis0": (correlation.correlation_fft, (a[0, :, 0],), {"axis": 0, "vector": True}),
        "acf_3d_vec": (correlation.correlation_fft, (a[0],), {"axis": 0, "vector": True}),
        "acf_3d_vec_avg": (correlation.correlation_fft, (a[0],), {"axis": 0, "vector": True, "average": True}),
        "acf_4d_vec": (correlation.correlation_fft, (a,), {"axis": 1, "vector": True}),
        "acf_4d_vec_dbl_avg": (correlation.correlation_fft, (a,), {"axis": 1, "vector": True, "double": True, "average": True}),
        "acf_4d_scalar": (correlation.correlation_fft, (a,), {"axis": 1}),
        "ccf_1d": (correlation.correlation_fft, (a[0, :, 0, 0], b[0, :, 0, 0]), {}),
        "ccf_1d_dbl": (correlation.correlation_fft, (a[0, :, 0, 0], b[0, :, 0, 0]), {"double": True}),
        "ccf_2d_axis1": (correlation.correlation_fft, (a[:, :, 0, 0], b[:, :, 0, 0]), {"axis": 1}),
        "ccf_3d_vec": (correlation.correlation_fft, (a[0], b[0]), {"axis": 0, "vector": True}),
        "ccf_4d_vec": (correlation.correlation_fft, (a, b), {"axis": 1, "vector": True}),
        "ccf_4d_vec_dbl": (correlation.correlation_fft, (a, b), {"axis": 1, "vector": True, "double": True}),
        "msd_self": (correlation.msd_fft, (walk,), {"axis": 1, "average": False}),
        "msd_avg": (correlation.msd_fft, (walk,), {"axis": 1}),
        "msd_coll": (correlation.msd_fft, (walk.sum(axis=2),), {"axis": 1}),
        "msd_cross": (correlation.msd_fft, (walk.sum(axis=2), walk2.sum(axis=2)), {"axis": 1}),
        "msd_cross_particles": (correlation.msd_fft, (walk, walk2), {"axis": 1, "average": False}),
        "msd_tn3_axis0": (correlation.msd_fft, (walk[0],), {"axis": 0, "average": False}),
        "msd_t3_axis0": (correlation.msd_fft, (walk[0, :, 0],), {"axis": 0}),
    }
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, (fn, args, kwargs) in cases.items():
            got = fn(*args, **kwargs)
            want = g["out_" + name]
            assert got.shape == want.shape, name
            scale = max(1.0, np.abs(want).max())
            assert np.allclose(got, want, rtol=1e-6, atol=1e-9 * scale), name
    # closed forms of the reference tests (test_algorithm_correlation.py:438-472)
    assert np.allclose(correlation.msd_fft(g["traj_1"].tolist()), [0, 3, 12, 27], atol=1e-9)
    assert np.allclose(correlation.msd_fft(g["traj_2"]), [0, 12, 48, 108], atol=1e-9)
    assert np.allclose(correlation.msd_fft(g["traj_1"], g["traj_2"].tolist()), [0, 6, 24, 54], atol=1e-9)
    # all-ones ACF = 1 (scalar) / d (vector)  (:67-102)
    ones = np.ones((3, 20, 4, 3))
    assert np.allclose(correlation.correlation_fft(ones[0, :, 0, 0]), 1)
    assert np.allclose(correlation.correlation_fft(ones, axis=1, vector=True), 3)
    # complex input goes through four real correlations
    z = a[0, :, 0, 0] + 1j * b[0, :, 0, 0]
    T = len(z)
    want = np.array([np.sum(np.conj(z[:T - m]) * z[m:]) / (T - m) for m in range(T)])
    assert np.allclose(correlation.correlation_fft(z), want)


def _random_walk_universe(T=400, sizes=(30, 20), seed=4, sigma=0.1):
    rng = np.random.default_rng(seed)
    N = sum(sizes)
    pos = 20.0 + np.cumsum(rng.normal(scale=sigma, size=(T, N, 3)), axis=0)
    u = mdhelper_amd.ArrayUniverse(pos, [40.0, 40.0, 40.0, 90, 90, 90], dt=2.0,
                                   charges=np.r_[np.ones(sizes[0]), -np.ones(sizes[1])])
    return u, pos.astype(np.float32).astype(np.float64)


def _onsager_ref(pos, sizes, n_blocks, dims=(40.0, 40.0, 40.0)):
    """Restatement of Onsager._conclude (reference transport.py:1016-1059) on the oracle."""
    T = (pos.shape[0] // n_blocks) * n_blocks
    pos = pos[:T]
    slices, idx = [], 0
    for n in sizes:
        slices.append(slice(idx, idx + n))
        idx += n
    g = len(sizes)
    pairs = [(i, j) for i in range(g) for j in range(i, g)]
    cross = np.empty((len(pairs), n_blocks, T // n_blocks))
    self_ = np.empty((g, n_blocks, T // n_blocks))
    for i, (i1, i2) in enumerate(pairs):
        p1 = pos[:, slices[i1]].reshape(n_blocks, -1, sizes[i1], 3)
        if i1 == i2:
            cross[i] = oc.msd_fft_ref(p1.sum(axis=2), axis=1)
            self_[i1] = oc.msd_fft_ref(p1, axis=1, average=False).sum(axis=-1) / sizes[i1]
        else:
            p2 = pos[:, slices[i2]].reshape(n_blocks, -1, sizes[i2], 3)
            cross[i] = oc.msd_fft_ref(p1.sum(axis=2), p2.sum(axis=2), axis=1)
    return cross / 6, self_ / 6


@pytest.mark.parametrize("n_blocks", [1, 4])
def test_onsager_matches_reference_driver(n_blocks):
    u, pos = _random_walk_universe()
    groups = (u.atoms[:30], u.atoms[30:])
    ons = Onsager(groups, temperature=1.0, reduced=True, n_blocks=n_blocks).run()
    cross, self_ = _onsager_ref(pos, (30, 20), n_blocks)
    assert ons.results.pairs == ((0, 0), (0, 1), (1, 1))
    assert np.allclose(ons.results.times, 2.0 * np.arange(400 // n_blocks))
    assert np.allclose(ons.results.msd_self, self_, rtol=1e-6, atol=1e-8)
    assert np.allclose(ons.results.msd_cross, cross, rtol=1e-6, atol=1e-7)
    # physics: MSD(m)/6 = sigma^2 m / 2 for a free walk
    m = np.arange(1, 20)
    assert np.allclose(ons.results.msd_self[0, 0, 1:20], 0.01 * m / 2, rtol=0.25)
    # direct definition agrees with the FFT route (reference test_algorithm_correlation.py:410-436)
    direct = Onsager(groups, temperature=1.0, reduced=True, n_blocks=n_blocks, fft=False).run(stop=120)
    fftrun = Onsager(groups, temperature=1.0, reduced=True, n_blocks=n_blocks).run(stop=120)
    assert np.allclose(direct.results.msd_self, fftrun.results.msd_self, rtol=1e-6, atol=1e-8)
    assert np.allclose(direct.results.msd_cross, fftrun.results.msd_cross, rtol=1e-6, atol=1e-7)
    ons.calculate_transport_coefficients(start=1, stop=60, scale="linear")
    ons.calculate_conductivity()
    ons.calculate_electrophoretic_mobility()
    ons.calculate_transference_number()
    assert ons.results.L_ij.shape == (n_blocks, 2, 2) and np.isfinite(ons.results.D_i).all()
    assert np.allclose(ons.results.transference_numbers.sum(axis=-1), 1)


@pytest.mark.parametrize("T,n_blocks,n_fft", [(2400, 4, 1600), (2400, 1, 6400), (9000, 2, 12800), (9000, 1, 25600),
                                               (8000, 1, 16384), (30000, 1, 65536), (30001, 3, 25600)])
def test_onsager_on_the_engines_own_transforms(T, n_blocks, n_fft):
    """The class on block lengths that take the engine's own two-pass transforms (400 x 2^k and 2^k points; the
    small cases above run on 800 points or rocFFT): every cross and self MSD element-wise against the oracle's
    restatement of Onsager._conclude (reference transport.py:1016-1059)."""
    from mdhelper_amd import _core
    u, pos = _random_walk_universe(T=T, sizes=(17, 12), seed=T + n_blocks)
    groups = (u.atoms[:17], u.atoms[17:])
    eng = _core.MsdEngine(T // n_blocks, n_blocks, 2)
    assert eng.n_fft == n_fft and eng.transform[0]
    eng.close()
    ons = Onsager(groups, temperature=1.0, reduced=True, n_blocks=n_blocks).run()
    cross, self_ = _onsager_ref(pos, (17, 12), n_blocks)
    assert np.allclose(ons.results.times, 2.0 * np.arange(T // n_blocks))
    assert np.allclose(ons.results.msd_self, self_, rtol=1e-6, atol=1e-8)
    assert np.allclose(ons.results.msd_cross, cross, rtol=1e-6, atol=1e-7)


def test_onsager_blocks_warning_center_and_unwrap():
    u, pos = _random_walk_universe(T=103)
    with pytest.warns(UserWarning):
        ons = Onsager(u.atoms, temperature=1.0, reduced=True, n_blocks=4).run()
    cross, self_ = _onsager_ref(pos, (50,), 4)
    assert np.allclose(ons.results.msd_self, self_, rtol=1e-6, atol=1e-8)
    # centre-of-mass removal (float64 positions here: the per-frame host path)
    cen = Onsager(u.atoms, temperature=1.0, reduced=True, center=True).run()
    centred = pos - pos.mean(axis=1, keepdims=True)
    _, self_c = _onsager_ref(centred, (50,), 1)
    assert np.allclose(cen.results.msd_self, self_c, rtol=1e-5, atol=1e-7)
    # wrapped input + unwrap=True recovers the unwrapped answer
    L = 40.0
    rng = np.random.default_rng(9)
    true = 20.0 + np.cumsum(rng.normal(scale=1.0, size=(150, 12, 3)), axis=0)
    wrapped = np.mod(true, L)
    uw = mdhelper_amd.ArrayUniverse(wrapped, [L, L, L, 90, 90, 90])
    ut = mdhelper_amd.ArrayUniverse(true, [L, L, L, 90, 90, 90])
    a = Onsager(uw.atoms, temperature=1.0, reduced=True, unwrap=True).run()
    b = Onsager(ut.atoms, temperature=1.0, reduced=True).run()
    assert np.allclose(a.results.msd_self, b.results.msd_self, rtol=1e-3, atol=1e-3)


def test_zero_dimension_is_dropped():
    u, pos = _random_walk_universe(T=90)
    ons = Onsager(u.atoms, temperature=1.0, reduced=True, dimensions=[40.0, 40.0, 0.0]).run()
    p = pos.copy()
    p[..., 2] = 0
    ref = oc.msd_fft_ref(p[None], axis=1, average=False).sum(axis=-1) / 50 / 4
    assert np.allclose(ons.results.msd_self[0], ref, rtol=1e-6, atol=1e-8)


def test_save_roundtrip(tmp_path):
    frames, L = lj_melt(F=3)
    u = mdhelper_amd.ArrayUniverse(frames, [L, L, L, 90, 90, 90])
    rdf = RadialDistributionFunction(u.atoms, n_bins=20, range=(0.0, 4.0), exclusion=(1, 1), parallel=True)
    rdf.run(n_jobs=4, module="joblib")
    rdf.save(str(tmp_path / "rdf"))
    data = np.load(tmp_path / "rdf.npz")
    assert np.array_equal(data["counts"], rdf.results.counts)


@pytest.mark.parametrize("mode", [None, "pair", "partial"])
def test_intermediate_scattering_function(mode):
    """ISF vs the restated reference driver (ring buffer semantics, normalisation)."""
    from mdhelper_amd.analysis import IntermediateScatteringFunction
    rng = np.random.default_rng(31)
    L = 16.0
    F, sizes = 23, (90, 60)
    N = sum(sizes)
    pos = np.mod(8.0 + np.cumsum(rng.normal(scale=0.15, size=(F, N, 3)), axis=0), L).astype(np.float32)
    u = mdhelper_amd.ArrayUniverse(pos, [L, L, L, 90, 90, 90], dt=0.5)
    groups = (u.atoms[:90], u.atoms[90:])
    q = of.grid_wavevectors([L, L, L], 3)
    for n_lags in (7, None):
        isf = IntermediateScatteringFunction(groups, mode=mode, n_points=3, n_lags=n_lags,
                                             incoherent=True).run()
        ref = of.isf_run_ref(pos, sizes, q, n_lags, mode=mode, incoherent=True)
        assert isf.results.pairs == ref["pairs"]
        assert np.allclose(isf.results.wavenumbers, ref["wavenumbers"])
        assert np.allclose(isf.results.times, 0.5 * np.arange(n_lags or F))
        scale = np.abs(ref["cisf"]).max()
        assert np.allclose(isf.results.cisf, ref["cisf"], rtol=1e-6, atol=1e-9 * scale)
        assert np.allclose(isf.results.iisf, ref["iisf"], rtol=1e-6, atol=1e-9)
    # lag 0 of the coherent part is the static structure factor of the same frames
    ssf = StructureFactor(groups, mode=mode, n_points=3).run()
    assert np.allclose(isf.results.cisf[0], ssf.results.ssf, rtol=1e-6, atol=1e-9)
    # trig form and coherent-only run give the same coherent part; strided selection works
    raw = IntermediateScatteringFunction(groups, mode=mode, form="trig", n_points=3, n_lags=5,
                                         sort=False, unique=False).run(step=2)
    ref = of.isf_run_ref(pos[::2], sizes, q, 5, mode=mode, sort=False, unique=False)
    assert np.allclose(raw.results.cisf, ref["cisf"], rtol=1e-6, atol=1e-9 * np.abs(ref["cisf"]).max())
    assert "iisf" not in raw.results and np.allclose(raw.results.times, np.arange(5))


def test_rdf_device_centres_of_mass_equal_the_per_frame_path(tmp_path):
    """groupings = residues / segments: centres of mass formed on the device (fast batched path,
    in-memory and from a file) against the generic per-frame path (host NumPy centres)."""
    import sys
    sys.path.insert(0, str(__import__("pathlib").Path(__file__).parent))
    from trajfiles import write_amber_netcdf
    from mdhelper_amd.analysis.base import DynamicAnalysisBase
    rng = np.random.default_rng(61)
    F, N, L = 6, 1200, 26.0
    frames = (rng.random((F, N, 3)) * L).astype(np.float32)
    dims = np.array([L, L + 1, L - 1, 90, 90, 90], dtype=np.float32)
    masses = rng.uniform(1.0, 40.0, N)
    resids = rng.permutation(np.repeat(np.arange(300), 4))        # residues interleaved in atom order
    segids = np.arange(N) // 100
    write_amber_netcdf(tmp_path / "m.nc", frames, dims[:3])
    universes = [mdhelper_amd.ArrayUniverse(frames, dims, masses=masses, resids=resids, segids=segids),
                 mdhelper_amd.FileUniverse(tmp_path / "m.nc", masses=masses, resids=resids, segids=segids)]
    cases = [dict(groupings="residues", exclusion=(1, 1)),
             dict(groupings=("atoms", "residues")),
             dict(groupings=("residues", "segments")),
             dict(groupings="segments", exclusion=(1, 1))]
    for kw in cases:
        ref = None
        for u in universes:
            a, b = u.atoms[:800], u.atoms[300:]
            args = (a, b) if isinstance(kw["groupings"], tuple) else (u.atoms,)
            fast = RadialDistributionFunction(*args, n_bins=40, range=(0.0, 11.0), **kw).run()
            if ref is None:
                slow = RadialDistributionFunction(*args, n_bins=40, range=(0.0, 11.0), **kw)
                DynamicAnalysisBase.run(slow)          # the per-frame driver
                ref = slow.results
                assert ref.counts.sum() > 0
            assert np.array_equal(fast.results.counts, ref.counts), kw
            assert np.allclose(fast.results.rdf, ref.rdf, rtol=1e-12), kw


@pytest.mark.parametrize("case", ["atoms", "residues, two groups, blocks", "unwrap from a file"])
def test_end_to_end_vector_acf_on_device(case, tmp_path):
    """``EndToEndVector`` (reference analysis/polymer.py:510-803): the ACF of the unit end-to-end
    vectors through the correlation engine (power spectra summed over chains and components on the
    device, one inverse transform per group and block) against the frame-by-frame restatement."""
    from mdhelper_amd.analysis import polymer
    from oracle import polymer as op
    rng = np.random.default_rng(12)

    def chains(T, M, n, L):
        conf = np.cumsum(rng.normal(scale=0.55, size=(1, M, n, 3)), axis=2)
        wiggle = np.cumsum(rng.normal(scale=0.06, size=(T, M, n, 3)), axis=0)
        drift = np.cumsum(rng.normal(scale=0.3, size=(T, M, 1, 3)), axis=0)
        return (rng.uniform(0, L, (1, M, 1, 3)) + conf + wiggle + drift).reshape(T, M * n, 3)

    if case == "atoms":
        pos = chains(500, 40, 10, 20.0).astype(np.float32)
        u = mdhelper_amd.ArrayUniverse(pos, [40, 40, 40, 90, 90, 90], dt=2.0)
        a = polymer.EndToEndVector(u.atoms, n_chains=40, n_monomers=10, verbose=False).run()
        ref, _ = op.end_to_end_run_ref(pos, [np.arange(400)], [40], [10], ["atoms"])
    elif case == "residues, two groups, blocks":
        pos = chains(301, 12, 18, 20.0).astype(np.float32)           # 12 chains of 6 monomers of 3 atoms
        masses = rng.uniform(1, 16, pos.shape[1])
        u = mdhelper_amd.ArrayUniverse(pos, [40, 40, 40, 90, 90, 90], dt=2.0, masses=masses)
        g1, g2 = u.atoms[:5 * 18], u.atoms[5 * 18:]
        with pytest.warns(UserWarning, match="not divisible"):
            a = polymer.EndToEndVector([g1, g2], ["residues", "atoms"], n_chains=(5, 7), n_monomers=(6, 18),
                                       n_blocks=3, verbose=False).run()
        ref, _ = op.end_to_end_run_ref(pos, [g1.indices, g2.indices], [5, 7], [6, 18], ["residues", "atoms"],
                                       masses=masses, n_blocks=3)
        assert a.results.acf.shape == (2, 3, 100)
    else:
        from trajfiles import write_amber_netcdf
        L = np.array([11.0, 12.0, 10.5])
        pos = chains(400, 30, 8, 11.0)
        wrapped = np.mod(pos, L).astype(np.float32)
        path = tmp_path / "chains.nc"
        write_amber_netcdf(path, wrapped, L, times=np.arange(400) * 2.0)
        u = mdhelper_amd.FileUniverse(path, dt=2.0)
        a = polymer.EndToEndVector(u.atoms, n_chains=30, n_monomers=8, unwrap=True, n_blocks=2,
                                   verbose=False).run()
        ref, e2e = op.end_to_end_run_ref(wrapped, [np.arange(240)], [30], [8], ["atoms"], dimensions=L,
                                         unwrap=True, n_blocks=2)
        true = pos.reshape(400, 30, 8, 3)
        assert np.allclose(e2e, true[:, :, -1] - true[:, :, 0], atol=1e-4)
    assert np.allclose(a.results.acf, ref, rtol=1e-9, atol=1e-11)
    assert np.allclose(a.results.acf[..., 0], 1.0, atol=1e-12)
    direct = polymer.EndToEndVector(*([u.atoms] if case != "residues, two groups, blocks" else [[g1, g2],
                                    ["residues", "atoms"]]),
                                    **({"n_chains": 40, "n_monomers": 10} if case == "atoms" else
                                       {"n_chains": (5, 7), "n_monomers": (6, 18), "n_blocks": 3}
                                       if case.startswith("residues") else
                                       {"n_chains": 30, "n_monomers": 8, "unwrap": True, "n_blocks": 2}),
                                    fft=False, verbose=False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        direct.run()
    assert np.allclose(direct.results.acf, a.results.acf, rtol=1e-9, atol=1e-11)


def test_end_to_end_relaxation_time_of_rotational_diffusion():
    """Dumbbells whose orientation diffuses on the sphere: C_ee decays exponentially with
    tau = 1 / (2 D_r); the fitted relaxation time recovers it."""
    from mdhelper_amd.analysis import polymer
    rng = np.random.default_rng(4)
    T, M, sigma = 4000, 600, 0.05
    u_vec = rng.normal(size=(M, 3))
    u_vec /= np.linalg.norm(u_vec, axis=1, keepdims=True)
    pos = np.empty((T, 2 * M, 3), dtype=np.float32)
    for t in range(T):
        pos[t, 0::2] = 10.0
        pos[t, 1::2] = 10.0 + u_vec
        u_vec = u_vec + sigma * rng.normal(size=(M, 3))
        u_vec /= np.linalg.norm(u_vec, axis=1, keepdims=True)
    uni = mdhelper_amd.ArrayUniverse(pos, [20, 20, 20, 90, 90, 90], dt=1.0)
    e = polymer.EndToEndVector(uni.atoms, n_chains=M, n_monomers=2, verbose=False).run()
    e.calculate_relaxation_time()
    # per step <u.u'> = 1 - sigma^2 (two transverse components): tau = 1 / sigma^2
    assert e.results.relaxation_times.shape == (1, 1)
    assert np.isclose(e.results.relaxation_times[0, 0], 1 / sigma ** 2, rtol=0.1)
    m = np.arange(1, 200)
    assert np.allclose(e.results.acf[0, 0, 1:200], np.exp(-m * sigma ** 2), atol=0.02)


@pytest.mark.parametrize("axis", ["x", 1, "z"])
def test_rdf_drop_axis_on_device_all_entry_points(axis, tmp_path):
    """2-D mode (structure.py:761-770) on the device: coordinate zeroed after the centre-of-mass
    stage, cell length along it = the largest one; batched in-memory path, file path and the
    frame-by-frame restatement agree bit for bit, counts and normalisation."""
    from trajfiles import write_amber_netcdf
    rng = np.random.default_rng(31)
    Lf = np.array([21.0, 26.5, 23.25], dtype=np.float32)
    F, N = 6, 1200
    frames = (rng.random((F, N, 3)) * Lf).astype(np.float32)
    dims = np.array([*Lf, 90, 90, 90], dtype=np.float32)
    k = ord(axis) - 120 if isinstance(axis, str) else axis
    masses = rng.uniform(1, 20, N)
    u = mdhelper_amd.ArrayUniverse(frames, dims, resids=np.arange(N) // 3, masses=masses)
    flat = frames.copy()
    flat[..., k] = 0
    box = dims.copy()
    box[k] = Lf.max()
    # atoms, self histogram
    r = RadialDistributionFunction(u.atoms, n_bins=40, range=(0.0, 9.0), drop_axis=axis, exclusion=(1, 1)).run()
    want = sum(orf.radial_histogram_ref(flat[f], flat[f], 40, (0.0, 9.0), box, exclusion=(1, 1)) for f in range(F))
    assert np.array_equal(r.results.counts, want)
    area = float(np.delete(Lf, k).prod())
    shell = np.pi * np.diff(r.results.edges ** 2)
    assert np.allclose(r.results.rdf, want / (F * shell * N * (N - 1) * F / (F * area)), rtol=1e-6)
    # residue centres of mass of one set against atoms of another: centres first, then the drop
    a, b = u.atoms[:600], u.atoms[600:]
    r2 = RadialDistributionFunction(a, b, n_bins=40, range=(0.5, 9.0), drop_axis=axis,
                                    groupings=("residues", "atoms")).run()
    m = masses[:600].reshape(200, 3)
    com = ((frames[:, :600].reshape(F, 200, 3, 3).astype(np.float64) * m[None, :, :, None]).sum(axis=2)
           / m.sum(axis=1)[None, :, None]).astype(np.float32)
    com[..., k] = 0
    want2 = sum(orf.radial_histogram_ref(com[f], flat[f, 600:], 40, (0.5, 9.0), box) for f in range(F))
    assert np.array_equal(r2.results.counts, want2)
    # trajectory file
    path = tmp_path / "flat.nc"
    write_amber_netcdf(path, frames, Lf, times=np.arange(F) * 1.0)
    uf = mdhelper_amd.FileUniverse(path, dt=1.0, resids=np.arange(N) // 3, masses=masses)
    r3 = RadialDistributionFunction(uf.atoms, n_bins=40, range=(0.0, 9.0), drop_axis=axis, exclusion=(1, 1)).run()
    assert np.array_equal(r3.results.counts, want)
    assert np.allclose(r3.results.rdf, r.results.rdf, rtol=1e-12)


class _PerFrameTrajectory:
    """Hides ``frame_block`` so that an analysis takes the generic frame-by-frame protocol."""

    def __init__(self, traj):
        self._traj = traj

    def __getattr__(self, name):
        if name in ("frame_block", "box_block", "native"):
            raise AttributeError(name)
        return getattr(self._traj, name)

    def __getitem__(self, item):
        return self._traj[item]

    def __len__(self):
        return len(self._traj)


@pytest.mark.parametrize("kind", ["memory", "netcdf"])
def test_structure_factor_and_isf_centres_of_mass_on_device(kind, tmp_path):
    """groupings="residues" / mixed for S(q) and the ISF: centres of mass formed on the
    device (mdx_sq_set_grouping, mdx_isf_set_grouping) equal the per-frame host path."""
    from mdhelper_amd.analysis import IntermediateScatteringFunction
    from trajfiles import write_amber_netcdf
    rng = np.random.default_rng(8)
    F, N, L = 9, 720, 18.0
    frames = (rng.random((F, N, 3)) * L).astype(np.float32)
    masses = rng.uniform(1, 30, N)
    resids = np.arange(N) // 3
    segids = np.arange(N) // 60
    kw = dict(masses=masses, resids=resids, segids=segids)
    if kind == "memory":
        u = mdhelper_amd.ArrayUniverse(frames, [L, L, L, 90, 90, 90], dt=0.5, **kw)
    else:
        path = tmp_path / "m.nc"
        write_amber_netcdf(path, frames, [L, L, L], times=np.arange(F) * 0.5)
        u = mdhelper_amd.FileUniverse(path, dt=0.5, **kw)
    slow_u = mdhelper_amd.ArrayUniverse(frames, [L, L, L, 90, 90, 90], dt=0.5, **kw)
    # shuffled membership: the molecules of a group are not contiguous in the file
    pick = rng.permutation(N // 3)[:150]
    sel_a = np.sort(np.concatenate([3 * pick, 3 * pick + 1, 3 * pick + 2]))
    rest = np.setdiff1d(np.arange(N), sel_a)
    for groupings in ("residues", ("residues", "atoms"), ("atoms", "residues")):
        def groups(univ):
            if groupings == ("atoms", "residues"):
                return univ.atoms[:360], univ.atoms[360:]
            return univ.atoms[sel_a], univ.atoms[rest]
        fast = StructureFactor(groups(u), groupings, mode="partial", n_points=4, verbose=False).run()
        slow = StructureFactor(groups(slow_u), groupings, mode="partial", n_points=4, verbose=False)
        slow._trajectory = _PerFrameTrajectory(slow_u.trajectory)
        slow.run()
        assert fast.results.ssf.shape == slow.results.ssf.shape
        assert np.allclose(fast.results.ssf, slow.results.ssf, rtol=1e-10, atol=1e-12), groupings
        assert np.abs(fast.results.ssf).max() > 0.1
        fi = IntermediateScatteringFunction(groups(u), groupings, mode="partial", n_points=3, n_lags=4,
                                            incoherent=True, verbose=False).run()
        si = IntermediateScatteringFunction(groups(slow_u), groupings, mode="partial", n_points=3, n_lags=4,
                                            incoherent=True, verbose=False)
        si._trajectory = _PerFrameTrajectory(slow_u.trajectory)
        si.run()
        assert np.allclose(fi.results.cisf, si.results.cisf, rtol=1e-10, atol=1e-12), groupings
        assert np.allclose(fi.results.iisf, si.results.iisf, rtol=1e-10, atol=1e-12), groupings


def test_structure_classes_on_frames_resident_in_hbm():
    """``ArrayUniverse.from_device``: float32 frames already in HBM are analysed where they lie (RDF of all
    particles, S(q) / ISF over groups that tile the particles in order); selections, float64 frames and scattered
    frame lists go through host memory.  Every route must give the in-memory universe's result."""
    from mdhelper_amd import _core
    from mdhelper_amd.analysis import IntermediateScatteringFunction
    rng = np.random.default_rng(23)
    F, N, L = 40, 1500, 19.0
    pos = np.mod(rng.uniform(0, L, (1, N, 3)) + np.cumsum(rng.normal(0, 0.2, (F, N, 3)), axis=0), L).astype(np.float32)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    um = mdhelper_amd.ArrayUniverse(pos, dims)
    d32, d64 = _core.DeviceArray.from_host(pos), _core.DeviceArray.from_host(pos.astype(np.float64))
    u32 = mdhelper_amd.ArrayUniverse.from_device(d32, dims)
    u64 = mdhelper_amd.ArrayUniverse.from_device(d64, dims)

    def rdf(u, **run):
        return RadialDistributionFunction(u.atoms, n_bins=60, range=(0.0, 8.0), exclusion=(1, 1),
                                          verbose=False).run(**run).results.counts

    want = rdf(um)
    assert np.array_equal(rdf(u32), want) and np.array_equal(rdf(u64), want)
    assert np.array_equal(rdf(u32, start=5, stop=31), rdf(um, start=5, stop=31))
    assert np.array_equal(rdf(u32, frames=np.arange(0, F, 3)), rdf(um, frames=np.arange(0, F, 3)))
    cross = lambda u: RadialDistributionFunction(u.atoms[:700], u.atoms[700:], n_bins=60, range=(0.0, 8.0),  # noqa: E731
                                                 verbose=False).run().results.counts
    assert np.array_equal(cross(u32), cross(um))

    def sq(u, **run):
        return StructureFactor((u.atoms[:600], u.atoms[600:]), mode="partial", n_points=5, verbose=False).run(**run)

    a, b, c = sq(um), sq(u32), sq(u64)
    assert np.allclose(b.results.ssf, a.results.ssf, rtol=1e-12, atol=1e-12 * np.abs(a.results.ssf).max())
    assert np.allclose(c.results.ssf, a.results.ssf, rtol=1e-12, atol=1e-12 * np.abs(a.results.ssf).max())
    a, b = sq(um, start=3, stop=33, step=2), sq(u32, start=3, stop=33, step=2)
    assert np.allclose(b.results.ssf, a.results.ssf, rtol=1e-12, atol=1e-12 * np.abs(a.results.ssf).max())

    def isf(u):
        return IntermediateScatteringFunction((u.atoms[:600], u.atoms[600:]), mode="partial", n_points=4, n_lags=6,
                                              incoherent=True, verbose=False).run()

    a, b = isf(um), isf(u32)
    for name in ("cisf", "iisf"):
        x, y = b.results[name], a.results[name]
        assert np.allclose(x, y, rtol=1e-12, atol=1e-12 * np.abs(y).max()), name
    d32.free()
    d64.free()
