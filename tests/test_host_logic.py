"""
Host-side logic that needs no GPU: the ArrayUniverse shim, frame selection,
constructor/argument error conventions of the reference, host post-processing
(radial Fourier transform, coordination numbers, transport-coefficient fits),
direct-definition correlations, frame-prep helpers.
"""
import warnings

import numpy as np
import pytest

import mdhelper_amd
from mdhelper_amd.algorithm import correlation, molecule, topology, utility
from mdhelper_amd.analysis import Onsager, RadialDistributionFunction, StructureFactor, structure, transport
from mdhelper_amd.analysis.base import Hash, SerialAnalysisBase
from mdhelper_amd.comm import shard_range
from oracle import correlation as oc


def _universe(F=6, N=40, L=12.0, **kw):
    rng = np.random.default_rng(0)
    return mdhelper_amd.ArrayUniverse((rng.random((F, N, 3)) * L).astype(np.float32),
                                      [L, L, L, 90, 90, 90], dt=0.5, **kw)


def test_universe_duck_type():
    u = _universe(charges=np.r_[np.ones(20), -np.ones(20)], resids=np.arange(40) // 4)
    ag = u.atoms[10:20]
    assert ag.universe is u and ag.n_atoms == 10 and ag.n_residues == 3
    assert ag.positions.dtype == np.float32 and ag.positions.shape == (10, 3)
    assert np.array_equal(ag.indices, np.arange(10, 20))
    ts = u.trajectory[3]
    assert ts.frame == 3 and ts.time == 1.5 and np.isclose(ts.volume, 12.0 ** 3)
    assert np.array_equal(ag.positions, u.trajectory._positions[3, 10:20])
    assert u.dimensions.shape == (6,) and len(u.trajectory[1:6:2]) == 3
    assert ag == u.atoms[10:20] and not (ag == u.atoms[11:21])
    assert len(u.atoms.residues) == 10 and np.allclose(u.atoms.residues.charges[:5], 4)


def test_frame_selection_and_hash():
    u = _universe()
    seen = []

    class Probe(SerialAnalysisBase):
        def _single_frame(self):
            seen.append((self._frame_index, self._ts.frame))

    p = Probe(u.trajectory).run(start=1, stop=6, step=2)
    assert seen == [(0, 1), (1, 3), (2, 5)] and p.n_frames == 3
    assert np.array_equal(p.frames, [1, 3, 5]) and np.allclose(p.times, [0.5, 1.5, 2.5])
    seen.clear()
    Probe(u.trajectory).run(frames=[0, 4])
    assert [s[1] for s in seen] == [0, 4]
    with pytest.raises(ValueError):
        Probe(u.trajectory).run(start=1, frames=[0])
    h = Hash({"a": 1}, b=2)
    h.c = 3
    assert h.a == 1 and h["c"] == 3 and h.missing is None and "b" in h
    with pytest.raises(TypeError):
        Hash(3)


def test_shard_range_is_a_partition():
    for n in (0, 1, 7, 100):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert max(hi - lo for lo, hi in parts) - min(hi - lo for lo, hi in parts) <= 1


def test_constructor_errors_follow_the_reference():
    u = _universe()
    with pytest.raises(ValueError):
        RadialDistributionFunction(u.atoms, groupings="molecules")
    with pytest.raises(ValueError):
        RadialDistributionFunction(u.atoms, drop_axis=5)
    with pytest.raises(ValueError):
        StructureFactor((u.atoms[:10],), mode=None)            # groups must cover the universe
    with pytest.raises(ValueError):
        StructureFactor((u.atoms[:10], u.atoms[10:20], u.atoms[20:]), mode="pair")
    with pytest.raises(ValueError):
        StructureFactor(u.atoms, dimensions=[1, 2])
    with pytest.raises(ValueError):
        StructureFactor(u.atoms, groupings=("atoms", "atoms"))
    with pytest.raises(ValueError):
        Onsager(u.atoms, groupings="chains", reduced=True)
    with pytest.raises(ValueError):
        Onsager((u.atoms[:20], u.atoms[20:]), charges=[1], reduced=True)
    with pytest.raises(ValueError):
        Onsager(u.atoms, dimensions=[1, 2], reduced=True)
    ons = Onsager(u.atoms, reduced=True, temperature=1.0)
    with pytest.raises(RuntimeError):
        ons.calculate_transport_coefficients()
    with pytest.raises(RuntimeError):
        ons.calculate_conductivity()
    no_box = mdhelper_amd.ArrayUniverse(np.zeros((2, 4, 3), dtype=np.float32))
    with pytest.raises(ValueError):
        RadialDistributionFunction(no_box.atoms)
    with pytest.raises(ValueError):
        Onsager(no_box.atoms, reduced=True)


def test_structure_factor_wavevectors():
    u = _universe()
    sf = StructureFactor(u.atoms, n_points=3)
    g1 = 2 * np.pi / 12.0
    assert sf._wavevectors.shape == (27, 3)
    assert np.allclose(sf._wavevectors[1], [0, 0, g1]) and np.allclose(sf._wavevectors[3], [g1, 0, 0])
    sf = StructureFactor(u.atoms, n_points=4, q_max=1.2 * g1)
    assert len(sf._wavevectors) == 4
    sf = StructureFactor(u.atoms, n_points=3, n_surfaces=2, n_surface_points=8)
    assert sf._wavevectors.shape == (27 + 16, 3)
    assert np.allclose(np.linalg.norm(sf._wavevectors[27:35], axis=1), g1)
    sf = StructureFactor(u.atoms, dimensions=[10.0, 12.0, 14.0], n_points=2)
    assert sf._wavevectors.shape == (8, 3)


def test_radial_fourier_transform_analytic():
    """exp(-ar)/r  <->  4 pi / (a^2 + q^2)   (reference tests/test_analysis_structure.py:42-53)."""
    alpha = 3.7
    r = np.linspace(1e-8, 20, 2_000)
    q = 1 / r
    f = np.exp(-alpha * r) / r
    assert np.allclose(4 * np.pi / (alpha ** 2 + q ** 2), structure.radial_fourier_transform(r, f, q), atol=4e-5)


def test_coordination_numbers_and_structure_factor_postprocessing():
    r = np.linspace(0.05, 12, 240)
    g = 1 + np.exp(-(r - 3) ** 2) * 1.5 - np.exp(-(r - 4.4) ** 2) * 0.6 + np.exp(-(r - 6) ** 2) * 0.4
    g[r < 2] = 0
    n = structure.calculate_coordination_numbers(r, g, 0.03, n_coord_nums=2)
    assert np.isfinite(n[0]) and n[0] > 0
    with pytest.raises(ValueError):
        structure.calculate_coordination_numbers(r, g, 0.03, n_dims=4)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert np.isnan(structure.calculate_coordination_numbers(r, np.ones_like(r), 0.03)).all()
    q, s = structure.calculate_structure_factor(r, g, True, 0.03, n_q=50)
    assert q.shape == s.shape == (50,)
    _, s_al = structure.calculate_structure_factor(r, g, False, 0.03, 0.5, 0.5, q, formalism="AL")
    assert np.allclose(s_al - 1, 0.5 * (s - 1))
    with pytest.raises(ValueError):
        structure.calculate_structure_factor(r, g, False, 0.03, 0.5, 0.5, q, formalism="XX")


def test_transport_coefficient_fits():
    t = 0.5 * np.arange(200)
    D = np.array([0.3, 0.7])
    msd_self = (D[:, None] * t)[:, None, :]
    msd_cross = np.stack([2.0 * t, 0.4 * t, 3.0 * t])[:, None, :]
    dims = np.array([10.0, 10.0, 10.0])
    for scale in ("linear", "log"):
        L, Ls, Di = transport.calculate_transport_coefficients(t, msd_cross, msd_self, (10, 20), dims, 2.0,
                                                               start=5, scale=scale)
        assert np.allclose(Di[0], D, rtol=1e-6)
        assert np.allclose(L[0], np.array([[2.0, 0.4], [0.4, 3.0]]) / 2000.0, rtol=1e-6)
        assert np.allclose(Ls[0], np.array([10, 20]) * D / 2000.0, rtol=1e-6)
    z = np.array([1.0, -1.0])
    assert np.allclose(transport.calculate_conductivity(L, z, reduced=True), (2.0 - 0.8 + 3.0) / 2000)
    tn = transport.calculate_transference_number(L, z)
    assert np.allclose(tn.sum(axis=-1), 1)
    mu = transport.calculate_electrophoretic_mobility(L, z, np.array([0.01, 0.02]), reduced=True)
    assert mu.shape == (1, 2)


def test_direct_correlations_match_oracle():
    rng = np.random.default_rng(2)
    a = rng.normal(size=(2, 30, 4, 3))
    b = rng.normal(size=(2, 30, 4, 3))
    assert np.allclose(correlation.correlation_shift(a, b, axis=1, vector=True),
                       oc.correlation_shift_ref(a, b, axis=1, vector=True))
    assert np.allclose(correlation.correlation_shift(a[0, :, 0, 0], double=True),
                       oc.correlation_shift_ref(a[0, :, 0, 0], double=True))
    assert np.allclose(correlation.msd_shift(a, axis=1, average=False), oc.msd_shift_ref(a, axis=1, average=False))
    assert np.allclose(correlation.msd_shift(a, b, axis=1), oc.msd_shift_ref(a, b, axis=1))
    traj_1 = np.array(((0, 0, 0), (1, 1, 1), (2, 2, 2), (3, 3, 3)))
    assert np.allclose(correlation.msd_shift(traj_1), [0, 3, 12, 27])


def test_correlation_argument_errors():
    """tests/test_algorithm_correlation.py:49-65, 162-178 (ACF), 238-260, 324-346 (CCF), 445-461, 504-520 (MSD) of
    the reference: every error case, raised before any GPU work."""
    for fn in (correlation.correlation_fft, correlation.correlation_shift):
        with pytest.raises(ValueError):
            fn(np.empty(0))
        with pytest.raises(ValueError):
            fn(np.empty((0, 3)))
        with pytest.raises(ValueError):
            fn(np.empty((2, 2, 2, 2, 2)))
        with pytest.raises(ValueError):
            fn(np.empty((2, 2, 2)), axis=2)
        # cross-correlations (:238-260, :324-346)
        with pytest.raises(ValueError):
            fn(np.empty(0), np.empty(0))
        with pytest.raises(ValueError):
            fn(np.empty((0, 3)), np.empty((0, 3)))
        with pytest.raises(ValueError):
            fn(np.empty((2, 2, 2, 2, 2)), np.empty((2, 2, 2, 2, 2)))
        with pytest.raises(ValueError):
            fn(np.empty((2, 3)), np.empty((3, 2)))
        with pytest.raises(ValueError):
            fn(np.empty((2, 2, 2)), np.empty((2, 2, 2)), axis=2)
    for fn in (correlation.msd_fft, correlation.msd_shift):
        with pytest.raises(ValueError):
            fn(np.empty(0))
        with pytest.raises(ValueError):
            fn(np.empty((2, 2, 2, 2, 2)))
        with pytest.raises(ValueError):
            fn(np.ones((4, 3)), np.ones((1, 3)))
        with pytest.raises(ValueError):
            fn(np.empty((2, 2, 2)), axis=2)
    with pytest.warns(UserWarning):
        correlation.correlation_shift(np.ones((4, 2)))


def test_frame_prep_helpers():
    rng = np.random.default_rng(3)
    L = np.array([10.0, 10.0, 10.0])
    true = np.cumsum(rng.normal(scale=1.5, size=(50, 8, 3)), axis=0) + 5
    wrapped = true - np.floor(true / L) * L
    old = wrapped[0].copy()
    images = np.zeros((8, 3), dtype=int)
    out = []
    for f in range(50):
        p = wrapped[f].copy()
        topology.unwrap(p, old, L, thresholds=L / 2, images=images)
        out.append(p)
    assert np.allclose(np.array(out) - out[0], true - true[0])
    w = topology.wrap(true[10], L, in_place=False)
    assert w.min() >= 0 and w.max() <= 10
    u = _universe(masses=np.arange(1, 41, dtype=float), resids=np.arange(40) // 4)
    com = molecule.center_of_mass(u.atoms, "residues")
    m = np.arange(1, 41, dtype=float).reshape(10, 4)
    want = (m[..., None] * u.atoms.positions.reshape(10, 4, 3)).sum(1) / m.sum(1, keepdims=True)
    assert np.allclose(com, want)
    assert np.allclose(molecule.center_of_mass(positions=u.atoms.positions, masses=u.atoms.masses),
                       molecule.center_of_mass(u.atoms))
    # (the three cases of the reference's tests/test_algorithm_utility.py:13-25)
    assert np.allclose(utility.get_closest_factors(1000, 3), 10 * np.ones(3, dtype=int))
    assert utility.get_closest_factors(35904, 3).tolist() == [32, 33, 34]
    assert utility.get_closest_factors(73440, 4, reverse=True).tolist() == [18, 17, 16, 15]


def test_bench_contract_without_a_device(tmp_path):
    """bench.py: flags of the driver's contract parse, more ranks than visible devices are refused
    with the device count, and without a HIP device the run fails loudly (no CPU path)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--help"], capture_output=True,
                         text=True, env=env, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--blocks", "--host-path", "--traj-file"):
        assert flag in out.stdout
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True,
                         text=True, env=env, timeout=120)
    assert out.returncode != 0 and "HIP device" in out.stderr and out.stdout.strip() == ""
    from mdhelper_amd import _lib
    if _lib.device_count() == 0:
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0"],
                             capture_output=True, text=True, env=env, timeout=300)
        assert out.returncode != 0 and out.stdout.strip() == ""      # no JSON line without a GPU


def test_fragments_and_make_whole_images():
    """`ArrayUniverse(..., bonds=...)` -> `atoms.fragments` (connected components, ordered by first atom) and
    the image flags that make every fragment of the current frame whole (reference transport.py:936-941 ->
    MDAnalysis make_whole, restated): a chain wrapped twice around the cell, a ring, a lone atom."""
    import mdhelper_amd
    from mdhelper_amd.algorithm.topology import make_whole_images
    L = np.array([10.0, 12.0, 9.0])
    # chain of 9 atoms, 1.5 apart along x starting at 8.5: crosses the x boundary once; continues wrapping
    chain = np.stack((8.5 + 1.5 * np.arange(9), np.full(9, 6.0), np.full(9, 4.5)), axis=1)
    # ring of 4 atoms straddling the y and z boundaries
    ring = np.array([[2.0, 11.6, 8.7], [2.0, 0.4, 8.7], [2.0, 0.4, 0.3], [2.0, 11.6, 0.3]])
    ring_true = np.array([[2.0, 11.6, 8.7], [2.0, 12.4, 8.7], [2.0, 12.4, 9.3], [2.0, 11.6, 9.3]])
    lone = np.array([[5.0, 5.0, 5.0]])
    true = np.concatenate((chain, ring_true, lone))
    wrapped = np.mod(true, L).astype(np.float32)
    bonds = [(i, i + 1) for i in range(8)] + [(9, 10), (10, 11), (11, 12), (12, 9)]
    # atoms shuffled so that fragments interleave in index order
    order = np.array([0, 9, 1, 10, 13, 2, 11, 3, 12, 4, 5, 6, 7, 8])
    inv = np.argsort(order)
    u = mdhelper_amd.ArrayUniverse(wrapped[order][None], [*L, 90, 90, 90],
                                   bonds=[(inv[i], inv[j]) for i, j in bonds])
    frags = u.atoms.fragments
    assert [len(f) for f in frags] == [9, 4, 1]                     # ordered by first atom: chain, ring, lone
    assert sorted(order[frags[1].indices]) == [9, 10, 11, 12]
    assert [len(f) for f in u.atoms[[4]].fragments] == [1]          # fragments of a sub-group: whole fragments
    img = make_whole_images(u, L)
    whole = wrapped[order].astype(float) + img * L
    # whole up to one lattice vector per fragment (the first atom keeps its stored image)
    for f in frags:
        d = whole[f.indices] - true[order][f.indices]
        shift = d[0]
        assert np.allclose(d, shift, atol=1e-5)
        assert np.allclose(shift / L, np.rint(shift / L), atol=1e-5)
    assert img[inv[0]].tolist() == [0, 0, 0] and img[inv[8]].tolist() == [2, 0, 0]
    # no bonds: nothing to make whole
    u0 = mdhelper_amd.ArrayUniverse(wrapped[None], [*L, 90, 90, 90])
    assert not make_whole_images(u0, L).any() and len(u0.atoms.fragments) == 14


def test_equal_wavenumber_means_follow_the_reference_loop():
    """structure.py:1536-1541 / 2116-2127: columns whose wavenumber is numpy.isclose to a unique one are averaged.
    The bisection + segmented-sum form must give the reference loop's result on grids (every column belongs to one
    unique wavenumber) and fall back to the loop itself where uniques lie within isclose's tolerance of each other."""
    from mdhelper_amd.analysis.structure import _mean_over_equal_wavenumbers as fast
    rng = np.random.default_rng(4)

    def loop(x, w, u):
        return np.stack([x[..., np.isclose(q, w)].mean(axis=-1) for q in u], axis=-1)

    for n, L in ((6, 31.0), (9, 68.94)):
        grid = 2 * np.pi * np.arange(n) / L
        w = np.linalg.norm(np.stack(np.meshgrid(grid, grid, grid), -1).reshape(-1, 3), axis=1)
        u = np.unique(w.round(11))
        for shape in ((3, len(w)), (4, 2, len(w))):
            x = rng.normal(size=shape)
            assert np.allclose(fast(x, w, u), loop(x, w, u), rtol=1e-13, atol=1e-15)
    # non-cubic cell, arbitrary wavevectors, unsorted uniques
    w = np.linalg.norm(rng.normal(size=(300, 3)), axis=1)
    w = np.concatenate((w, w[:40]))
    u = np.unique(w.round(11))[::-1].copy()
    x = rng.normal(size=(2, len(w)))
    assert np.allclose(fast(x, w, u), loop(x, w, u), rtol=1e-13, atol=1e-15)
    # uniques closer than isclose's tolerance: both columns count for both (the reference's behaviour)
    w = np.array([1.0, 1.0 + 2e-6, 2.0, 2.0])
    u = np.unique(w.round(11))
    x = np.arange(8.0).reshape(2, 4)
    assert np.array_equal(fast(x, w, u), loop(x, w, u))


def test_center_of_mass_follows_the_reference_tests():
    """The reference's own test_algorithm_molecule.py (its AdK universe comes from MDAnalysisTests, so the cases are
    restated on a synthetic universe): the four error cases of `test_center_of_mass_errors` (:19-40) and the
    definitional cases 2-6, 8-11 of `test_center_of_mass_mda` (:42-122) against sum(m x) / sum(m)."""
    import mdhelper_amd as mdx
    rng = np.random.default_rng(8)
    N, R = 40, 10
    pos = rng.uniform(0, 20, (2, N, 3)).astype(np.float32)
    masses = rng.uniform(1, 16, N)
    u = mdx.ArrayUniverse(pos, [20, 20, 20, 90, 90, 90], masses=masses, resids=np.arange(N) // 4,
                          segids=np.zeros(N, dtype=int))
    # errors (:19-40)
    with pytest.raises(ValueError):
        molecule.center_of_mass()
    with pytest.raises(ValueError):
        molecule.center_of_mass(u.atoms, "atoms")
    no_box = mdx.ArrayUniverse(pos, None, masses=masses)
    with pytest.raises(ValueError):
        molecule.center_of_mass(no_box.atoms, images=np.zeros((N, 3)))
    with pytest.raises(ValueError):
        molecule.center_of_mass(masses=u.atoms.masses,
                                positions=[u.atoms.positions[4 * i:4 * i + 4] for i in range(R)])
    # values (:52-122)
    x = np.asarray(u.atoms.positions, dtype=float)
    com = (masses[:, None] * x).sum(0) / masses.sum()
    assert np.allclose(molecule.center_of_mass(u.atoms), com)
    c, m, p = molecule.center_of_mass(u.atoms, images=np.zeros((N, 3), dtype=int), dimensions=np.array((0, 0, 0)),
                                      raw=True)
    assert np.allclose(c, com) and np.allclose(m, masses) and np.allclose(p, x)
    res = (masses.reshape(R, 4)[..., None] * x.reshape(R, 4, 3)).sum(1) / masses.reshape(R, 4).sum(1, keepdims=True)
    assert np.allclose(molecule.center_of_mass(u.atoms, "residues"), res)
    assert np.allclose(molecule.center_of_mass(masses=[masses[4 * i:4 * i + 4] for i in range(R)],
                                               positions=[x[4 * i:4 * i + 4] for i in range(R)]), res)
    assert np.allclose(molecule.center_of_mass(u.atoms, n_groups=R), res)
    assert np.allclose(molecule.center_of_mass(masses=masses, positions=x, n_groups=R), res)
    assert np.allclose(molecule.center_of_mass(u.atoms, "segments"), com)
