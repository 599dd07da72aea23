"""
End-to-end vector ACF (reference analysis/polymer.py:510-803): the CPU restatement against
closed forms, and the host side of ``mdhelper_amd.analysis.polymer`` (``fft=False`` needs no
GPU: direct sliding-window correlation in NumPy) against the restatement.
"""
import warnings

import numpy as np
import pytest

import mdhelper_amd
from mdhelper_amd.algorithm import topology
from mdhelper_amd.analysis import polymer
from oracle import polymer as op


def _rotors(T=64, M=6, omega=0.11, seed=3):
    """Dumbbells rotating rigidly in random planes: C_ee(m) = cos(omega m) exactly."""
    rng = np.random.default_rng(seed)
    a = rng.normal(size=(M, 3))
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = np.cross(a, rng.normal(size=(M, 3)))
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    t = np.arange(T)[:, None, None]
    bond = 1.5 * (np.cos(omega * t) * a + np.sin(omega * t) * b)           # [T, M, 3]
    centre = rng.uniform(3, 9, (1, M, 3)) + 0.01 * np.cumsum(rng.normal(size=(T, M, 3)), axis=0)
    pos = np.empty((T, 2 * M, 3))
    pos[:, 0::2] = centre - bond / 2
    pos[:, 1::2] = centre + bond / 2
    return pos


def _chains(T=40, M=5, n=7, L=9.0, seed=5, per_monomer=1):
    """Random-walk chains diffusing in an unbounded space (bond length ~1)."""
    rng = np.random.default_rng(seed)
    start = rng.uniform(0, L, (1, M, 1, 3))
    steps = rng.normal(scale=0.55, size=(T, M, n * per_monomer, 3))
    steps[1:] *= 0.08
    conf = np.cumsum(steps, axis=2)
    conf = conf[:1] + np.cumsum(np.r_[np.zeros_like(conf[:1]), conf[1:] - conf[1:]], axis=0) \
        + np.cumsum(np.r_[np.zeros((1, M, n * per_monomer, 3)), steps[1:]], axis=0)
    drift = np.cumsum(rng.normal(scale=0.35, size=(T, M, 1, 3)), axis=0)
    return (start + conf + drift).reshape(T, M * n * per_monomer, 3)


def test_restatement_closed_form_rigid_rotors():
    pos = _rotors()
    acf, e2e = op.end_to_end_run_ref(pos, [np.arange(12)], [6], [2], ["atoms"])
    assert np.allclose(np.linalg.norm(e2e, axis=-1), 1.5)
    assert np.allclose(acf[0, 0], np.cos(0.11 * np.arange(64)), atol=1e-12)
    direct, _ = op.end_to_end_run_ref(pos, [np.arange(12)], [6], [2], ["atoms"], fft=False)
    assert np.allclose(direct, acf, atol=1e-12)
    two, _ = op.end_to_end_run_ref(pos, [np.arange(12)], [6], [2], ["atoms"], n_blocks=2)
    assert two.shape == (1, 2, 32) and np.allclose(two[0, 1], np.cos(0.11 * np.arange(32)), atol=1e-12)


def test_relaxation_time_of_an_exponential():
    t = 0.5 * np.arange(200)
    tau = op.relaxation_time_ref(t, np.exp(-t / 7.0))
    assert np.isclose(tau, 7.0, rtol=1e-6)
    assert np.isclose(polymer.calculate_relaxation_time(t, np.exp(-t / 7.0)), tau, rtol=1e-9)
    stretched = np.exp(-(t / 5.0) ** 0.6)
    from scipy import special
    assert np.isclose(polymer.calculate_relaxation_time(t, stretched), 5.0 * special.gamma(1 + 1 / 0.6), rtol=1e-5)


def test_unwrap_edge_makes_chains_whole():
    L = np.array([9.0, 8.0, 10.0])
    whole = _chains(T=1, L=9.0)[0]
    masses = np.random.default_rng(1).uniform(1, 20, len(whole))
    wrapped = np.mod(whole, L)
    bonds = np.array([(c * 7 + j, c * 7 + j + 1) for c in range(5) for j in range(6)])
    got = topology.unwrap_edge(positions=wrapped.copy(), bonds=bonds, dimensions=L, masses=masses)
    ref = op.unwrap_edge_chain_ref(wrapped, 5, L, masses)
    assert np.allclose(got, ref, atol=1e-12)
    # bond vectors are those of the unwrapped chains, centres of mass lie in the cell
    for c in range(5):
        sl = slice(c * 7, (c + 1) * 7)
        assert np.allclose(np.diff(got[sl], axis=0), np.diff(whole[sl], axis=0), atol=1e-9)
        com = (masses[sl, None] * got[sl]).sum(0) / masses[sl].sum()
        assert np.all(com >= 0) and np.all(com <= L)
    # shuffled bond order and a triclinic cell with right angles written out give the same answer
    again = topology.unwrap_edge(positions=wrapped.copy(), bonds=bonds[::-1], dimensions=[*L, 90, 90, 90],
                                 masses=masses)
    assert np.allclose(again, got, atol=1e-12)
    with pytest.raises(ValueError):
        topology.unwrap_edge(positions=wrapped.copy(), dimensions=L)
    with pytest.raises(ValueError):
        topology.unwrap_edge(positions=wrapped.copy(), bonds=bonds)
    with pytest.warns(UserWarning):
        topology.unwrap_edge(positions=wrapped.copy(), bonds=bonds, dimensions=L)


def test_triclinic_minimum_image_is_the_shortest():
    rng = np.random.default_rng(9)
    dims = np.array([9.0, 8.0, 10.0, 70.0, 100.0, 60.0])
    v = rng.uniform(-30, 30, (200, 3))
    got = topology._minimum_image(v, dims)
    al, be, ga = np.deg2rad(dims[3:])
    a = np.array([9.0, 0, 0])
    b = 8.0 * np.array([np.cos(ga), np.sin(ga), 0])
    cx, cy = np.cos(be), (np.cos(al) - np.cos(be) * np.cos(ga)) / np.sin(ga)
    c = 10.0 * np.array([cx, cy, np.sqrt(1 - cx * cx - cy * cy)])
    n = np.arange(-6, 7)
    shifts = (n[:, None, None, None] * a + n[None, :, None, None] * b + n[None, None, :, None] * c).reshape(-1, 3)
    best = np.min(np.linalg.norm(v[:, None] + shifts[None], axis=-1), axis=1)
    assert np.allclose(np.linalg.norm(got, axis=-1), best, atol=1e-9)


@pytest.mark.parametrize("case", ["atoms", "residues explicit", "residues internal", "two groups, blocks"])
def test_host_path_equals_restatement(case):
    """``fft=False``: everything but the device ACF — end-monomer selection, centres of mass,
    frame gathering, block split — against the frame-by-frame restatement."""
    if case == "atoms":
        pos = _chains()
        u = mdhelper_amd.ArrayUniverse(pos.astype(np.float32), [30, 30, 30, 90, 90, 90], dt=0.5)
        a = polymer.EndToEndVector(u.atoms, n_chains=5, n_monomers=7, fft=False, verbose=False).run()
        ref, _ = op.end_to_end_run_ref(pos.astype(np.float32), [np.arange(35)], [5], [7], ["atoms"], fft=False)
    elif case == "residues explicit":
        pos = _chains(per_monomer=3)
        masses = np.random.default_rng(2).uniform(1, 16, pos.shape[1])
        u = mdhelper_amd.ArrayUniverse(pos.astype(np.float32), [30, 30, 30, 90, 90, 90], dt=0.5, masses=masses)
        a = polymer.EndToEndVector(u.atoms, "residues", n_chains=5, n_monomers=7, fft=False, verbose=False).run()
        ref, _ = op.end_to_end_run_ref(pos.astype(np.float32), [np.arange(105)], [5], [7], ["residues"],
                                       masses=masses, fft=False)
    elif case == "residues internal":
        pos = _chains(per_monomer=3)
        masses = np.random.default_rng(2).uniform(1, 16, pos.shape[1])
        u = mdhelper_amd.ArrayUniverse(pos.astype(np.float32), [30, 30, 30, 90, 90, 90], dt=0.5, masses=masses,
                                       resids=np.arange(105) // 3, segids=np.arange(105) // 21)
        a = polymer.EndToEndVector(u.atoms, "residues", fft=False, verbose=False).run()
        ref, _ = op.end_to_end_run_ref(pos.astype(np.float32), [np.arange(105)], [5], [7], ["residues"],
                                       masses=masses, fft=False)
        assert a._internal and a._n_chains[0] == 5
    else:
        pos = _chains(T=41)
        u = mdhelper_amd.ArrayUniverse(pos.astype(np.float32), [30, 30, 30, 90, 90, 90], dt=0.5)
        with pytest.warns(UserWarning, match="not divisible"):
            a = polymer.EndToEndVector([u.atoms[:14], u.atoms[14:]], n_chains=(2, 3), n_monomers=(7, 7),
                                       n_blocks=2, fft=False, verbose=False).run()
        ref, _ = op.end_to_end_run_ref(pos.astype(np.float32), [np.arange(14), np.arange(14, 35)], [2, 3],
                                       [7, 7], ["atoms", "atoms"], n_blocks=2, fft=False)
        assert a.results.acf.shape == (2, 2, 20)
    assert np.allclose(a.results.acf, ref, rtol=1e-10, atol=1e-12)
    assert np.allclose(a.results.acf[..., 0], 1.0)
    assert np.allclose(a.results.times, 0.5 * np.arange(a.results.acf.shape[-1]))
    # the generic per-frame protocol gives the same numbers as the gathered path
    class Plain:
        def __init__(self, t): self._t = t
        def __getattr__(self, k):
            if k == "frame_block":
                raise AttributeError(k)
            return getattr(self._t, k)
        def __getitem__(self, i): return self._t[i]
        def __len__(self): return len(self._t)
    b_kwargs = dict(fft=False, verbose=False)
    if case == "atoms":
        b = polymer.EndToEndVector(u.atoms, n_chains=5, n_monomers=7, **b_kwargs)
        b._trajectory = Plain(u.trajectory)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            b.run()
        assert np.allclose(b.results.acf, a.results.acf, rtol=1e-12, atol=1e-14)


def test_unwrap_follows_the_ends_across_the_boundaries():
    pos = _chains(T=60, L=9.0)
    L = np.array([9.0, 9.0, 9.0])
    wrapped = np.mod(pos, L).astype(np.float32)
    u = mdhelper_amd.ArrayUniverse(wrapped, [*L, 90, 90, 90], dt=1.0)
    a = polymer.EndToEndVector(u.atoms, n_chains=5, n_monomers=7, unwrap=True, fft=False, verbose=False).run()
    ref, e2e = op.end_to_end_run_ref(wrapped, [np.arange(35)], [5], [7], ["atoms"], dimensions=L, unwrap=True,
                                     fft=False)
    assert np.allclose(a.results.acf, ref, rtol=1e-10, atol=1e-12)
    # ... and recovers the end-to-end vectors of the unwrapped chains
    true = pos.reshape(60, 5, 7, 3)
    assert np.allclose(e2e, true[:, :, -1] - true[:, :, 0], atol=1e-4)
    assert np.allclose(a._e2e, e2e, atol=1e-12)


def test_constructor_errors_follow_the_reference():
    u = mdhelper_amd.ArrayUniverse(np.zeros((4, 12, 3), np.float32), [5, 5, 5, 90, 90, 90])
    with pytest.raises(ValueError, match="Invalid grouping"):
        polymer.EndToEndVector(u.atoms, "segments", n_chains=2, n_monomers=6)
    with pytest.raises(ValueError, match="number of grouping values"):
        polymer.EndToEndVector(u.atoms, ("atoms", "atoms"), n_chains=2, n_monomers=6)
    with pytest.raises(ValueError, match="polymer counts"):
        polymer.EndToEndVector(u.atoms, n_chains=(2, 2), n_monomers=6)
    with pytest.raises(ValueError, match="chain lengths"):
        polymer.EndToEndVector(u.atoms, n_chains=2, n_monomers=(6, 6))
    e = polymer.EndToEndVector(u.atoms, n_chains=2, n_monomers=6, parallel=True)
    with pytest.raises(RuntimeError, match="Call EndToEndVector.run"):
        e.calculate_relaxation_time()
    with pytest.raises(ValueError, match="cannot be divided"):
        polymer.EndToEndVector(u.atoms, n_chains=5, n_monomers=2, fft=False).run()
