"""
Trajectory files on the GPU path: raw frames file -> pinned -> HBM -> unpack kernel
(byte swap / plane transpose / gather) must reproduce the host reader bit for bit, and an
analysis fed from a FileUniverse must give the counts of the same frames held in memory.
"""

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import mdhelper_amd  # noqa: E402
from mdhelper_amd import _core  # noqa: E402
from mdhelper_amd.analysis import RadialDistributionFunction  # noqa: E402
from mdhelper_amd.io import TrajectoryFile  # noqa: E402
from oracle import rdf as orf  # noqa: E402
from trajfiles import per_frame, write_amber_netcdf, write_dcd  # noqa: E402


def _frames(F, N, L, seed):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(0, L, (1, N, 3)) + np.cumsum(rng.normal(0, 0.3, (F, N, 3)), axis=0)
    return np.mod(pos, L).astype(np.float32)


@pytest.mark.parametrize("kind", ["netcdf", "dcd", "dcd_be"])
def test_load_device_equals_host_reader(tmp_path, kind):
    # 52 MB of coordinates: more than one 32 MiB pinned chunk, buffers are reused
    F, N, L = 220, 20011, 60.0
    pos = _frames(F, N, L, 11)
    path = tmp_path / ("t.nc" if kind == "netcdf" else "t.dcd")
    if kind == "netcdf":
        write_amber_netcdf(path, pos, (L, L, L))
    else:
        write_dcd(path, pos, [[L, L, L, 90, 90, 90]], big_endian=(kind == "dcd_be"))
    t = TrajectoryFile(path)
    frames = np.arange(F - 1, -1, -1)            # reversed: arbitrary frame lists
    out = _core.DeviceArray((F, N, 3), np.float32)
    t.load_device(frames, out.ptr)
    assert np.array_equal(out.to_host(), pos[::-1])
    # gathered selection
    idx = np.random.default_rng(3).permutation(N)[:777].astype(np.int32)
    d_idx = _core.DeviceArray.from_host(idx)
    sel = _core.DeviceArray((9, 777, 3), np.float32)
    t.load_device(np.arange(3, 12), sel.ptr, d_index=d_idx.ptr, n_sel=777)
    assert np.array_equal(sel.to_host(), pos[3:12][:, idx])
    for a in (out, sel, d_idx):
        a.free()
    t.close()


@pytest.mark.parametrize("kind", ["netcdf", "dcd"])
def test_rdf_from_file_universe_bit_exact(tmp_path, kind):
    F, N, L = 24, 3000, 31.0
    pos = _frames(F, N, L, 12)
    lengths = np.array([[L + 0.01 * f, L, L - 0.02 * f] for f in range(F)], dtype=np.float32)
    path = tmp_path / ("r.nc" if kind == "netcdf" else "r.dcd")
    if kind == "netcdf":
        write_amber_netcdf(path, pos, lengths)
    else:
        write_dcd(path, pos, np.hstack([lengths, np.full((F, 3), 90.0)]), cosines=True)
    dims = np.hstack([lengths, np.full((F, 3), 90.0, dtype=np.float32)])
    uf = mdhelper_amd.FileUniverse(path)
    um = mdhelper_amd.ArrayUniverse(pos, dims)
    kw = dict(n_bins=150, range=(0.0, 12.0), exclusion=(1, 1))
    a = RadialDistributionFunction(uf.atoms, **kw).run()
    b = RadialDistributionFunction(um.atoms, **kw).run()
    ref = orf.rdf_run_ref(pos, dims, 150, (0.0, 12.0), exclusion=(1, 1))
    assert np.array_equal(a.results.counts, ref["counts"])
    assert np.array_equal(a.results.counts, b.results.counts)
    assert np.allclose(a.results.rdf, b.results.rdf, rtol=1e-12)
    # strided frames, two different selections (cations vs anions)
    g1, g2 = np.arange(0, N, 3), np.arange(1, N, 2)
    a = RadialDistributionFunction(uf.select(g1), uf.select(g2), n_bins=90, range=(1.0, 10.0)).run(step=5)
    b = RadialDistributionFunction(um.select(g1), um.select(g2), n_bins=90, range=(1.0, 10.0)).run(step=5)
    assert a.n_frames == 5 and np.array_equal(a.results.counts, b.results.counts)
    assert a.results.counts.sum() > 0
    # one group against all particles
    a = RadialDistributionFunction(uf.select(g1), uf.atoms, n_bins=64, range=(0.0, 8.0)).run(frames=[2, 7])
    b = RadialDistributionFunction(um.select(g1), um.atoms, n_bins=64, range=(0.0, 8.0)).run(frames=[2, 7])
    assert np.array_equal(a.results.counts, b.results.counts)


def test_rdf_host_and_file_pipelines_through_ramped_slabs(tmp_path, monkeypatch):
    """The staging pipeline of mdx_rdf_accumulate / mdx_rdf_accumulate_traj with slabs of 1 MiB: 300 frames of
    1 100 atoms go through slabs of 16, 32, 72, 72, 72 and 36 frames (the first two a quarter and a half of a slab:
    StagePipeline::run's ramp), alternating between the two staging sets; counts equal those of plain calls."""
    monkeypatch.setenv("MDX_RDF_PIPE_MB", "1")
    F, N, L = 300, 1100, 22.0
    pos = _frames(F, N, L, 31)
    lengths = np.full((F, 3), L, dtype=np.float32)
    dims = np.hstack([lengths, np.full((F, 3), 90.0, dtype=np.float32)])
    edges = np.linspace(0.0, 9.0, 91)
    eng = _core.RdfEngine(edges, (1, 1))
    eng.accumulate(pos[:7], None, dims[:7])            # one small call first: the ring and the sets exist
    eng.reset()
    eng.accumulate(pos, None, dims)                    # host pipeline, six slabs
    host_counts = eng.counts()
    eng.close()
    monkeypatch.delenv("MDX_RDF_PIPE_MB")
    eng = _core.RdfEngine(edges, (1, 1))
    for f0 in range(0, F, 100):                        # three plain calls (one slab each)
        eng.accumulate(pos[f0:f0 + 100], None, dims[f0:f0 + 100])
    want = eng.counts()
    eng.close()
    assert want.sum() > 0 and np.array_equal(host_counts, want)
    monkeypatch.setenv("MDX_RDF_PIPE_MB", "1")
    path = tmp_path / "ramp.nc"
    write_amber_netcdf(path, pos, lengths)
    uf = mdhelper_amd.FileUniverse(path)
    got = RadialDistributionFunction(uf.atoms, n_bins=90, range=(0.0, 9.0), exclusion=(1, 1)).run()
    assert np.array_equal(got.results.counts, want)


def test_structure_factor_and_isf_from_file_universe(tmp_path):
    from mdhelper_amd.analysis import IntermediateScatteringFunction, StructureFactor
    from oracle import fourier as of
    F, N, L = 12, 1500, 25.0
    pos = _frames(F, N, L, 21)
    write_amber_netcdf(tmp_path / "s.nc", pos, (L, L, L))
    write_dcd(tmp_path / "s.dcd", pos, [[L, L, L, 90, 90, 90]])
    um = mdhelper_amd.ArrayUniverse(pos, [L, L, L, 90, 90, 90])
    g1, g2 = np.arange(1, N, 2), np.arange(0, N, 2)          # interleaved groups: a real gather
    cat = np.concatenate([g1, g2])
    ref = of.ssf_run_ref(pos[:, cat].astype(np.float64), [len(g1), len(g2)],
                         of.grid_wavevectors([L, L, L], 4), mode="partial")
    for name in ("s.nc", "s.dcd"):
        uf = mdhelper_amd.FileUniverse(tmp_path / name)
        a = StructureFactor([uf.select(g1), uf.select(g2)], mode="partial", n_points=4).run()
        b = StructureFactor([um.select(g1), um.select(g2)], mode="partial", n_points=4).run()
        assert np.allclose(a.results.ssf, ref["ssf"], rtol=1e-6, atol=1e-9)
        assert np.allclose(a.results.ssf, b.results.ssf, rtol=1e-12, atol=1e-12)
        a = StructureFactor(uf.atoms, n_points=3).run(start=2, step=3)
        b = StructureFactor(um.atoms, n_points=3).run(start=2, step=3)
        assert a.n_frames == 4 and np.allclose(a.results.ssf, b.results.ssf, rtol=1e-12, atol=1e-12)
    uf = mdhelper_amd.FileUniverse(tmp_path / "s.nc", dt=2.0)
    kw = dict(mode="partial", n_points=3, n_lags=5, incoherent=True)
    a = IntermediateScatteringFunction([uf.select(g1), uf.select(g2)], **kw).run()
    b = IntermediateScatteringFunction([um.select(g1), um.select(g2)], dt=2.0, **kw).run()
    ref = of.isf_run_ref(pos[:, cat], [len(g1), len(g2)], of.grid_wavevectors([L, L, L], 3), 5,
                         mode="partial", incoherent=True)
    assert np.allclose(a.results.cisf, ref["cisf"], rtol=1e-6, atol=1e-9)
    assert np.allclose(a.results.iisf, ref["iisf"], rtol=1e-6, atol=1e-9)
    assert np.allclose(a.results.cisf, b.results.cisf, rtol=1e-12, atol=1e-12)
    assert np.array_equal(a.results.times, b.results.times)


@pytest.mark.parametrize("kind", ["netcdf", "dcd"])
def test_onsager_from_file_with_device_unwrap(tmp_path, kind):
    """Wrapped trajectory on disk -> device unwrap + MSD equals the host-unwrapped analysis."""
    import warnings
    from mdhelper_amd.analysis import Onsager
    rng = np.random.default_rng(33)
    T, N, L = 400, 96, np.array([11.0, 12.5, 10.25])
    walk = rng.uniform(0, L, (1, N, 3)) + np.cumsum(rng.normal(0, 0.45, (T, N, 3)), axis=0)
    wrapped = np.mod(walk, L).astype(np.float32)          # crosses the box many times
    path = tmp_path / ("w.nc" if kind == "netcdf" else "w.dcd")
    if kind == "netcdf":
        write_amber_netcdf(path, wrapped, L, times=np.arange(T) * 0.5)
    else:
        write_dcd(path, wrapped, [[*L, 90, 90, 90]])
    dims = np.array([*L, 90, 90, 90], dtype=np.float32)
    charges = np.where(np.arange(N) % 2 == 0, 1.0, -1.0)
    uf = mdhelper_amd.FileUniverse(path, dt=0.5, charges=charges)
    um = mdhelper_amd.ArrayUniverse(wrapped, dims, dt=0.5, charges=charges)
    cat, an = np.arange(0, N, 2), np.arange(1, N, 2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = Onsager([uf.select(cat), uf.select(an)], temperature=300, n_blocks=3, unwrap=True,
                    verbose=False).run()
        b = per_frame(Onsager([um.select(cat), um.select(an)], temperature=300, n_blocks=3, unwrap=True,
                              verbose=False)).run()
        c = Onsager([um.select(cat), um.select(an)], temperature=300, n_blocks=3, unwrap=True,
                    verbose=False).run()          # in-memory float32 frames, same device stages
    assert a._from_file and c._from_file and not b._from_file
    assert np.allclose(c.results.msd_self, b.results.msd_self, rtol=1e-9,
                       atol=1e-9 * np.abs(b.results.msd_self).max())
    assert a.results.msd_self.shape == (2, 3, 133)
    scale = np.abs(b.results.msd_self).max()
    assert np.allclose(a.results.msd_self, b.results.msd_self, rtol=1e-9, atol=1e-9 * scale)
    assert np.allclose(a.results.msd_cross, b.results.msd_cross, rtol=1e-9,
                       atol=1e-9 * np.abs(b.results.msd_cross).max())
    # the walk really is diffusive after unwrapping: MSD(m) ~ 3 sigma^2 m / (2 D) with D = 3
    m = np.arange(1, 40)
    assert np.allclose(a.results.msd_self[:, :, 1:40].mean(axis=(0, 1)), 0.45 ** 2 * m / 2, rtol=0.15)
    # without unwrapping the file path equals the in-memory fast path
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = Onsager(uf.atoms, temperature=300, verbose=False).run(stop=300)
        b = Onsager(um.atoms, temperature=300, verbose=False).run(stop=300)
    assert np.allclose(a.results.msd_self, b.results.msd_self, rtol=1e-9,
                       atol=1e-9 * np.abs(b.results.msd_self).max())


@pytest.mark.parametrize("mode", ["groups", "groups_wrap", "atoms", "atoms_wrap_unwrap"])
def test_onsager_center_from_file_on_device(tmp_path, mode):
    """Onsager(center=True) on a trajectory file: the per-frame system centre of mass is formed on
    the device and subtracted there (reference transport.py:993-1014); equals the per-frame host
    analysis of the same frames."""
    import warnings
    from mdhelper_amd.analysis import Onsager
    rng = np.random.default_rng(71)
    T, N, L = 240, 60, np.array([9.0, 10.5, 8.25])
    drift = np.cumsum(rng.normal(0.05, 0.02, (T, 1, 3)), axis=0)       # a moving centre of mass
    walk = rng.uniform(0, L, (1, N, 3)) + np.cumsum(rng.normal(0, 0.3, (T, N, 3)), axis=0) + drift
    unwrap = mode == "atoms_wrap_unwrap"
    stored = (np.mod(walk, L) if unwrap else walk).astype(np.float32)
    path = tmp_path / "c.nc"
    write_amber_netcdf(path, stored, L, times=np.arange(T) * 0.5)
    dims = np.array([*L, 90, 90, 90], dtype=np.float32)
    masses = rng.uniform(1.0, 30.0, N)
    charges = np.where(np.arange(N) % 2 == 0, 1.0, -1.0)
    uf = mdhelper_amd.FileUniverse(path, dt=0.5, charges=charges, masses=masses)
    um = mdhelper_amd.ArrayUniverse(stored, dims, dt=0.5, charges=charges, masses=masses)
    cat, an = np.arange(0, N - 10, 2), np.arange(1, N - 10, 2)          # 10 atoms in no group
    kw = dict(temperature=300, n_blocks=2, center=True, verbose=False, unwrap=unwrap,
              center_atom=mode.startswith("atoms"), center_wrap="wrap" in mode)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = Onsager([uf.select(cat), uf.select(an)], **kw).run()
        b = per_frame(Onsager([um.select(cat), um.select(an)], **kw)).run()
        c = Onsager([um.select(cat), um.select(an)], **kw).run()
        plain = per_frame(Onsager([um.select(cat), um.select(an)], **{**kw, "center": False})).run()
    assert a._from_file and c._from_file and not b._from_file
    assert np.allclose(c.results.msd_cross, b.results.msd_cross, rtol=1e-8,
                       atol=1e-8 * np.abs(b.results.msd_cross).max())
    for name in ("msd_self", "msd_cross"):
        x, y = getattr(a.results, name), getattr(b.results, name)
        assert np.allclose(x, y, rtol=1e-8, atol=1e-8 * np.abs(y).max()), name
    # the subtraction matters for this drifting system
    assert not np.allclose(plain.results.msd_cross, b.results.msd_cross, rtol=1e-3)


@pytest.mark.parametrize("mode", ["residues", "mixed, unwrap", "residues, unwrap, center", "segments, center_atom",
                                  "residues, unwrap, center_wrap", "mixed, center_wrap",
                                  "segments, unwrap, center_wrap, float64"])
def test_onsager_molecule_groupings_from_file_on_device(tmp_path, mode):
    """Onsager(groupings="residues"/"segments") on a trajectory file: the float64 centres of mass of
    the unwrapped particles are formed on the device (mdx_msd_set_grouping); equals the per-frame
    host analysis (center_of_mass with image flags, transport.py:983-1014).  center_wrap without
    center_atom: the system centre of mass of the WRAPPED molecule centres (:1004-1014), per group
    on the device; float64: in-memory float64 frames through mdx_msd_push_f64."""
    import warnings
    from mdhelper_amd.analysis import Onsager
    rng = np.random.default_rng(91)
    T, n_mol, per = 200, 40, 3
    N = n_mol * per + 9                                              # 9 loose atoms at the end
    L = np.array([10.0, 11.5, 9.25])
    centres = rng.uniform(0, L, (1, n_mol, 1, 3)) + np.cumsum(rng.normal(0.02, 0.3, (T, n_mol, 1, 3)), axis=0)
    shape = rng.normal(0, 0.4, (1, n_mol, per, 3)) + 0.02 * np.cumsum(rng.normal(size=(T, n_mol, per, 3)), axis=0)
    loose = rng.uniform(0, L, (1, 9, 3)) + np.cumsum(rng.normal(0, 0.3, (T, 9, 3)), axis=0)
    walk = np.concatenate([(centres + shape).reshape(T, n_mol * per, 3), loose], axis=1)
    unwrap = "unwrap" in mode
    stored = (np.mod(walk, L) if unwrap else walk).astype(np.float32)
    path = tmp_path / "mol.nc"
    write_amber_netcdf(path, stored, L, times=np.arange(T) * 0.5)
    dims = np.array([*L, 90, 90, 90], dtype=np.float32)
    masses = rng.uniform(1.0, 30.0, N)
    resids = np.r_[np.repeat(np.arange(n_mol), per), n_mol + np.arange(9)]
    segids = np.r_[np.repeat(np.arange(n_mol // 4), 4 * per), n_mol // 4 + np.arange(9) // 3]
    kw_u = dict(dt=0.5, masses=masses, resids=resids, segids=segids,
                charges=np.where(np.arange(N) % 2 == 0, 1.0, -1.0))
    uf = mdhelper_amd.FileUniverse(path, **kw_u)
    um = mdhelper_amd.ArrayUniverse(stored, dims, **kw_u)
    half = (n_mol // 2) * per

    def groups(u):
        if mode.startswith("mixed"):
            return [u.atoms[:half], u.atoms[n_mol * per:]], ["residues", "atoms"]
        if mode.startswith("segments"):
            return [u.atoms[:half], u.atoms[half:n_mol * per]], "segments"
        return [u.atoms[:half], u.atoms[half:n_mol * per]], "residues"

    kw = dict(temperature=300, n_blocks=2, verbose=False, unwrap=unwrap, center="center" in mode,
              center_atom="center_atom" in mode, center_wrap="center_wrap" in mode)
    if "float64" in mode:      # an in-memory float64 trajectory takes the device stages too
        um = mdhelper_amd.ArrayUniverse(stored.astype(np.float64), dims, **kw_u)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ga, gra = groups(uf)
        a = Onsager(ga, gra, **kw).run()
        gb, grb = groups(um)
        b = per_frame(Onsager(gb, grb, **kw)).run()
        c = Onsager(gb, grb, **kw).run()
    assert a._from_file and c._from_file and not b._from_file
    for name in ("msd_self", "msd_cross"):
        x, y = getattr(c.results, name), getattr(b.results, name)
        assert np.allclose(x, y, rtol=1e-8, atol=1e-8 * np.abs(y).max()), name
    assert a.results.msd_self.shape == b.results.msd_self.shape
    for name in ("msd_self", "msd_cross"):
        x, y = getattr(a.results, name), getattr(b.results, name)
        assert np.allclose(x, y, rtol=1e-8, atol=1e-8 * np.abs(y).max()), name
    assert np.abs(b.results.msd_self).max() > 1.0


def test_one_trajectory_outlives_the_engines_that_stage_from_it(tmp_path):
    """
    Round-2 fault (commit 5575fea), as a sequence: engine A stages frames out of a TrajectoryFile
    and is destroyed — with it its streams — while the trajectory stays open; engine B then stages
    out of the same trajectory.  The pinned ring's hand-over events used to be recorded on the
    engine's stream; HIP keeps a pointer to that stream inside the event and reads its capture state
    on the next hipEventSynchronize, i.e. freed memory once the stream is gone ("operation not
    permitted on an event last recorded in a capturing stream", now and then).  The ring, its
    stream and its events now belong to the trajectory (csrc/mdx_common.hpp: HostStager), copies
    stay in flight across calls, and this sequence — several engine generations, of three kinds,
    on different streams — must give the counts of the in-memory frames every time.
    """
    from oracle import fourier as of
    F, N, L = 96, 9000, 44.0                      # 10 MB: the ring (3 x 16 MiB) is never drained by size
    pos = _frames(F, N, L, 21)
    path = tmp_path / "shared.nc"
    write_amber_netcdf(path, pos, (L, L, L))
    t = TrajectoryFile(path)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    edges = np.linspace(0.0, 10.0, 51)
    ref = _core.RdfEngine(edges, (1, 1))
    ref.accumulate(pos, None, dims)
    want = ref.counts()
    ref.close()
    frames = np.arange(F)
    boxes = np.tile(dims, (F, 1))
    q = of.grid_wavevectors([L, L, L], 3)
    for generation in range(4):
        a = _core.RdfEngine(edges, (1, 1))
        a.accumulate_traj(t, frames, boxes)
        if generation % 2:
            a.close()                              # closed with its kernels possibly still queued
            a = None
        s = _core.SqEngine(q, [N], of.ssf_pairs(1, None))      # another engine kind, another stream
        s.accumulate_traj(t, frames[:8])
        s.close()
        b = _core.RdfEngine(edges, (1, 1))
        b.accumulate_traj(t, frames[::-1], boxes)
        assert np.array_equal(b.counts(), want), generation
        b.close()
        if a is not None:
            assert np.array_equal(a.counts(), want), generation
            a.close()
    t.close()


def test_failed_read_leaves_the_trajectory_usable(tmp_path):
    """A read that fails in the middle of a staged batch (the file loses its tail after it was
    opened) returns the I/O error and leaves ring, events and stream consistent: the next engine
    stages the frames that are still there and bins the right counts."""
    import os
    F, N, L = 64, 6000, 38.0
    pos = _frames(F, N, L, 22)
    path = tmp_path / "cut.nc"
    write_amber_netcdf(path, pos, (L, L, L))
    t = TrajectoryFile(path)
    assert t.n_frames == F
    size = os.path.getsize(path)
    os.truncate(path, size - 20 * 12 * N)          # the last ~20 frames are gone
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    edges = np.linspace(0.0, 9.0, 41)
    boxes = np.tile(dims, (F, 1))
    a = _core.RdfEngine(edges, (1, 1))
    with pytest.raises(OSError):
        a.accumulate_traj(t, np.arange(F), boxes)
    a.close()
    keep = np.arange(32)
    b = _core.RdfEngine(edges, (1, 1))
    b.accumulate_traj(t, keep, boxes[:32])
    got = b.counts()
    b.close()
    ref = _core.RdfEngine(edges, (1, 1))
    ref.accumulate(pos[:32], None, dims)
    assert np.array_equal(got, ref.counts())
    ref.close()
    t.close()


def test_onsager_unwrap_makes_split_molecules_whole_first(tmp_path):
    """
    Reference transport.py:936-941: with ``unwrap=True`` every fragment of the first analysed frame is
    made whole before the starting positions are stored.  Dimers, several of them split across the
    cell boundary in frame 0, ``groupings="residues"``, ``center=True, center_wrap=True`` (the one
    combination that sees it: the wrapped CENTRE of a split molecule depends on which images its atoms
    start in).  Three routes must agree: the device stages fed from memory, the device stages fed from
    a file, the per-frame host protocol; and all of them must equal the analysis of the TRUE
    (never wrapped, whole) coordinates without unwrapping — those differ from the made-whole walk by
    one lattice vector per molecule, which the wrap of the centres removes.  A universe without bonds
    (nothing to make whole) must NOT agree: that is what the step is for.
    """
    import warnings
    from mdhelper_amd.analysis import Onsager
    rng = np.random.default_rng(44)
    T, M, L = 48, 40, np.array([12.0, 13.0, 11.5])
    masses = np.tile([12.0, 1.5], M)
    centre0 = rng.uniform(0, L, (M, 3))
    centre0[:6, 0] = L[0] - 0.05 * np.arange(1, 7)         # six dimers straddle the x boundary in frame 0
    centre0[6:9, 2] = 0.02                                 # three the z boundary
    axis = rng.normal(size=(M, 3))
    axis[:6] = [1.0, 0.1, 0.0]
    axis[6:9] = [0.0, 0.1, 1.0]
    axis /= np.linalg.norm(axis, axis=1, keepdims=True)
    walk = centre0 + np.cumsum(rng.normal(0, 0.25, (T, M, 3)), axis=0)
    true = np.empty((T, 2 * M, 3))
    true[:, 0::2] = walk - 0.5 * axis + rng.normal(0, 0.01, (T, M, 3))
    true[:, 1::2] = walk + 0.5 * axis + rng.normal(0, 0.01, (T, M, 3))
    wrapped = np.mod(true, L).astype(np.float32)
    split0 = np.any(np.abs(wrapped[0, 0::2] - wrapped[0, 1::2]) > 0.5 * L, axis=1)
    assert split0.sum() >= 8
    dims = np.array([*L, 90, 90, 90], dtype=np.float32)
    resids = np.repeat(np.arange(M), 2)
    bonds = np.stack((np.arange(0, 2 * M, 2), np.arange(1, 2 * M, 2)), axis=1)
    half = M                                              # atoms of the first 20 dimers / the rest
    kw = dict(temperature=300, groupings="residues", center=True, center_wrap=True, verbose=False)

    def run(u, unwrap, route=None):
        groups = [u.atoms[:half], u.atoms[half:]]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = Onsager(groups, unwrap=unwrap, **kw)
            if route == "per_frame":
                o = per_frame(o)
            return o.run()

    um = mdhelper_amd.ArrayUniverse(wrapped, dims, dt=1.0, masses=masses, resids=resids, bonds=bonds)
    dev = run(um, True)
    host = run(um, True, "per_frame")
    assert dev._from_file and not host._from_file
    path = tmp_path / "split.nc"
    write_amber_netcdf(path, wrapped, L)
    uf = mdhelper_amd.FileUniverse(path, dt=1.0, masses=masses, resids=resids, bonds=bonds)
    fil = run(uf, True)
    ut = mdhelper_amd.ArrayUniverse(true.astype(np.float32), dims, dt=1.0, masses=masses, resids=resids)
    ref = run(ut, False, "per_frame")
    nobond = run(mdhelper_amd.ArrayUniverse(wrapped, dims, dt=1.0, masses=masses, resids=resids), True)
    for name in ("msd_self", "msd_cross"):
        want = host.results[name]
        scale = np.abs(want).max()
        assert np.allclose(dev.results[name], want, rtol=1e-8, atol=1e-9 * scale), name
        assert np.allclose(fil.results[name], want, rtol=1e-8, atol=1e-9 * scale), name
        # float32 storage of wrapped vs never-wrapped coordinates: ~1e-6 A per coordinate
        assert np.allclose(want, ref.results[name], rtol=2e-4, atol=2e-5 * scale), name
    far = np.abs(nobond.results.msd_self - host.results.msd_self).max()
    assert far > 1e-3 * np.abs(host.results.msd_self).max(), far


@pytest.mark.parametrize("mode", ["atoms", "atoms, unwrap, center", "residues, unwrap, center_wrap",
                                  "mixed, center_atom, zero"])
def test_onsager_resident_streamed_and_hbm_universe_routes_agree(tmp_path, mode):
    """The analysed frames are brought into HBM once and every group is prepared from them on the device
    (mdx_msd_push_frames_device); frames too large for that stream group by group (mdx_msd_push_traj /
    mdx_msd_push_f32: forced here through Onsager._hbm_share = 0); a universe over frames already in HBM
    (ArrayUniverse.from_device, float32 and float64) stages nothing.  All of them must give the per-frame
    host protocol's result, for every kind of frame preparation."""
    import warnings
    from mdhelper_amd.analysis import Onsager
    rng = np.random.default_rng(123)
    T, n_mol, per = 300, 36, 3            # 300 frames: three segments of the segment-parallel unwrap
    N = n_mol * per + 7
    L = np.array([9.0, 10.5, 8.25])
    centres = rng.uniform(0, L, (1, n_mol, 1, 3)) + np.cumsum(rng.normal(0.02, 0.35, (T, n_mol, 1, 3)), axis=0)
    shape = rng.normal(0, 0.4, (1, n_mol, per, 3)) + 0.02 * np.cumsum(rng.normal(size=(T, n_mol, per, 3)), axis=0)
    loose = rng.uniform(0, L, (1, 7, 3)) + np.cumsum(rng.normal(0, 0.3, (T, 7, 3)), axis=0)
    walk = np.concatenate([(centres + shape).reshape(T, n_mol * per, 3), loose], axis=1)
    unwrap = "unwrap" in mode
    stored = (np.mod(walk, L) if unwrap else walk).astype(np.float32)
    path = tmp_path / "routes.nc"
    write_amber_netcdf(path, stored, L, times=np.arange(T) * 0.5)
    dims = np.array([*L, 90, 90, 90], dtype=np.float32)
    kw_u = dict(dt=0.5, masses=rng.uniform(1.0, 30.0, N),
                resids=np.r_[np.repeat(np.arange(n_mol), per), n_mol + np.arange(7)],
                charges=np.where(np.arange(N) % 2 == 0, 1.0, -1.0))
    half = (n_mol // 2) * per

    def groups(u):
        if mode.startswith("mixed"):
            return [u.atoms[:half], u.atoms[n_mol * per:]], ["residues", "atoms"]
        if mode.startswith("residues"):
            return [u.atoms[:half], u.atoms[half:n_mol * per]], "residues"
        # atoms: one contiguous range, one scattered selection
        return [u.atoms[:half], u.select(np.arange(half + 1, N, 2))], "atoms"

    kw = dict(temperature=300, n_blocks=2, verbose=False, unwrap=unwrap, center="center" in mode,
              center_atom="center_atom" in mode, center_wrap="center_wrap" in mode,
              dimensions=[L[0], L[1], 0.0] if "zero" in mode else None)

    def run(u, share=None, route=None):
        g, gr = groups(u)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = Onsager(g, gr, **kw)
            if share is not None:
                o._hbm_share = share
            return (per_frame(o) if route == "per_frame" else o).run()

    um = mdhelper_amd.ArrayUniverse(stored, dims, **kw_u)
    uf = mdhelper_amd.FileUniverse(path, **kw_u)
    d32 = _core.DeviceArray.from_host(stored)
    d64 = _core.DeviceArray.from_host(stored.astype(np.float64))
    ud32 = mdhelper_amd.ArrayUniverse.from_device(d32, dims, **kw_u)
    ud64 = mdhelper_amd.ArrayUniverse.from_device(d64, dims, **kw_u)
    want = run(um, route="per_frame")
    assert not want._from_file
    routes = {"memory, resident": run(um), "memory, streamed": run(um, 0.0), "file, resident": run(uf),
              "file, streamed": run(uf, 0.0), "hbm float32": run(ud32), "hbm float64": run(ud64),
              "hbm float64, a frame range": None}
    # a sub-range of the HBM frames is a window, irregular frames go through the host
    g, gr = groups(ud64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        win = Onsager(g, gr, **kw).run(start=20, stop=260)
        irr = Onsager(g, gr, **kw).run(frames=np.arange(20, 260, 1)[::2])
        g, gr = groups(um)
        win_ref = per_frame(Onsager(g, gr, **kw)).run(start=20, stop=260)
        irr_ref = per_frame(Onsager(g, gr, **kw)).run(frames=np.arange(20, 260, 1)[::2])
    del routes["hbm float64, a frame range"]
    for name in ("msd_self", "msd_cross"):
        y = want.results[name]
        for route, got in routes.items():
            assert got._from_file, route
            assert np.allclose(got.results[name], y, rtol=1e-8, atol=1e-8 * np.abs(y).max()), (route, name)
        for got, ref in ((win, win_ref), (irr, irr_ref)):
            y = ref.results[name]
            assert np.allclose(got.results[name], y, rtol=1e-8, atol=1e-8 * np.abs(y).max()), name
    assert np.abs(want.results.msd_self).max() > 1.0
    d32.free()
    d64.free()


@pytest.mark.parametrize("mode", ["plain", "unwrap", "unwrap, bonds", "zero, blocks"])
def test_onsager_host_groups_streamed_in_column_chunks(mode):
    """Plain atom groups over consecutive particles of an in-memory trajectory travel in column chunks
    (mdx_upload_rows into two alternating device blocks, the chunk before transformed meanwhile:
    Onsager._stream_host_groups); the result must be that of whole frames kept in HBM and of the per-frame
    host protocol — ragged last chunks, unwrapping with its per-chunk state, molecules made whole in the first
    frame, a zeroed dimension, several blocks."""
    import warnings
    from mdhelper_amd.analysis import Onsager
    rng = np.random.default_rng(77)
    T, sizes, L = 260, (53, 38), np.array([9.0, 10.5, 8.25])
    N = sum(sizes) + 5
    walk = rng.uniform(0, L, (1, N, 3)) + np.cumsum(rng.normal(0.01, 0.3, (T, N, 3)), axis=0)
    unwrap = "unwrap" in mode
    stored = (np.mod(walk, L) if unwrap else walk).astype(np.float32)
    dims = np.array([*L, 90, 90, 90], dtype=np.float32)
    bonds = np.stack((np.arange(0, N - 1, 2), np.arange(1, N, 2)), axis=1) if "bonds" in mode else None
    u = mdhelper_amd.ArrayUniverse(stored, dims, dt=0.5, bonds=bonds)
    kw = dict(temperature=300, n_blocks=4 if "blocks" in mode else 1, unwrap=unwrap, verbose=False,
              dimensions=[L[0], 0.0, L[2]] if "zero" in mode else None)

    def run(route):
        groups = [u.atoms[2:2 + sizes[0]], u.atoms[2 + sizes[0]:2 + sum(sizes)]]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            o = Onsager(groups, **kw)
            o._stream_columns = route == "columns"
            return (per_frame(o) if route == "per_frame" else o).run()

    cols, whole, host = run("columns"), run("whole"), run("per_frame")
    assert cols._from_file and whole._from_file and not host._from_file
    for name in ("msd_self", "msd_cross"):
        y = host.results[name]
        assert np.allclose(cols.results[name], y, rtol=1e-8, atol=1e-9 * np.abs(y).max()), name
        assert np.allclose(whole.results[name], y, rtol=1e-8, atol=1e-9 * np.abs(y).max()), name
    assert np.abs(host.results.msd_self).max() > 0.5


def test_onsager_unwrap_of_a_chain_longer_than_two_cells_starts_like_the_reference():
    """Reference topology.py:366-376: the first ``unwrap`` call after ``make_whole`` (transport.py:936-941)
    moves an image flag by sign(x - x_whole) — by one — however many cells a made-whole atom lies from
    its stored image.  A 7-bead chain with bonds of 0.4 L spans more than two cells: the device routes
    must start from the clipped flags like the per-frame host protocol does (center=True,
    center_wrap=True with molecule groupings is where it shows)."""
    import warnings
    from mdhelper_amd.algorithm.topology import make_whole_images
    from mdhelper_amd.analysis import Onsager
    rng = np.random.default_rng(5)
    T, L, n_chain, beads = 60, np.array([10.0, 10.0, 10.0]), 6, 7
    start = rng.uniform(0, L, (n_chain, 1, 3))
    chain = start + np.arange(beads)[None, :, None] * np.array([0.4 * L[0], 0.05, -0.03])
    true = chain.reshape(1, n_chain * beads, 3) + np.cumsum(rng.normal(0, 0.15, (T, n_chain * beads, 3)), axis=0)
    wrapped = np.mod(true, L).astype(np.float32)
    N = n_chain * beads
    bonds = np.array([(c * beads + b, c * beads + b + 1) for c in range(n_chain) for b in range(beads - 1)])
    dims = np.array([*L, 90, 90, 90], dtype=np.float32)
    u = mdhelper_amd.ArrayUniverse(wrapped, dims, dt=1.0, masses=rng.uniform(1, 20, N),
                                   resids=np.repeat(np.arange(n_chain), beads), bonds=bonds)
    u.trajectory[0]
    assert np.abs(make_whole_images(u, L.astype(float))).max() >= 2      # the case the clipping is for
    kw = dict(temperature=300, groupings="residues", center=True, center_wrap=True, unwrap=True, verbose=False)
    groups = [u.atoms[:3 * beads], u.atoms[3 * beads:]]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        dev = Onsager(groups, **kw).run()
        streamed = Onsager(groups, **kw)
        streamed._hbm_share = 0.0
        streamed.run()
        host = per_frame(Onsager(groups, **kw)).run()
    assert dev._from_file and streamed._from_file and not host._from_file
    for name in ("msd_self", "msd_cross"):
        y = host.results[name]
        assert np.allclose(dev.results[name], y, rtol=1e-8, atol=1e-9 * np.abs(y).max()), name
        assert np.allclose(streamed.results[name], y, rtol=1e-8, atol=1e-9 * np.abs(y).max()), name


def test_page_locked_caller_memory_is_read_in_place(tmp_path):
    """`mdx_host_register`: a caller buffer page-locked once is handed to the DMA engine where it lies
    (no staging copy through the pinned ring); pageable, registered and again pageable (after
    `mdx_host_unregister`) give the same counts, through engines that come and go (cached device blocks
    and streams are reused between them)."""
    from mdhelper_amd import _lib
    rng = np.random.default_rng(9)
    F, N, L = 300, 5000, np.float32(40.0)                 # 18 MB: more than one 16 MiB ring chunk
    pos = (rng.random((F, N, 3)) * L).astype(np.float32)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    edges = np.linspace(0.0, 8.0, 33)

    def counts():
        eng = _core.RdfEngine(edges, (1, 1))
        eng.accumulate(pos, None, dims)
        out = eng.counts()
        eng.close()
        return out

    want = counts()
    assert want.sum() > 0
    _lib.check(_lib.lib().mdx_host_register(0, pos.ctypes.data, pos.nbytes))
    try:
        for _ in range(3):
            assert np.array_equal(counts(), want)
    finally:
        _lib.check(_lib.lib().mdx_host_unregister(0, pos.ctypes.data))
    assert np.array_equal(counts(), want)
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().mdx_host_register(0, None, 0))
