"""
GPU parity tests at the C-ABI level (through the ctypes wrappers of
mdhelper_amd._core): every engine of libmdx.so against the CPU oracle on the
same seeded inputs.  Integer bin counts must be bit-exact; floats within 1e-6
relative (absolute at MSD lag 0, which is pure round-off).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from mdhelper_amd import _core  # noqa: E402
from oracle import correlation as oc  # noqa: E402
from oracle import fourier as of  # noqa: E402
from oracle import rdf as orf  # noqa: E402
from oracle.cbind import c_radial_histogram  # noqa: E402

ALGOS = ["exact", "filter", "cell"]


def _edges(n_bins, rng):
    return np.linspace(rng[0], rng[1], n_bins + 1)


def _gpu_hist(p1, p2, n_bins, rng, dims, exclusion=None, algo="auto"):
    eng = _core.RdfEngine(_edges(n_bins, rng), exclusion, algo=algo)
    eng.accumulate(p1, p2, dims)
    out = eng.counts()
    eng.close()
    return out


@pytest.mark.parametrize("algo", ALGOS)
@pytest.mark.parametrize("rng_range", [(0.0, 15.0), (2.0, 9.5), (0.0, 34.47), (1.5, 34.47)])
@pytest.mark.parametrize("exclusion", [None, (1, 1), (3, 3)])
def test_rdf_self_cubic(algo, rng_range, exclusion):
    rng = np.random.default_rng(5)
    L = np.float32(68.94)
    pos = (rng.random((1500, 3)) * L).astype(np.float32)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    want = c_radial_histogram(pos, pos, 201, rng_range, dims, exclusion=exclusion)
    got = _gpu_hist(pos, None, 201, rng_range, dims, exclusion, algo)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("algo", ALGOS)
def test_rdf_two_groups_non_cubic_outside_box(algo):
    rng = np.random.default_rng(6)
    dims = np.array([30.0, 41.5, 27.25, 90, 90, 90], dtype=np.float32)
    p1 = (rng.random((700, 3)) * dims[:3]).astype(np.float32)
    p2 = (rng.random((1100, 3)) * dims[:3] * 5 - 2 * dims[:3]).astype(np.float32)
    for excl in (None, (2, 5)):
        want = c_radial_histogram(p1, p2, 64, (0.5, 12.0), dims, exclusion=excl)
        got = _gpu_hist(p1, p2, 64, (0.5, 12.0), dims, excl, algo)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("algo", ALGOS)
def test_rdf_no_box_and_large_tiles(algo):
    rng = np.random.default_rng(8)
    pos = (rng.normal(size=(2500, 3)) * 12).astype(np.float32)
    want = c_radial_histogram(pos, pos, 100, (0.0, 20.0), None, exclusion=(1, 1))
    got = _gpu_hist(pos, None, 100, (0.0, 20.0), None, (1, 1), algo)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("algo", ALGOS)
def test_rdf_frames_with_changing_box(algo):
    rng = np.random.default_rng(9)
    F, N = 5, 1200
    Ls = (40 + 3 * rng.random(F)).astype(np.float32)
    frames = (rng.random((F, N, 3)) * Ls[:, None, None]).astype(np.float32)
    boxes = np.stack([np.array([L, L, L, 90, 90, 90], dtype=np.float32) for L in Ls])
    want = np.zeros(150, dtype=np.int64)
    for f in range(F):
        want += c_radial_histogram(frames[f], frames[f], 150, (0.0, 14.0), boxes[f], exclusion=(1, 1))
    eng = _core.RdfEngine(_edges(150, (0.0, 14.0)), (1, 1), algo=algo, timing=True)
    eng.accumulate(frames, None, boxes)
    got = eng.counts()
    st = eng.stats()
    assert np.array_equal(got, want)
    assert st["pairs_evaluated"] == F * N * N and st["kernel_ms"] > 0
    # accumulate again: counts double
    eng.accumulate(frames, None, boxes)
    assert np.array_equal(eng.counts(), 2 * want)
    eng.reset()
    assert eng.counts().sum() == 0
    eng.close()


@pytest.mark.parametrize("algo", ALGOS)
def test_rdf_adversarial_lattice_on_bin_edges(algo):
    """Simple-cubic lattice whose spacing equals the bin width: distances sit on edges."""
    n, a = 12, np.float32(0.75)
    idx = np.arange(n)
    pos = (a * np.stack(np.meshgrid(idx, idx, idx, indexing="ij"), -1).reshape(-1, 3)).astype(np.float32)
    L = np.float32(n * a)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    for rng_range, nb in [((0.0, 4.5), 6), ((0.0, 4.5), 60), ((0.75, 3.75), 4)]:
        want = c_radial_histogram(pos, pos, nb, rng_range, dims, exclusion=(1, 1))
        got = _gpu_hist(pos, None, nb, rng_range, dims, (1, 1), algo)
        assert np.array_equal(got, want), (rng_range, nb)


@pytest.mark.parametrize("algo", ALGOS)
def test_rdf_half_box_separations(algo):
    """Pairs exactly half a box apart (image choice ambiguous) with range up to L/2."""
    rng = np.random.default_rng(10)
    L = np.float32(20.0)
    base = (rng.random((600, 3)) * L).astype(np.float32)
    shifted = base.copy()
    shifted[:, 0] = np.mod(shifted[:, 0] + L / 2, L)
    pos = np.vstack((base, shifted)).astype(np.float32)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    want = c_radial_histogram(pos, pos, 50, (0.0, 10.0), dims)
    got = _gpu_hist(pos, None, 50, (0.0, 10.0), dims, None, algo)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("exclusion", [None, (2, 2)])
@pytest.mark.parametrize("dims", [(40.0, 40.0, 40.0), (30.0, 75.0, 120.0), (110.0, 31.0, 34.0), (64.0, 100.0, 33.0)])
def test_rdf_cell_tile_pairs_straddling_half_a_box(dims, exclusion):
    """Ranges up to half the shortest edge: tile pairs whose separation reaches L/2 in some components
    take the per-pair image fold in exactly those components (mdx_rdf_cell.hpp, MDX_CELL_GENERAL); the
    elongated cells make every subset of components occur (long edges never straddle)."""
    rng = np.random.default_rng(31)
    box = np.array([*dims, 90, 90, 90], dtype=np.float32)
    pos = (rng.random((6000, 3)) * box[:3]).astype(np.float32)
    pos[:300] -= box[:3] * np.array([1, 2, -1], dtype=np.float32)      # unwrapped images
    other = (rng.random((1500, 3)) * box[:3]).astype(np.float32)
    half = 0.5 * float(min(dims))
    for rng_range, nb in [((0.0, half), 201), ((0.0, 0.97 * half), 64), ((0.3 * half, half), 50)]:
        want = c_radial_histogram(pos, pos, nb, rng_range, box, exclusion=exclusion)
        eng = _core.RdfEngine(_edges(nb, rng_range), exclusion, algo="cell", timing=True)
        eng.accumulate(pos, None, box)
        got = eng.counts()
        st = eng.stats()
        eng.close()
        assert np.array_equal(got, want), (rng_range, nb)
        assert st["cell_units_general"] > 0, "no tile pair took the per-pair image path"
    want = c_radial_histogram(pos, other, 120, (0.0, half), box, exclusion=exclusion)
    got = _gpu_hist(pos, other, 120, (0.0, half), box, exclusion, "cell")
    assert np.array_equal(got, want)


def test_rdf_cell_many_slabs(monkeypatch):
    """Frames beyond one slab of sorted copies (mdx_rdf.hip::accumulate_cell: sort and pair kernel of slab after slab
    on the handle's stream).  MDX_RDF_SLAB_BYTES shrinks the slab so that 23 frames take eight slabs; two groups,
    changing boxes, two accumulate calls on one engine."""
    monkeypatch.setenv("MDX_RDF_SLAB_BYTES", str(3 * 33 * (3072 + 1024)))     # three frames per slab
    rng = np.random.default_rng(41)
    F, n1, n2 = 23, 3000, 1000
    Ls = (36 + 4 * rng.random((F, 3))).astype(np.float32)
    boxes = np.concatenate([Ls, np.full((F, 3), 90, np.float32)], axis=1)
    a = (rng.random((F, n1, 3)) * Ls[:, None, :]).astype(np.float32)
    b = (rng.random((F, n2, 3)) * Ls[:, None, :]).astype(np.float32)
    want = np.zeros(90, dtype=np.int64)
    for f in range(F):
        want += c_radial_histogram(a[f], b[f], 90, (0.5, 11.0), boxes[f])
    eng = _core.RdfEngine(_edges(90, (0.5, 11.0)), None, algo="cell")
    eng.accumulate(a[:14], b[:14], boxes[:14])
    eng.accumulate(a[14:], b[14:], boxes[14:])
    got = eng.counts()
    eng.close()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("exclusion", [None, (1, 1), (4, 4)])
def test_rdf_cell_culling_regime(exclusion):
    """Enough particles and a short range: tiles are culled and take the shifted fast path."""
    rng = np.random.default_rng(18)
    dims = np.array([52.0, 47.5, 61.25, 90, 90, 90], dtype=np.float32)
    pos = (rng.random((7000, 3)) * dims[:3]).astype(np.float32)
    pos[:500] += dims[:3] * np.array([3, -2, 1], dtype=np.float32)     # unwrapped images
    for rng_range, nb in [((0.0, 7.5), 201), ((1.0, 9.0), 64)]:
        want = c_radial_histogram(pos, pos, nb, rng_range, dims, exclusion=exclusion)
        eng = _core.RdfEngine(_edges(nb, rng_range), exclusion, algo="cell", timing=True)
        eng.accumulate(pos, None, dims)
        got = eng.counts()
        st = eng.stats()
        eng.close()
        assert np.array_equal(got, want), (rng_range, nb)
        # culling actually happened: far fewer distance evaluations than the pair space
        assert 0 < st["pairs_computed"] < 0.5 * st["pairs_evaluated"]
    # two different groups through the cell path
    p2 = (rng.random((3000, 3)) * dims[:3]).astype(np.float32)
    want = c_radial_histogram(pos, p2, 100, (0.0, 8.0), dims, exclusion=exclusion)
    assert np.array_equal(_gpu_hist(pos, p2, 100, (0.0, 8.0), dims, exclusion, "cell"), want)


def test_rdf_cell_random_walk_frames_match_filter():
    """Synthetic bench-like frames: cell == filter == oracle on a few frames."""
    L = 40.0
    d = _core.synth_random_walk(4, 6400, [L, L, L], 0.3, seed=5)
    frames = d.to_host()
    d.free()
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    want = np.zeros(201, dtype=np.int64)
    for f in range(4):
        want += c_radial_histogram(frames[f], frames[f], 201, (0.0, 15.0), dims, exclusion=(1, 1))
    for algo in ("cell", "filter", "auto"):
        eng = _core.RdfEngine(_edges(201, (0.0, 15.0)), (1, 1), algo=algo)
        eng.accumulate(frames, None, dims)
        assert np.array_equal(eng.counts(), want), algo
        eng.close()


@pytest.mark.parametrize("seed", (0, 5, 11, 17, 23, 39))
def test_rdf_reference_test_geometry(seed):
    """The reference's own exact-count test (reference tests/test_analysis_structure.py:21-40), asserted as the
    reference asserts it — ``np.array_equal(np.histogram(norm, ...)[0], radial_histogram(origin, neighbors, ...))`` —
    on the HIP path: through ``mdx_radial_histogram``, through the module function with the reference's argument
    forms (``pos1`` of shape (3,), integer ``dims``), and through the class on a one-frame universe.  Seeds and the
    reason a float32 coordinate cannot move a count there: tests/test_oracle_rdf.py::reference_test_geometry."""
    import mdhelper_amd
    from mdhelper_amd.analysis import RadialDistributionFunction, structure
    from test_oracle_rdf import reference_test_geometry
    origin, neighbors, half_L, dims, counts, margin = reference_test_geometry(seed)
    assert margin > 1e-5 and counts.sum() == 1000
    edges = np.linspace(0, half_L + 1, half_L + 1)
    assert np.array_equal(counts, _core.radial_histogram_device(origin, neighbors, half_L, edges, dims))
    assert np.array_equal(counts, structure.radial_histogram(origin, neighbors, n_bins=half_L,
                                                             range=(0, half_L + 1), dims=dims))
    for algo in ALGOS:
        got = _gpu_hist(origin[None].astype(np.float32), neighbors.astype(np.float32), half_L, (0, half_L + 1),
                        dims.astype(np.float32), None, algo)
        assert np.array_equal(counts, got), algo
    u = mdhelper_amd.ArrayUniverse(np.concatenate([origin[None], neighbors])[None].astype(np.float32),
                                   dims.astype(np.float32))
    rdf = RadialDistributionFunction(u.atoms[:1], u.atoms[1:], n_bins=half_L, range=(0, half_L + 1)).run()
    assert np.array_equal(counts, rdf.results.counts)


def test_rdf_bad_cell_rejected_and_empty():
    pos = np.zeros((4, 3), dtype=np.float32)
    eng = _core.RdfEngine(_edges(4, (0, 1)))
    with pytest.raises(ValueError):      # angles that span no volume
        eng.accumulate(pos, None, np.array([10, 10, 10, 30, 30, 170], dtype=np.float32))
    with pytest.raises(ValueError):
        eng.accumulate(pos, None, np.array([10, -1, 10, 90, 90, 90], dtype=np.float32))
    eng.accumulate(np.zeros((0, 3), dtype=np.float32).reshape(1, 0, 3), None, None)
    assert eng.counts().sum() == 0
    eng.close()


def test_rdf_many_bins_and_tiny_bins():
    rng = np.random.default_rng(12)
    L = np.float32(25.0)
    pos = (rng.random((800, 3)) * L).astype(np.float32)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    for nb in (1, 7, 5000, 20000):
        want = c_radial_histogram(pos, pos, nb, (0.0, 12.0), dims, exclusion=(1, 1))
        for algo in ALGOS:
            got = _gpu_hist(pos, None, nb, (0.0, 12.0), dims, (1, 1), algo)
            assert np.array_equal(got, want), (nb, algo)


def test_fourier_sum_matches_golden(golden_dir):
    g = np.load(golden_dir / "fourier_ref.npz")
    got = _core.fourier_sum_device(g["qs"], g["rs"])
    assert np.allclose(got, g["out_fourier_sum"], rtol=1e-9, atol=1e-9)


def test_accelerated_drop_ins_match_the_reference_vectors(golden_dir):
    """Every function of the reference's ``algorithm/accelerated.py`` (:12-627) by its own name, and the public
    static methods ``StructureFactor.ssf_trigonometric_2d`` / ``psf_trigonometric_2d_2d`` (structure.py:1238-1317),
    computed on the device, against vectors the reference's own loop bodies produced (scripts/make_golden.py)."""
    from mdhelper_amd.algorithm import accelerated as acc
    from mdhelper_amd.analysis import StructureFactor
    g = np.load(golden_dir / "fourier_ref.npz")
    qs, rs, rs2 = g["qs"], g["rs"], g["rs2"]
    tol = dict(rtol=1e-10, atol=1e-10)
    assert np.allclose(acc.delta_fourier_transform_sum_2d_2d(qs, rs), g["out_fourier_sum"], **tol)
    assert np.allclose(acc.delta_fourier_transform_sum_parallel_2d_2d(qs, rs2), g["out_fourier_sum_parallel"], **tol)
    qr = acc.inner_2d_2d(qs, rs)
    qr2 = acc.inner_parallel_2d_2d(qs, rs2)
    assert np.allclose(qr, g["out_inner"], rtol=0, atol=1e-12)
    assert np.allclose(qr2, g["out_inner_parallel"], rtol=0, atol=1e-12)
    assert np.allclose(StructureFactor.ssf_trigonometric_2d(g["out_inner"]), g["out_pythag"], **tol)
    assert np.allclose(StructureFactor.psf_trigonometric_2d_2d(g["out_inner"], g["out_inner_parallel"]),
                       g["out_pythag_cross"], **tol)
    assert np.allclose([acc.pythagorean_trigonometric_identity_1d(row) for row in qr[:5]], g["out_pythag"][:5], **tol)
    assert np.allclose([acc.pythagorean_trigonometric_identity_1d_1d(a, b) for a, b in zip(qr[:5], qr2[:5])],
                       g["out_pythag_cross"][:5], **tol)
    assert np.allclose(acc.cosine_sum_2d(qr), g["out_cosine_sum_2d"], **tol)
    assert np.allclose(acc.sine_sum_2d(qr), g["out_sine_sum_2d"], **tol)
    assert np.allclose(acc.cosine_sum_parallel_2d(qr2), g["out_cosine_sum_parallel_2d"], **tol)
    assert np.allclose(acc.sine_sum_parallel_2d(qr2), g["out_sine_sum_parallel_2d"], **tol)
    assert np.isclose(acc.cosine_sum_1d(qr[3]), g["out_cosine_sum_1d"], **tol)
    assert np.isclose(acc.sine_sum_1d(qr[3]), g["out_sine_sum_1d"], **tol)
    for name in ("cosine_sum_inplace_2d", "cosine_sum_inplace_parallel_2d", "sine_sum_inplace_2d",
                 "sine_sum_inplace_parallel_2d"):
        hold = np.full(len(qs), np.nan)
        getattr(acc, name)(qr2, hold)
        assert np.allclose(hold, g["out_" + name], **tol), name
    assert np.isclose(acc.dot_1d_1d(qs[5], rs[7]), g["out_dot_1d_1d"], rtol=1e-15)
    assert np.isclose(acc.delta_fourier_transform_1d_1d(qs[5], rs[7]), g["out_delta_1d_1d"], rtol=1e-14)
    with pytest.raises(ValueError):
        StructureFactor.psf_trigonometric_2d_2d(qr, qr2[:3])


def test_trig_rowsums_at_size_and_at_large_phases():
    """mdx_trig_rowsums beyond the fixture: ragged columns split across blocks, more rows than one launch's grid
    holds, row slabs, an empty row set of columns, and phases past the fast reduction's range — against numpy."""
    rng = np.random.default_rng(77)
    for shape in ((3, 100_003), (70_000, 7), (257, 4099), (5, 0), (1, 1)):
        x = rng.uniform(-300.0, 300.0, size=shape)
        c, s = _core.trig_rowsums_device(x)
        assert np.allclose(c, np.cos(x).sum(axis=1), rtol=1e-10, atol=1e-9 * max(1, shape[1]) ** 0.5), shape
        assert np.allclose(s, np.sin(x).sum(axis=1), rtol=1e-10, atol=1e-9 * max(1, shape[1]) ** 0.5), shape
    x = rng.uniform(-1e12, 1e12, size=(4, 1000))
    c, s = _core.trig_rowsums_device(x)
    assert np.allclose(c, np.cos(x).sum(axis=1), atol=1e-9) and np.allclose(s, np.sin(x).sum(axis=1), atol=1e-9)
    only_c, none = _core.trig_rowsums_device(x, sin=False)
    assert none is None and np.array_equal(only_c, c)
    q = rng.normal(size=(40, 3))
    r = rng.normal(size=(1234, 3)) * 10
    want = (q[:, None, 0] * r[None, :, 0] + q[:, None, 1] * r[None, :, 1]) + q[:, None, 2] * r[None, :, 2]
    assert np.allclose(_core.inner_device(q, r), want, rtol=0, atol=1e-12)


def test_sq_engine_partial_and_total():
    rng = np.random.default_rng(13)
    L = 31.7
    F, sizes = 3, [700, 500, 300]
    N = sum(sizes)
    frames = (rng.random((F, N, 3)) * L).astype(np.float32)
    q = of.grid_wavevectors([L, L, L], 5)
    for mode in (None, "partial", "pair"):
        pairs = of.ssf_pairs(len(sizes), mode)
        eng = _core.SqEngine(q, sizes, pairs, timing=True)
        eng.accumulate(frames)
        got = eng.result()
        slices, idx = [], 0
        for n in sizes:
            slices.append(slice(idx, idx + n))
            idx += n
        want = sum(of.ssf_frame_ref(q, frames[f].astype(np.float64), slices, pairs, mode) for f in range(F))
        scale = np.abs(want).max()
        assert np.allclose(got, want, rtol=1e-6, atol=1e-9 * scale), mode
        assert eng.stats()["kernel_ms"] > 0
        eng.close()


def test_sq_large_phase_accuracy():
    """Unwrapped coordinates far from the origin: |q.r| ~ 1e4 still within 1e-6."""
    rng = np.random.default_rng(14)
    L = 20.0
    pos = (rng.random((4000, 3)) * L + 40 * L).astype(np.float32)
    q = of.grid_wavevectors([L, L, L], 6)[1:]
    eng = _core.SqEngine(q, [4000], ((None, None),))
    eng.accumulate(pos[None])
    got = eng.result()[0]
    rho = of.fourier_sum_ref(q, pos.astype(np.float64))
    want = (rho * rho.conj()).real
    assert np.allclose(got, want, rtol=1e-6, atol=1e-6 * 4000)
    eng.close()


def test_msd_engine_matches_oracle():
    rng = np.random.default_rng(15)
    B, Tb, sizes = 2, 60, [7, 5]
    N = sum(sizes)
    pos = np.cumsum(rng.normal(size=(B * Tb + 3, N, 3)), axis=0) + 50.0   # 3 trailing frames ignored
    eng = _core.MsdEngine(Tb, B, len(sizes), timing=True)
    first = 0
    for g, n in enumerate(sizes):
        eng.push(g, pos, first, n)
        first += n
    msd, traj = eng.result()
    first = 0
    for g, n in enumerate(sizes):
        p = pos[:B * Tb, first:first + n].reshape(B, Tb, n, 3)
        want = oc.msd_fft_ref(p, axis=1, average=False).sum(axis=-1)
        assert np.allclose(msd[g], want, rtol=1e-6, atol=1e-7), g
        assert np.allclose(traj[g], p.sum(axis=2), rtol=1e-12, atol=1e-9)
        first += n
    assert eng.stats()["bytes_moved"] > 0
    eng.close()


def test_msd_zero_dims_and_split_push():
    rng = np.random.default_rng(16)
    Tb, n = 100, 9
    pos = np.cumsum(rng.normal(size=(Tb, n, 3)), axis=0)
    eng = _core.MsdEngine(Tb, 1, 1)
    eng.push(0, pos, 0, 4, zero_dims=0b100)
    eng.push(0, pos, 4, 5, zero_dims=0b100)
    msd, _ = eng.result()
    p = pos.copy()
    p[..., 2] = 0
    want = oc.msd_fft_ref(p[None], axis=1, average=False).sum(axis=-1)
    assert np.allclose(msd[0], want, rtol=1e-6, atol=1e-7)
    eng.close()


def test_correlate_matches_oracle():
    rng = np.random.default_rng(17)
    a = rng.normal(size=(6, 37))
    b = rng.normal(size=(6, 37))
    pos, neg = _core.correlate_device(a, b, negative=True)
    full = oc.correlation_fft_ref(a, b, axis=1)          # normalised, lags -(T-1)..T-1
    T = 37
    w = np.arange(T, 0, -1)
    assert np.allclose(pos / w, full[:, T - 1:], rtol=1e-9, atol=1e-10)
    assert np.allclose((neg / w)[:, 1:], full[:, T - 2::-1], rtol=1e-9, atol=1e-10)
    acf = _core.correlate_device(a)
    assert np.allclose(acf / w, oc.correlation_fft_ref(a, axis=1), rtol=1e-9, atol=1e-10)


def test_synth_walk_is_deterministic_and_wrapped():
    L = 68.94
    d = _core.synth_random_walk(6, 4096, [L, L, L], 0.3, seed=2)
    a = d.to_host()
    d2 = _core.synth_random_walk(6, 4096, [L, L, L], 0.3, seed=2)
    assert np.array_equal(a, d2.to_host())
    assert a.min() >= 0 and a.max() < L
    step = a[1] - a[0]
    step -= L * np.rint(step / L)
    assert 0.25 < step.std() < 0.35
    d.free()
    d2.free()


def test_rccl_comm_single_rank_paths():
    """RCCL communicator with one rank: the all-reduce entry points run and leave results unchanged."""
    comm = _core.RcclComm(0, 1, _core.RcclComm.unique_id())
    comm.barrier()
    assert np.array_equal(comm.allreduce(np.arange(5, dtype=np.int64)), np.arange(5))
    assert np.allclose(comm.allreduce(np.array([1.5, -2.0]), op="max"), [1.5, -2.0])
    rng = np.random.default_rng(30)
    L = np.float32(20.0)
    pos = (rng.random((2, 1500, 3)) * L).astype(np.float32)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    eng = _core.RdfEngine(_edges(50, (0.0, 8.0)), (1, 1))
    eng.accumulate(pos, None, dims)
    before = eng.counts()
    eng.allreduce(comm)
    assert np.array_equal(eng.counts(), before)
    with pytest.raises(RuntimeError):
        eng.accumulate(pos, None, dims)          # all-reduced handle must be reset first
    eng.reset()
    eng.accumulate(pos, None, dims)
    assert np.array_equal(eng.counts(), before)
    eng.close()
    q = of.grid_wavevectors([20.0, 20.0, 20.0], 3)
    sq = _core.SqEngine(q, [1500], ((None, None),))
    sq.accumulate(pos)
    ref = sq.result()
    sq.allreduce(comm)
    assert np.array_equal(sq.result(), ref)
    sq.close()
    walk = np.cumsum(rng.normal(size=(64, 6, 3)), axis=0)
    m = _core.MsdEngine(64, 1, 1)
    m.push(0, walk, 0, 6)
    a0, t0 = m.result()
    m.reset()
    m.push(0, walk, 0, 6)
    m.allreduce(comm)
    a1, t1 = m.result()
    assert np.allclose(a0, a1) and np.allclose(t0, t1)
    m.close()
    comm.close()


def test_sq_lattice_and_general_paths_agree(monkeypatch):
    """Wavevectors on a (non-cubic, signed) reciprocal lattice take the separable-table kernel;
    arbitrary wavevectors take the sincos kernel; both within 1e-6 of the oracle."""
    rng = np.random.default_rng(32)
    dims = np.array([21.0, 17.5, 26.0])
    pos = (rng.random((2, 3000, 3)) * dims * 3 - dims).astype(np.float32)     # unwrapped too
    m = rng.integers(-9, 10, size=(300, 3))
    m[0] = 0
    q_lat = 2 * np.pi * m / dims
    q_gen = q_lat + rng.normal(scale=1e-3, size=q_lat.shape)                  # off the lattice
    sizes = [1800, 1200]
    slices = [slice(0, 1800), slice(1800, 3000)]
    pairs = of.ssf_pairs(2, "partial")
    for q in (q_lat, q_gen):
        want = sum(of.ssf_frame_ref(q, pos[f].astype(np.float64), slices, pairs, "partial") for f in range(2))
        eng = _core.SqEngine(q, sizes, pairs)
        eng.accumulate(pos)
        got = eng.result()
        eng.close()
        assert np.allclose(got, want, rtol=1e-6, atol=1e-9 * np.abs(want).max())
    # the lattice detector can be switched off; same numbers from the general kernel
    eng = _core.SqEngine(q_lat, sizes, pairs)
    eng.accumulate(pos)
    fast = eng.result()
    eng.close()
    monkeypatch.setenv("MDX_SQ_NO_LATTICE", "1")
    eng = _core.SqEngine(q_lat, sizes, pairs)
    eng.accumulate(pos)
    slow = eng.result()
    eng.close()
    assert np.allclose(fast, slow, rtol=1e-9, atol=1e-9 * np.abs(slow).max())
    # the reference's default grid size: 32^3 wavevectors (tables of 32 entries per axis)
    monkeypatch.delenv("MDX_SQ_NO_LATTICE")
    L = 30.0
    p = (rng.random((1, 500, 3)) * L).astype(np.float32)
    q = of.grid_wavevectors([L, L, L], 32)
    eng = _core.SqEngine(q, [500], ((None, None),))
    eng.accumulate(p)
    got = eng.result()[0]
    eng.close()
    rho = of.fourier_sum_ref(q[::97], p[0].astype(np.float64))
    assert np.allclose(got[::97], (rho * rho.conj()).real, rtol=1e-6, atol=1e-6)


def _dimers(n_pairs, L, r_lo, r_hi, seed):
    """Isolated pairs on a coarse grid (cells far apart) with separations swept over [r_lo, r_hi]."""
    rng = np.random.default_rng(seed)
    side = int(np.ceil(n_pairs ** (1 / 3)))
    grid = (np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n_pairs]
            + 0.5) * (L / side)
    u = rng.normal(size=(n_pairs, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    r = np.linspace(r_lo, r_hi, n_pairs)
    a = grid + rng.uniform(-0.2, 0.2, (n_pairs, 3))
    b = a + u * r[:, None]
    return np.mod(np.vstack((a, b)), L).astype(np.float32)


@pytest.mark.parametrize("algo", ["filter", "cell"])
@pytest.mark.parametrize("case", [
    # (range, n_bins): separations swept finely through both range ends, where the float32
    # filter has to hand every pair near r0 / r1 to the exact path.  The second and third
    # cases have r1 / width >> n_bins (narrow window far from the origin).
    ((0.0, 3.0), 150), ((2.5, 3.0), 150), ((2.9, 3.0), 512), ((0.0, 3.0), 3)])
def test_rdf_range_ends_swept_by_dimers(algo, case):
    (r0, r1), nb = case
    L = np.float32(140.0)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    parts = [_dimers(6000, float(L), r1 - 2e-4, r1 + 2e-4, 31)]
    if r0 > 0:
        parts.append(_dimers(6000, float(L), r0 - 2e-4, r0 + 2e-4, 32) + np.float32(3.0))
    pos = np.mod(np.vstack(parts), L).astype(np.float32)
    want = c_radial_histogram(pos, pos, nb, (r0, r1), dims, exclusion=(1, 1))
    for _ in range(3):     # the order inside a cell differs from run to run
        got = _gpu_hist(pos, None, nb, (r0, r1), dims, (1, 1), algo)
        assert np.array_equal(got, want), np.nonzero(got - want)[0]
    assert want[-1] > 1000


def test_rdf_cell_regression_pair_at_top_of_candidate_window():
    """
    A frame that used to gain two counts in bin 0 in ~85 % of the runs: one pair whose float32
    distance sat in the last 1e-6 A of the candidate window was called "sure" with index
    n_bins, which aliased the next per-wave histogram (slack budget).
    """
    rng = np.random.default_rng(12)
    F, N, L = 24, 3000, 31.0
    pos = rng.uniform(0, L, (1, N, 3)) + np.cumsum(rng.normal(0, 0.3, (F, N, 3)), axis=0)
    pos = np.mod(pos, L).astype(np.float32)[17:18]
    dims = np.array([[L + 0.17, L, L - 0.34, 90, 90, 90]], dtype=np.float32)
    want = c_radial_histogram(pos[0], pos[0], 150, (0.0, 12.0), dims[0], exclusion=(1, 1))
    for _ in range(10):
        got = _gpu_hist(pos, None, 150, (0.0, 12.0), dims, (1, 1), "cell")
        assert np.array_equal(got, want)


TRICLINIC = [(31.0, 28.5, 35.25, 75.0, 80.0, 110.0), (25.0, 25.0, 25.0, 60.0, 60.0, 90.0),
             (40.0, 22.0, 30.0, 90.0, 90.0, 120.0), (18.0, 30.0, 27.0, 101.5, 90.0, 67.25)]


@pytest.mark.parametrize("cell", TRICLINIC)
@pytest.mark.parametrize("exclusion", [None, (1, 1), (3, 3)])
def test_rdf_triclinic_cells(cell, exclusion):
    """Non-orthogonal cells: 27-image search on wrapped coordinates, bit-exact vs the C oracle."""
    rng = np.random.default_rng(44)
    dims = np.array(cell, dtype=np.float32)
    B = orf.triclinic_vectors(dims).astype(np.float64)
    pos = (rng.random((1100, 3)) @ B + rng.normal(0, 25.0, (1100, 3))).astype(np.float32)   # far outside the cell too
    for rng_range, nb in [((0.0, 9.0), 120), ((1.5, 8.0), 33)]:
        want = c_radial_histogram(pos, pos, nb, rng_range, dims, exclusion=exclusion)
        got = _gpu_hist(pos, None, nb, rng_range, dims, exclusion, "auto")
        assert np.array_equal(got, want), (cell, rng_range)
        assert want.sum() > 10000
    other = (rng.random((700, 3)) @ B).astype(np.float32)
    want = c_radial_histogram(pos, other, 64, (0.0, 7.0), dims)
    assert np.array_equal(_gpu_hist(pos, other, 64, (0.0, 7.0), dims, None, "cell"), want)


def test_rdf_mixed_orthorhombic_and_triclinic_frames_in_one_batch():
    rng = np.random.default_rng(45)
    F, N = 9, 1300
    boxes = np.array([[30, 30, 30, 90, 90, 90]] * 3 + [[30, 31, 29, 80, 95, 105]] * 2
                     + [[30, 30, 30, 90, 90, 90]] + [[28, 30, 33, 90, 70, 90]] * 3, dtype=np.float32)
    pos = (rng.random((F, N, 3)) * 30.0).astype(np.float32)
    want = np.zeros(150, dtype=np.int64)
    for f in range(F):
        want += c_radial_histogram(pos[f], pos[f], 150, (0.0, 10.0), boxes[f], exclusion=(1, 1))
    for algo in ("auto", "filter"):
        eng = _core.RdfEngine(_edges(150, (0.0, 10.0)), (1, 1), algo=algo)
        eng.accumulate(pos, None, boxes)
        assert np.array_equal(eng.counts(), want), algo
        # device-resident entry point: the boxes are inspected after a copy back
        eng.reset()
        d_pos = _core.DeviceArray.from_host(pos)
        d_box = _core.DeviceArray.from_host(boxes)
        eng.accumulate_device(d_pos.ptr, N, None, N, d_box.ptr, F)
        assert np.array_equal(eng.counts(), want), algo
        d_pos.free(); d_box.free()
        eng.close()


@pytest.mark.parametrize("case", [(70001, 1, 37, 0), (66000, 2, 20, 2), (131072, 1, 9, 5), (40000, 3, 11, 0),
                                  (140000, 1, 9, 0), (262144, 1, 6, 4), (150000, 2, 7, 0), (300000, 1, 5, 1),
                                  (524288, 1, 7, 0), (12500, 8, 30, 0), (8193, 3, 21, 2), (16384, 2, 40, 0),
                                  (16385, 5, 13, 4), (32768, 1, 19, 1), (25000, 4, 23, 0), (33000, 2, 11, 0),
                                  (2049, 7, 50, 0), (4096, 1, 33, 2), (5000, 20, 16, 0), (8192, 2, 25, 1),
                                  (100000, 1, 29, 0), (102400, 1, 8, 3), (102401, 1, 8, 0), (32769, 3, 17, 0),
                                  # 25 600 = 400 x 64 points: blocks of 8 193 .. 12 800 frames
                                  (12800, 2, 19, 0), (12801, 2, 19, 0), (9000, 11, 37, 6), (12500, 8, 16, 1),
                                  # single pass — 400, 800 = 400 x 2, 1 600 = 400 x 4 points: <= 200, 201 .. 400, 401 .. 800
                                  (400, 5, 23, 0), (401, 5, 23, 0), (201, 12, 70, 4), (800, 3, 11, 0), (500, 40, 9, 1),
                                  (200, 7, 23, 0), (199, 30, 41, 2), (64, 100, 33, 5), (37, 300, 16, 0), (2, 50, 9, 0),
                                  (1, 9, 7, 0), (300, 70, 129, 0), (640, 90, 64, 3), (400, 250, 48, 0),
                                  # 3 200 = 400 x 8: 801 .. 1 600
                                  (1600, 4, 21, 0), (1601, 4, 21, 0), (801, 7, 30, 3), (1000, 50, 9, 0),
                                  # 6 400 = 400 x 16 and 12 800 = 400 x 32 points: 2 049 .. 3 200 and 4 097 .. 6 400
                                  (3200, 3, 25, 0), (3201, 3, 25, 0), (2049, 9, 40, 2), (3000, 1, 8, 5),
                                  (6400, 2, 33, 0), (6401, 2, 33, 0), (4097, 5, 17, 1), (6250, 16, 10, 0),
                                  # 51 200 = 400 x 128 and 102 400 = 400 x 256 points: 16 385 .. 25 600 and 32 769 .. 51 200
                                  (25600, 1, 21, 0), (25601, 1, 9, 3), (20000, 3, 43, 0), (16385, 2, 32, 4),
                                  (51200, 1, 11, 0), (51201, 1, 7, 0), (45000, 2, 29, 5), (32769, 1, 64, 0),
                                  # 409 600 = 400 x 1024 points: 131 073 .. 204 800
                                  (204800, 1, 7, 0), (204801, 1, 5, 0), (131073, 1, 9, 2), (160000, 2, 6, 0),
                                  # rows of whole 128-byte lines (16 | 3 n_atoms): the second push starts 9
                                  # coordinates into a line and pass A enters its chunk early (head)
                                  (70001, 1, 16, 0), (40000, 2, 32, 5), (50000, 1, 48, 2)])
def test_msd_own_two_pass_transform_equals_rocfft(case, monkeypatch):
    """n_fft = 2^13, 2^14, 25 600, 2^15, 51 200, 2^16, 102 400, 204 800, 2^18, 2^19, 2^20: the engine's own packed
    two-pass transforms against the rocFFT pipeline."""
    t_block, n_blocks, n_atoms, zero_dims = case
    rng = np.random.default_rng(7)
    T = t_block * n_blocks
    pos = np.cumsum(rng.normal(0, 0.3, (T, n_atoms, 3)), axis=0) + rng.uniform(0, 50, (1, n_atoms, 3))
    out = {}
    for mode in ("own", "rocfft"):
        if mode == "rocfft":
            monkeypatch.setenv("MDX_MSD_ROCFFT", "1")
        else:
            monkeypatch.delenv("MDX_MSD_ROCFFT", raising=False)
        eng = _core.MsdEngine(t_block, n_blocks, 2)
        want = (400 if t_block <= 200 else 800 if t_block <= 400 else 1600 if t_block <= 800 else 3200 if t_block <= 1600 else 6400 if t_block <= 3200 else 8192 if t_block <= 4096 else 12800 if t_block <= 6400
                else 16384 if t_block <= 8192 else 25600 if t_block <= 12800
                else 32768 if t_block <= 16384 else 51200 if t_block <= 25600
                else 65536 if t_block <= 32768 else 102400 if t_block <= 51200 else 204800 if t_block <= 102400
                else 262144 if t_block <= 131072 else 409600 if t_block <= 204800
                else 524288 if t_block <= 262144 else 1048576)
        assert mode != "own" or eng.n_fft == want     # rocFFT runs its own choice of length
        own, r1, r2 = eng.transform
        assert own == (mode == "own") and (r1 * r2 == eng.n_fft if own else (r1, r2) == (0, 0))
        eng.push(0, pos, 0, n_atoms, zero_dims)
        eng.push(1, pos, 3, n_atoms - 5, zero_dims)
        out[mode] = eng.result()
        eng.close()
    for a, b in zip(out["own"], out["rocfft"]):
        # the two pipelines add up the per-frame sums D_t in different orders; at the last lags
        # S_m = (2 sum D - cumulative sums) / (T - m) cancels ~10 digits of them, so a few 1e-10 of
        # the largest value is the rounding floor of either
        # (a block of ONE frame has lag 0 only: S_0 - 2 A_0 = 0 up to a few ulps of sum x^2 in either pipeline)
        scale = np.abs(b).max()
        assert np.allclose(a, b, rtol=1e-10, atol=5e-10 * scale + 2e-14 * float((pos ** 2).sum()) / n_blocks)
    # and against the direct definition on a few lags
    msd = out["own"][0][0, 0] / n_atoms
    keep = [k for k in range(3) if not (zero_dims >> k) & 1]
    p = pos[:t_block][:, :, keep]
    for m in sorted({m for m in (1, 17, min(4096, t_block // 2), t_block - 3) if 0 < m < t_block}):
        d = p[m:] - p[:-m]
        assert np.isclose(msd[m], (d * d).sum(-1).mean(), rtol=1e-8)


@pytest.mark.parametrize("case", [(400, 6, 27, 0), (201, 9, 70, 4), (800, 3, 11, 1), (401, 14, 33, 0), (777, 5, 16, 2)])
def test_msd_single_pass_equals_the_two_pass_pipeline(case, monkeypatch):
    """800 and 1 600 points: the single-pass kernel (no `Y`) against the two-pass pipeline it replaced for blocks
    of 201 .. 800 frames (MDX_MSD_TWO_PASS=1), same engine, same pushes."""
    t_block, n_blocks, n_atoms, zero_dims = case
    rng = np.random.default_rng(8)
    pos = np.cumsum(rng.normal(0, 0.3, (t_block * n_blocks, n_atoms, 3)), axis=0) + rng.uniform(0, 50, (1, n_atoms, 3))
    out = {}
    for mode in ("single", "two_pass"):
        if mode == "two_pass":
            monkeypatch.setenv("MDX_MSD_TWO_PASS", "1")
        else:
            monkeypatch.delenv("MDX_MSD_TWO_PASS", raising=False)
        eng = _core.MsdEngine(t_block, n_blocks, 2)
        assert eng.n_fft == (800 if t_block <= 400 else 1600)
        eng.push(0, pos, 0, n_atoms, zero_dims)
        eng.push(1, pos, 2, n_atoms - 3, zero_dims)
        out[mode] = eng.result()
        eng.close()
    for a, b in zip(out["single"], out["two_pass"]):
        assert np.allclose(a, b, rtol=1e-10, atol=5e-10 * np.abs(b).max())


@pytest.mark.parametrize("mode", [None, "partial"])
def test_isf_lattice_tables_equal_general_sincos_path(mode, monkeypatch):
    """Grid wavevectors: separable phase tables (coherent part; incoherent part through the
    register-blocked column kernel and through the per-wavevector tables, MDX_ISF_NO_QUADS)
    against the general fp64 sincos kernels, and all against the restated driver on a subset."""
    rng = np.random.default_rng(71)
    F, sizes, L = 14, (2600, 1900), 33.0
    N = sum(sizes)
    pos = np.mod(rng.uniform(0, L, (1, N, 3)) + np.cumsum(rng.normal(0, 0.25, (F, N, 3)), axis=0), L).astype(np.float32)
    q = of.grid_wavevectors([L, L, L], 4)[1:]            # drop q = 0: a q_max-style subset
    pairs = of.ssf_pairs(2, mode)
    out = {}
    for kind in ("lattice", "tables", "general"):
        monkeypatch.delenv("MDX_SQ_NO_LATTICE", raising=False)
        monkeypatch.delenv("MDX_ISF_NO_QUADS", raising=False)
        if kind == "general":
            monkeypatch.setenv("MDX_SQ_NO_LATTICE", "1")
        elif kind == "tables":
            monkeypatch.setenv("MDX_ISF_NO_QUADS", "1")
        eng = _core.IsfEngine(q, sizes if mode else [N], pairs, 6, True)
        eng.accumulate(pos[:5])
        eng.accumulate(pos[5:])
        out[kind] = eng.result()
        eng.close()
    for kind in ("lattice", "tables"):
        for a, b in zip(out[kind], out["general"]):
            assert np.allclose(a, b, rtol=1e-9, atol=1e-9 * np.abs(b).max()), kind
    ref = of.isf_run_ref(pos, sizes, q, 6, mode=mode, incoherent=True, sort=False, unique=False)
    norm = N * np.arange(F, F - 6, -1)[:, None, None]
    assert np.allclose(out["lattice"][0] / norm, ref["cisf"], rtol=1e-6, atol=1e-9 * np.abs(ref["cisf"]).max())
    assert np.allclose(out["lattice"][1] / norm, ref["iisf"], rtol=1e-6, atol=1e-9)


def test_sq_column_form_equals_lattice_and_general_kernels(monkeypatch):
    """Grid wavevector sets through the four S(q) kernels: register-blocked columns (default),
    column form (MDX_SQ_NO_QUADS), per-q lattice tables (MDX_SQ_NO_COLUMNS) and general fp64
    sincos (MDX_SQ_NO_LATTICE)."""
    rng = np.random.default_rng(81)
    F, sizes, L = 5, (1700, 1301), np.array([31.0, 29.5, 33.25])
    N = sum(sizes)
    pos = (rng.random((F, N, 3)) * L).astype(np.float32)

    def grid_of(lo, hi):
        return np.stack(np.meshgrid(*[2 * np.pi * np.arange(lo, hi) / x for x in L], indexing="ij"),
                        -1).reshape(-1, 3)

    grid = grid_of(-3, 4)
    big = grid_of(-6, 7)
    sets = {"full 7^3 grid": grid,
            "positive octant 8^3 (one item copy per 16 threads)": grid_of(0, 8),
            "13^3 grid (two m_z chunks, 86 items)": big,
            "21 x 21 x 9 grid (more than 256 items: several blocks)":
                np.stack(np.meshgrid(2 * np.pi * np.arange(-10, 11) / L[0], 2 * np.pi * np.arange(-10, 11) / L[1],
                                     2 * np.pi * np.arange(0, 9) / L[2], indexing="ij"), -1).reshape(-1, 3),
            # the reference's grids and their q_max subsets (m >= 0: structure.py:1376-1381, 1412-1414): regular quad
            # items from aligned 4 x 8 blocks, entries outside the set dropped; item counts that are no power of two
            "full 10^3 grid (rows padded to 12 / 16)": grid_of(0, 10),
            "octant sphere of the 12^3 grid (q_max)": (lambda g: g[np.linalg.norm(g, axis=1) <= 2.2])(grid_of(0, 12)),
            "octant sphere of the 20^3 grid (several blocks)":
                (lambda g: g[np.linalg.norm(g, axis=1) <= 3.4])(grid_of(0, 20)),
            "grid from m = 2 (tables start at m = 0)": grid_of(2, 9),
            "5 x 9 x 17 grid (95 items, unequal axes)":
                np.stack(np.meshgrid(2 * np.pi * np.arange(0, 5) / L[0], 2 * np.pi * np.arange(0, 9) / L[1],
                                     2 * np.pi * np.arange(0, 17) / L[2], indexing="ij"), -1).reshape(-1, 3),
            "sphere |q| < 0.5": grid[np.linalg.norm(grid, axis=1) < 0.5],
            "sphere of the 13^3 grid": big[np.linalg.norm(big, axis=1) < 1.1],
            "ragged columns": grid[rng.random(len(grid)) < 0.6]}
    pairs = of.ssf_pairs(2, "partial")
    envs = ("MDX_SQ_NO_QUADS", "MDX_SQ_NO_COLUMNS", "MDX_SQ_NO_LATTICE")
    for name, q in sets.items():
        out = {}
        for kind, env in (("quads", {}), ("columns", {"MDX_SQ_NO_QUADS": "1"}),
                          ("lattice", {"MDX_SQ_NO_COLUMNS": "1"}), ("general", {"MDX_SQ_NO_LATTICE": "1"})):
            for k in envs:
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            eng = _core.SqEngine(q, sizes, pairs)
            eng.accumulate(pos)
            out[kind] = eng.result()
            eng.close()
        scale = np.abs(out["general"]).max()
        for kind in ("quads", "columns", "lattice"):
            assert np.allclose(out[kind], out["general"], rtol=1e-9, atol=1e-9 * scale), (name, kind)
    for k in envs:
        monkeypatch.delenv(k, raising=False)
    slices = [slice(0, sizes[0]), slice(sizes[0], N)]
    ref = sum(of.ssf_frame_ref(q, pos[f].astype(np.float64), slices, pairs, "partial") for f in range(F))
    assert np.allclose(out["quads"], ref, rtol=1e-6, atol=1e-9 * np.abs(ref).max())
    # groups smaller than one table tile, and a single particle
    for tiny in ((3, 1), (50, 17)):
        n = sum(tiny)
        eng = _core.SqEngine(grid, tiny, pairs)
        eng.accumulate(pos[:, :n])
        got = eng.result()
        eng.close()
        sl = [slice(0, tiny[0]), slice(tiny[0], n)]
        ref = sum(of.ssf_frame_ref(grid, pos[f, :n].astype(np.float64), sl, pairs, "partial") for f in range(F))
        assert np.allclose(got, ref, rtol=1e-6, atol=1e-9 * np.abs(ref).max()), tiny


@pytest.mark.parametrize("cell", [(42.0, 40.5, 45.25, 75.0, 80.0, 110.0), (36.0, 36.0, 36.0, 60.0, 60.0, 90.0),
                                  (50.0, 38.0, 41.0, 90.0, 90.0, 120.0), (33.0, 46.0, 39.0, 101.5, 90.0, 67.25)])
def test_rdf_triclinic_culled_kernel(cell, monkeypatch):
    """Triclinic cells with the cut below half the smallest cell height: the cell-sorted kernel
    with 27 tile images (float32 filter + 27-image contract for undecided pairs) against the C
    oracle and against the brute-force 27-image kernel."""
    rng = np.random.default_rng(46)
    dims = np.array(cell, dtype=np.float32)
    B = orf.triclinic_vectors(dims).astype(np.float64)
    n = 5200
    pos = (rng.random((n, 3)) @ B + rng.normal(0, 30.0, (n, 3))).astype(np.float32)
    other = (rng.random((2300, 3)) @ B).astype(np.float32)
    for rng_range, nb, exclusion in [((0.0, 9.0), 120, (1, 1)), ((1.5, 8.0), 33, None), ((0.0, 6.0), 201, (4, 4))]:
        want = c_radial_histogram(pos, pos, nb, rng_range, dims, exclusion=exclusion)
        monkeypatch.delenv("MDX_RDF_TRI_BRUTE", raising=False)
        eng = _core.RdfEngine(_edges(nb, rng_range), exclusion, timing=True)
        eng.accumulate(pos, None, dims)
        got = eng.counts()
        st = eng.stats()
        eng.close()
        assert np.array_equal(got, want), (cell, rng_range)
        assert 0 < st["pairs_computed"] < 0.6 * st["pairs_evaluated"]      # the culled path ran
        monkeypatch.setenv("MDX_RDF_TRI_BRUTE", "1")
        assert np.array_equal(_gpu_hist(pos, None, nb, rng_range, dims, exclusion, "auto"), want)
    monkeypatch.delenv("MDX_RDF_TRI_BRUTE", raising=False)
    want = c_radial_histogram(pos, other, 64, (0.0, 7.0), dims)
    assert np.array_equal(_gpu_hist(pos, other, 64, (0.0, 7.0), dims, None, "auto"), want)
    # a range beyond half the smallest height falls back to the brute-force kernel
    want = c_radial_histogram(pos[:1500], pos[:1500], 50, (0.0, 22.0), dims, exclusion=(1, 1))
    assert np.array_equal(_gpu_hist(pos[:1500], None, 50, (0.0, 22.0), dims, (1, 1), "auto"), want)


@pytest.mark.parametrize("units", ["1", "40", "1000"])
def test_rdf_persistent_blocks_flush_their_lds_bins_between_items(units, monkeypatch):
    """The persistent pair kernel keeps 32-bit LDS bins across the items a block serves and flushes them to
    the 64-bit replicas before a bin could wrap (2^18 j tiles of 2^14 possible adds; at the BASELINE sizes
    that happens once per block and launch at C5 only).  MDX_RDF_LDS_FLUSH_UNITS lowers the bound so that
    small inputs flush before every round ("1"), every few items, or now and then: 160 frames of 2 500
    particles = 3 200 items on 1 792 blocks, two groups, counts identical with the C oracle."""
    monkeypatch.setenv("MDX_RDF_LDS_FLUSH_UNITS", units)
    rng = np.random.default_rng(77)
    F, n1, n2, L = 160, 2500, 1500, np.float32(29.0)
    a = (rng.random((F, n1, 3)) * L).astype(np.float32)
    b = (rng.random((F, n2, 3)) * L).astype(np.float32)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    want_self = np.zeros(64, dtype=np.int64)
    want_cross = np.zeros(64, dtype=np.int64)
    for f in range(0, F, 16):            # the oracle on every 16th frame; the GPU on the same ten
        want_self += c_radial_histogram(a[f], a[f], 64, (0.0, 9.0), dims, exclusion=(1, 1))
        want_cross += c_radial_histogram(a[f], b[f], 64, (0.0, 9.0), dims)
    sub = np.arange(0, F, 16)
    eng = _core.RdfEngine(_edges(64, (0.0, 9.0)), (1, 1), algo="cell")
    # all 160 frames in one launch (many items per block), then minus the 150 the oracle skipped: linearity
    eng.accumulate(a, None, dims)
    all_counts = eng.counts()
    eng.reset()
    rest = np.setdiff1d(np.arange(F), sub)
    eng.accumulate(a[rest], None, dims)
    assert np.array_equal(all_counts - eng.counts(), want_self)
    eng.close()
    eng = _core.RdfEngine(_edges(64, (0.0, 9.0)), None, algo="cell")
    eng.accumulate(a, b, dims)
    all_counts = eng.counts()
    eng.reset()
    eng.accumulate(a[rest], b[rest], dims)
    assert np.array_equal(all_counts - eng.counts(), want_cross)
    eng.close()


@pytest.mark.parametrize("t_block,n_blocks", [(150, 9), (400, 5), (777, 4), (801, 3), (4000, 3), (5000, 2), (8000, 2),
                                              (16384, 1), (30000, 1), (40000, 1)])
def test_msd_pass_a_reads_float32_frames_in_place(t_block, n_blocks):
    """mdx_msd_push_device_f32: float32 frames resident in HBM, a plain particle range — pass A of the 400 x R2 transforms
    widens them as it stages them.  Bit for bit what the widened float64 frames give through mdx_msd_push_device
    (zeroed dimension, a range that starts inside a 128-byte line, two groups); engines whose transforms do not read
    float32 say so."""
    rng = np.random.default_rng(t_block)
    n, T = 37, t_block * n_blocks
    pos32 = (rng.uniform(0, 30, (1, n, 3)) + np.cumsum(rng.normal(0, 0.2, (T, n, 3)), axis=0)).astype(np.float32)
    d32 = _core.DeviceArray.from_host(pos32)
    d64 = _core.DeviceArray.from_host(pos32.astype(np.float64))
    out = []
    for f32 in (False, True):
        eng = _core.MsdEngine(t_block, n_blocks, 2)
        assert eng.reads_f32
        for g, (first, count, zero) in enumerate(((0, 21, 0), (21, 16, 2))):
            if f32:
                eng.push_device_f32(g, d32.ptr, n, first, count, zero)
            else:
                eng.push_device(g, d64.ptr, n, first, count, zero)
        out.append(eng.result())
        eng.close()
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    other = _core.MsdEngine(110000, 1, 1)            # 2^18 = 512 x 512: no float32 route
    assert not other.reads_f32
    with pytest.raises(NotImplementedError):
        other.push_device_f32(0, d32.ptr, n, 0, n)
    other.close()
    d32.free()
    d64.free()
