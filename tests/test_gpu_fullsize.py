"""
GPU tests at BASELINE.json's full sizes through size-independent properties (the oracle
cannot finish these sizes in seconds): conservation of the pair count, bit-exact
agreement of the three RDF algorithms, permutation invariance, frame additivity and
group decomposition of the integer histogram; lattice sums for S(q); FFT MSD against the
direct definition at a few lags and the free-walk slope at long trajectories.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from mdhelper_amd import _core  # noqa: E402

N = 32768
L = np.float32(68.94)
DIMS = np.array([L, L, L, 90, 90, 90], dtype=np.float32)


def _frames(n_frames, seed=2):
    d = _core.synth_random_walk(n_frames, N, [L, L, L], 0.3, seed=seed)
    out = d.to_host()
    d.free()
    return out


def _hist(pos1, pos2, edges, exclusion, algo):
    eng = _core.RdfEngine(edges, exclusion, algo=algo)
    eng.accumulate(pos1, pos2, DIMS)
    out = eng.counts()
    eng.close()
    return out


def test_c2_algorithms_agree_bit_for_bit():
    """32 768 atoms: contract arithmetic on every pair == float32 filter == cell-sorted path."""
    frames = _frames(2)
    for rng_range in [(0.0, 15.0), (0.0, float(L) / 2)]:
        edges = np.linspace(*rng_range, 202)
        ref = _hist(frames, None, edges, (1, 1), "exact")
        assert np.array_equal(_hist(frames, None, edges, (1, 1), "filter"), ref)
        assert np.array_equal(_hist(frames, None, edges, (1, 1), "cell"), ref)
        assert ref.sum() > 0


def test_c2_every_pair_is_counted_once():
    """A range beyond the largest minimum-image distance bins all N(N-1) ordered pairs."""
    frames = _frames(1, seed=3)
    edges = np.linspace(0.0, 60.0, 121)            # L * sqrt(3) / 2 = 59.7
    for algo in ("cell", "filter"):
        counts = _hist(frames, None, edges, (1, 1), algo)
        assert counts.sum() == N * (N - 1), algo
    # without the exclusion the N self pairs (d = 0) join bin 0
    counts0 = _hist(frames, None, edges, None, "cell")
    assert counts0.sum() == N * N and counts0[0] - counts[0] == N


def test_c2_permutation_additivity_and_group_decomposition():
    frames = _frames(3, seed=4)
    edges = np.linspace(0.0, 15.0, 202)
    whole = _hist(frames, None, edges, (1, 1), "cell")
    # frame additivity
    parts = sum(_hist(frames[f:f + 1], None, edges, (1, 1), "cell") for f in range(3))
    assert np.array_equal(whole, parts)
    # permutation invariance of the particle order (self pairs excluded by identity)
    perm = np.random.default_rng(0).permutation(N)
    assert np.array_equal(_hist(frames[:, perm], None, edges, (1, 1), "cell"), whole)
    # g(A+B, A+B) = g(A,A) + g(B,B) + g(A,B) + g(B,A)
    a, b = np.ascontiguousarray(frames[:, :20000]), np.ascontiguousarray(frames[:, 20000:])
    total = (_hist(a, None, edges, (1, 1), "cell") + _hist(b, None, edges, (1, 1), "cell")
             + _hist(a, b, edges, None, "cell") + _hist(b, a, edges, None, "cell"))
    assert np.array_equal(total, whole)
    # the same histogram whatever images the coordinates are given in
    shifted = frames + (np.float32(2) * L)        # exact in float32 for these magnitudes? no:
    # rounding moves coordinates by < 1e-5 A, so only near-edge pairs may change bins
    moved = _hist(shifted.astype(np.float32), None, edges, (1, 1), "cell")
    assert np.abs(moved - whole).sum() <= 1e-4 * whole.sum()


def test_c3_structure_factor_of_a_lattice_at_full_size():
    """32 768 = 32^3 particles on a simple-cubic lattice: N at Bragg points, 0 elsewhere."""
    n = 32
    a = float(L) / n
    idx = np.arange(n)
    pos = (a * np.stack(np.meshgrid(idx, idx, idx, indexing="ij"), -1).reshape(-1, 3)).astype(np.float32)
    grid = 2 * np.pi * np.arange(8) / float(L)
    q = np.stack(np.meshgrid(grid, grid, grid), -1).reshape(-1, 3)       # 512 wavevectors
    q_bragg = 2 * np.pi / a * np.array([[1, 0, 0], [0, 1, 1], [1, 1, 1]], dtype=float)
    for wavevectors in (q, np.vstack((q, q_bragg))):
        eng = _core.SqEngine(wavevectors, [N // 2, N // 2], ((None, None),))
        eng.accumulate(pos[None])
        ssf = eng.result()[0] / N
        eng.close()
        assert np.isclose(ssf[0], N)                      # q = 0
        assert np.allclose(ssf[1:512], 0.0, atol=1e-5)    # no Bragg point below 2 pi / a on this grid
        if len(wavevectors) > 512:
            assert np.allclose(ssf[512:], N, rtol=1e-6)


def test_c4_long_trajectory_msd():
    """T = 100 000 frames: FFT MSD equals the direct definition at chosen lags; free-walk slope."""
    T, n = 100_000, 64
    d = _core.synth_random_walk(T, n, [1.0, 1.0, 1.0], 0.1, seed=7, dtype=np.float64)
    eng = _core.MsdEngine(T, 1, 1)
    eng.push_device(0, d.ptr, n, 0, n)
    msd_sum, traj = eng.result()
    eng.close()
    pos = d.to_host()
    d.free()
    msd = msd_sum[0, 0] / n
    for m in (1, 7, 1000, 50_000, 99_999):
        direct = ((pos[m:] - pos[:-m]) ** 2).sum(axis=-1).mean()
        assert np.isclose(msd[m], direct, rtol=1e-6), m
    assert abs(msd[0]) < 1e-6
    m = np.arange(1, 2000)
    assert np.allclose(msd[1:2000] / (3 * 0.01 * m), 1.0, atol=0.05)
    assert np.allclose(traj[0, 0], pos.sum(axis=1), rtol=1e-12, atol=1e-9)


def test_c5_size_cell_path_equals_filter_path():
    """131 072 atoms (BASELINE C5): 2 048 j tiles per frame, i.e. two rounds of the survivor
    queue; the culled kernel must agree bit for bit with the brute-force float32-filter tiles,
    and the contract-arithmetic kernel on a sub-block."""
    n = 131072
    Lc = np.float32(109.4)
    dims = np.array([Lc, Lc, Lc, 90, 90, 90], dtype=np.float32)
    d = _core.synth_random_walk(2, n, [Lc, Lc, Lc], 0.3, seed=5)
    frames = d.to_host()
    d.free()
    edges = np.linspace(0.0, 15.0, 202)
    out = {}
    for algo in ("cell", "filter"):
        eng = _core.RdfEngine(edges, (1, 1), algo=algo)
        eng.accumulate(frames, None, dims)
        out[algo] = eng.counts()
        eng.close()
    assert np.array_equal(out["cell"], out["filter"]) and out["cell"].sum() > 0
    # density 0.1 / A^3: about 4/3 pi 15^3 * 0.1 = 1414 neighbours per atom
    assert abs(out["cell"].sum() / (2 * n) - 1413.7) < 15
    sub = frames[:1, :20000]
    eng = _core.RdfEngine(edges, (1, 1), algo="exact")
    eng.accumulate(sub, None, dims)
    ref = eng.counts()
    eng.close()
    eng = _core.RdfEngine(edges, (1, 1), algo="cell")
    eng.accumulate(sub, None, dims)
    assert np.array_equal(eng.counts(), ref)
    eng.close()


@pytest.mark.parametrize("n", [400_000, 1_000_000])
def test_beyond_c5_cell_path_equals_filter_path(n):
    """Far past the BASELINE sizes (10^6 particles: 10^12 ordered pairs in one frame, 7 813 tiles):
    the cell-sorted culled kernel and the brute-force float32-filter tiles bin the same counts."""
    rng = np.random.default_rng(6)
    L = np.float32((n / 0.1) ** (1 / 3))
    pos = (rng.random((1, n, 3)) * L).astype(np.float32)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    edges = np.linspace(0.0, 15.0, 202)
    out = {}
    for algo in ("auto", "filter"):
        eng = _core.RdfEngine(edges, (1, 1), algo=algo)
        eng.accumulate(pos, None, dims)
        out[algo] = eng.counts()
        eng.close()
    assert np.array_equal(out["auto"], out["filter"])
    assert abs(out["auto"].sum() / n - 1413.7) < 10


# ----------------------------------------------------------------------------------------------
# Oracle parity AT the BASELINE sizes (the CPU oracle finishes these single frames in seconds)
# ----------------------------------------------------------------------------------------------

def test_c2_full_size_frame_equals_the_c_oracle():
    """C2: one 32 768-atom bench frame, default path (auto = cell-sorted culled kernel) and explicit
    cell, ranges (0, 15) and (0, L/2): u64 counts bit-identical with oracle/c/rdf_oracle.c."""
    from oracle.cbind import c_radial_histogram
    frames = _frames(2)          # the bench generator (seed 2): frame 0 uniform, frame 1 one walk step
    for rng_range in [(0.0, 15.0), (0.0, float(L) / 2)]:
        edges = np.linspace(*rng_range, 202)
        for f in range(2 if rng_range[1] == 15.0 else 1):
            want = c_radial_histogram(frames[f], frames[f], 201, rng_range, DIMS, exclusion=(1, 1), n_threads=16)
            assert want.sum() > 4e7
            for algo in ("auto", "cell"):
                got = _hist(frames[f:f + 1], None, edges, (1, 1), algo)
                assert np.array_equal(got, want), (rng_range, f, algo)


def _assert_partial_ssf_close(got, ref, pairs, group_sizes, n_frames, rel=1e-6):
    """Element-wise 1e-6 on every entry of every pair column — no norm over the column, whose q = 0
    entry (n_frames N_j N_k) is four orders of magnitude above a typical one.  Diagonal columns
    |rho_j|^2: relative, plus 1e-9 N_j per frame for entries that cancel to ~0 (lattices).  Cross
    columns 2 Re(rho_j rho_k*) may pass through zero; the rounding of each factor is relative to
    |rho_j| |rho_k|, so the bound there is rel * sqrt(S_jj S_kk) (Cauchy-Schwarz over the frames)."""
    diag = {p[0]: ref[i] for i, p in enumerate(pairs) if p[0] == p[1]}
    for i, (j, k) in enumerate(pairs):
        err = np.abs(got[i] - ref[i])
        if j == k:
            bound = rel * np.abs(ref[i]) + 1e-9 * group_sizes[j] * n_frames
        else:
            bound = rel * np.sqrt(diag[j] * diag[k]) + 1e-9 * np.sqrt(group_sizes[j] * group_sizes[k]) * n_frames
        worst = int(np.argmax(err - bound))
        assert np.all(err <= bound), (i, worst, float(err[worst]), float(bound[worst]), float(ref[i][worst]))


def test_c3_full_size_random_frame_equals_the_numpy_oracle():
    """C3: 32 768 randomly placed atoms (not a lattice), the 512 grid wavevectors, mode="partial":
    every entry of all three pair columns within 1e-6 of oracle/fourier.py, element by element."""
    from oracle import fourier as of
    frames = _frames(2, seed=11)
    grid = 2 * np.pi * np.arange(8) / float(L)
    q = np.stack(np.meshgrid(grid, grid, grid), -1).reshape(-1, 3)
    sizes = [N // 2, N - N // 2]
    pairs = of.ssf_pairs(2, "partial")
    slices = [slice(0, N // 2), slice(N // 2, N)]
    eng = _core.SqEngine(q, sizes, pairs)
    eng.accumulate(frames)
    got = eng.result()
    eng.close()
    ref = sum(of.ssf_frame_ref(q, frames[f].astype(np.float64), slices, pairs, "partial") for f in range(2))
    assert got.shape == ref.shape == (3, 512)
    _assert_partial_ssf_close(got, ref, pairs, sizes, 2)
    # and a non-lattice wavevector set of the same size through the general sincos kernel
    qg = q + np.random.default_rng(0).normal(scale=1e-3, size=q.shape)
    eng = _core.SqEngine(qg, sizes, pairs)
    eng.accumulate(frames[:1])
    got = eng.result()
    eng.close()
    ref = of.ssf_frame_ref(qg, frames[0].astype(np.float64), slices, pairs, "partial")
    _assert_partial_ssf_close(got, ref, pairs, sizes, 1)


def _direct_msd_sum(d, n_total, first, count, t0, t_len, lag):
    """sum over particles [first, first+count) and over origins of |r(t+lag) - r(t)|^2 within the block
    [t0, t0+t_len) — the direct definition (correlation.py:670-850), from the device array in slabs."""
    total, slab = 0.0, 4000
    n_orig = t_len - lag
    for a in range(0, n_orig, slab):
        b = min(n_orig, a + slab)
        x0 = d.to_host(t0 + a, b - a)[:, first:first + count]
        x1 = d.to_host(t0 + a + lag, b - a)[:, first:first + count]
        total += float(((x1 - x0) ** 2).sum())
    return total / n_orig


@pytest.mark.parametrize("n_blocks", [1, 8])
def test_c4_full_size_msd(n_blocks):
    """C4 at its real size: 10 000 particles x 100 000 frames resident in HBM, two groups of 5 000
    pushed as bench_msd does (whole-group chunks, the 31.5 GB half-transformed buffer, the parts
    rule), n_blocks = 1 and 8.  Checked against the direct definition (a) for ALL particles of each
    group at large lags (few origins: cheap), (b) for 32 sampled particles at small and large lags
    through a second engine on the same HBM array, plus (c) linearity over sub-pushes, (d) the summed
    trajectories, (e) the free-walk slope."""
    T, n = 100_000, 10_000
    sigma = 0.1
    d = _core.synth_random_walk(T, n, [1.0, 1.0, 1.0], sigma, seed=4, dtype=np.float64)
    tb = T // n_blocks
    groups = [(0, n // 2), (n // 2, n - n // 2)]
    eng = _core.MsdEngine(tb, n_blocks, 2)
    for g, (first, count) in enumerate(groups):
        eng.push_device(g, d.ptr, n, first, count)
    msd, traj = eng.result()
    eng.close()
    assert msd.shape == (2, n_blocks, tb)
    # (a) all particles, large lags, first and last block
    for g, (first, count) in enumerate(groups):
        for b in sorted({0, n_blocks - 1}):
            for lag in (tb - 1, tb - 10, tb - 1000):
                direct = _direct_msd_sum(d, n, first, count, b * tb, tb, lag)
                assert np.isclose(msd[g, b, lag], direct, rtol=1e-6), (g, b, lag)
    # (e) MSD(m) = 3 sigma^2 m per particle
    m = np.arange(1, 200)
    assert np.allclose(msd[0, 0, 1:200] / (n // 2) / (3 * sigma ** 2 * m), 1.0, atol=0.01)
    assert np.all(np.abs(msd[:, :, 0]) < 1e-6 * msd[:, :, 1])
    # (d) summed trajectories at sampled frames
    for t in (0, 1, tb - 1, T - tb, T - 1):
        row = d.to_host(t, 1)[0]
        b, k = divmod(t, tb)
        if b < n_blocks:
            assert np.allclose(traj[0, b, k], row[:n // 2].sum(axis=0), rtol=1e-12, atol=1e-9)
            assert np.allclose(traj[1, b, k], row[n // 2:].sum(axis=0), rtol=1e-12, atol=1e-9)
    # (b) 32 sampled particles (atoms 0..31; the generator is keyed by the atom index, so a 32-atom walk
    # with the same seed IS their trajectory — checked on two frames) against the direct definition
    s = _core.synth_random_walk(T, 32, [1.0, 1.0, 1.0], sigma, seed=4, dtype=np.float64)
    pos = s.to_host()
    s.free()
    assert np.array_equal(pos[0], d.to_host(0, 1)[0, :32]) and np.array_equal(pos[T - 1], d.to_host(T - 1, 1)[0, :32])
    small = _core.MsdEngine(tb, n_blocks, 1)
    small.push_device(0, d.ptr, n, 0, 32)
    msd32 = small.result()[0][0]
    small.close()
    for b in sorted({0, n_blocks - 1}):
        blk = pos[b * tb:(b + 1) * tb]
        for lag in (1, 7, 1000, tb // 2, tb - 1):
            direct = ((blk[lag:] - blk[:-lag]) ** 2).sum(axis=(1, 2)).mean()
            assert np.isclose(msd32[b, lag], direct, rtol=1e-6), (b, lag)
    # (c) linearity: group 0 as five sub-pushes of 1 000 particles == one push of 5 000
    parts = _core.MsdEngine(tb, n_blocks, 1)
    for k in range(5):
        parts.push_device(0, d.ptr, n, 1000 * k, 1000)
    msd_parts = parts.result()[0][0]
    parts.close()
    scale = np.abs(msd[0]).max(axis=-1, keepdims=True)
    assert np.abs(msd_parts - msd[0]).max() <= 1e-9 * scale.max()
    d.free()


def _direct_msd_sums(h, halves, t0, t_len, lag, n_threads=8):
    """[sum over the particles of each range in `halves` and over origins of |r(t+lag) - r(t)|^2] / origins
    within the block [t0, t0+t_len) of the host float32 trajectory h — the direct definition
    (correlation.py:670-850) in float64, slabs of frames on a few threads."""
    from concurrent.futures import ThreadPoolExecutor
    n_orig = t_len - lag
    slab = max(1, min(n_orig, (48 << 20) // (h.shape[1] * 12)))

    def part(a):
        b = min(n_orig, a + slab)
        d = np.subtract(h[t0 + a + lag:t0 + b + lag], h[t0 + a:t0 + b], dtype=np.float64)
        return [float(np.einsum("tak,tak->", d[:, lo:hi], d[:, lo:hi])) for lo, hi in halves]

    with ThreadPoolExecutor(n_threads) as pool:
        parts = list(pool.map(part, range(0, n_orig, slab)))
    return np.sum(parts, axis=0) / n_orig


@pytest.mark.parametrize("n_blocks", [1, 8])
def test_c4_onsager_and_transport_coefficients_end_to_end(n_blocks):
    """BASELINE C4 as worded — "MSD + Onsager transport coeffs, 10k atoms x 100k frames" — through the
    operator surface: ``Onsager((g0, g1), temperature=1, reduced=True, charges=+-1).run()`` on the float32
    frames an MDAnalysis memory reader would hold (12 GB of host memory, uploaded once, prepared on the
    device), then ``calculate_transport_coefficients`` / conductivity / transference numbers
    (reference transport.py:59-286, 912-1059).  Checked: (a) all three cross MSDs element-wise against the
    oracle's msd_fft of the summed trajectories (summed on the host); (b) msd_self against the direct
    definition over ALL particles at small, middle and large lags; (c) a 64-particle analysis element-wise
    against the oracle; (d) D_i against the free walk's sigma^2 / 2 dt, L_ii_self = N_i D_i / (kBT V), the
    symmetry of L_ij, L_ii against L_ii_self and L_01 against 0 within the statistics of ONE summed
    trajectory, conductivity and transference numbers from L_ij."""
    import time
    import mdhelper_amd
    from mdhelper_amd.analysis import Onsager
    from oracle import correlation as oc
    T, n, sigma, box = 100_000, 10_000, 0.1, 50.0
    d = _core.synth_random_walk(T, n, [1.0, 1.0, 1.0], sigma, seed=4, wrap=False)     # float32, unwrapped
    h = d.to_host()
    d.free()
    tb = T // n_blocks
    u = mdhelper_amd.ArrayUniverse(h, [box, box, box, 90, 90, 90], dt=1.0,
                                   charges=np.r_[np.ones(n // 2), -np.ones(n - n // 2)])
    assert u.trajectory._positions is h or np.shares_memory(u.trajectory._positions, h)
    groups = (u.atoms[:n // 2], u.atoms[n // 2:])
    t0 = time.perf_counter()
    ons = Onsager(groups, temperature=1, reduced=True, n_blocks=n_blocks, verbose=False).run()
    wall = time.perf_counter() - t0
    print(f"Onsager.run() on 12 GB of host float32, n_blocks={n_blocks}: {wall:.2f} s")
    assert ons._from_file
    res = ons.results
    assert res.pairs == ((0, 0), (0, 1), (1, 1))
    assert res.msd_self.shape == (2, n_blocks, tb) and res.msd_cross.shape == (3, n_blocks, tb)
    assert np.array_equal(res.times, np.arange(tb, dtype=float))

    # (a) cross MSDs: the oracle on the summed trajectories
    halves = [(0, n // 2), (n // 2, n)]
    sums = [np.concatenate([h[a:a + 2000, lo:hi].sum(axis=1, dtype=np.float64) for a in range(0, T, 2000)])
            .reshape(n_blocks, tb, 3) for lo, hi in halves]
    ref = {(0, 0): oc.msd_fft_ref(sums[0], axis=1) / 6, (1, 1): oc.msd_fft_ref(sums[1], axis=1) / 6,
           (0, 1): oc.msd_fft_ref(sums[0], sums[1], axis=1) / 6}
    scale = np.sqrt(np.abs(ref[(0, 0)] * ref[(1, 1)]))          # per lag: what a cross term is measured against
    for i, pair in enumerate(res.pairs):
        err = np.abs(res.msd_cross[i] - ref[pair])
        assert np.all(err[:, 1:] <= 1e-6 * np.maximum(np.abs(ref[pair]), scale)[:, 1:]), pair
        assert np.all(err[:, 0] <= 1e-6 * scale[:, 1]), pair     # lag 0: pure round-off

    # (b) self MSDs: the direct definition, every particle
    for b in sorted({0, n_blocks - 1}):
        for lag in (1, 1000, tb // 2, tb - 1):
            direct = _direct_msd_sums(h, halves, b * tb, tb, lag) / (n // 2) / 6
            assert np.allclose(res.msd_self[:, b, lag], direct, rtol=1e-6), (b, lag)
    assert np.all(np.abs(res.msd_self[:, :, 0]) < 1e-6 * res.msd_self[:, :, 1])

    # (c) 64 particles, every lag, against the oracle's per-particle msd_fft
    small = Onsager(u.atoms[:64], temperature=1, reduced=True, n_blocks=n_blocks, verbose=False).run()
    p64 = h[:, :64].astype(np.float64).reshape(n_blocks, tb, 64, 3)
    want = oc.msd_fft_ref(p64, axis=1, average=False).mean(axis=-1) / 6
    assert np.allclose(small.results.msd_self[0][:, 1:], want[:, 1:], rtol=1e-6)
    assert np.allclose(small.results.msd_cross[0], oc.msd_fft_ref(p64.sum(axis=2), axis=1) / 6, rtol=1e-6,
                       atol=1e-6 * want[:, 1].max() * 64)

    # (d) transport coefficients of the free walk: MSD_self / 6 = sigma^2 m / 2  =>  D = sigma^2 / (2 dt)
    D_true, kBT, V = sigma ** 2 / 2, 1.0, box ** 3
    ons.calculate_transport_coefficients(start=1, stop=tb // 10, scale="linear")
    assert res.L_ij.shape == (n_blocks, 2, 2) and res.D_i.shape == (n_blocks, 2)
    assert np.allclose(res.D_i, D_true, rtol=0.02), res.D_i
    assert np.allclose(res.L_ii_self, (n // 2) * res.D_i / (kBT * V), rtol=1e-13)
    assert np.array_equal(res.L_ij, res.L_ij.transpose(0, 2, 1))
    L_self = (n // 2) * D_true / (kBT * V)
    # one summed trajectory per (group, block): the collective slope scatters by tens of per cent
    diag = np.stack((res.L_ij[:, 0, 0], res.L_ij[:, 1, 1]), axis=1)
    assert np.all(np.abs(diag / L_self - 1) < 0.6), diag / L_self
    assert abs(diag.mean() / L_self - 1) < (0.45 if n_blocks == 1 else 0.25), diag.mean() / L_self
    assert np.all(np.abs(res.L_ij[:, 0, 1]) < 0.6 * L_self), res.L_ij[:, 0, 1] / L_self
    ons.calculate_conductivity()
    ons.calculate_transference_number()
    ons.calculate_electrophoretic_mobility()
    kappa = res.L_ij[:, 0, 0] + res.L_ij[:, 1, 1] - 2 * res.L_ij[:, 0, 1]             # z = (+1, -1)
    assert np.allclose(res.conductivities, kappa, rtol=1e-12)
    t_plus = (res.L_ij[:, 0, 0] - res.L_ij[:, 0, 1]) / kappa
    assert np.allclose(res.transference_numbers, np.stack((t_plus, 1 - t_plus), axis=1), rtol=1e-10, atol=1e-12)
    rho = (n // 2) / V
    assert np.allclose(res.electrophoretic_mobilities[:, 0], (res.L_ij[:, 0, 0] - res.L_ij[:, 0, 1]) / rho,
                       rtol=1e-12)
    # the reference's default fit (log scale, slope fixed to 1, every lag from 1 on: dominated by the
    # long, poorly sampled lags) still lands on the walk's diffusivity
    ons.calculate_transport_coefficients()
    assert np.allclose(res.D_i, D_true, rtol=0.10), res.D_i


def test_c5_size_rows_against_all_atoms_equal_the_c_oracle():
    """C5 size (131 072 atoms, L = 109.4): 2 048 atoms against all atoms of one frame, two-group call,
    bit-identical with the C oracle (2.7e8 ordered pairs); default path and explicit cell."""
    from oracle.cbind import c_radial_histogram
    n = 131072
    Lc = np.float32(109.4)
    dims = np.array([Lc, Lc, Lc, 90, 90, 90], dtype=np.float32)
    d = _core.synth_random_walk(1, n, [Lc, Lc, Lc], 0.3, seed=5)
    frame = d.to_host()[0]
    d.free()
    edges = np.linspace(0.0, 15.0, 202)
    for lo in (0, 70_000):
        rows = np.ascontiguousarray(frame[lo:lo + 2048])
        want = c_radial_histogram(rows, frame, 201, (0.0, 15.0), dims, exclusion=None, n_threads=16)
        for algo in ("auto", "cell"):
            eng = _core.RdfEngine(edges, None, algo=algo)
            eng.accumulate(rows[None], frame[None], dims)
            got = eng.counts()
            eng.close()
            assert np.array_equal(got, want), (lo, algo)
        assert want[0] >= 2048          # the 2 048 self pairs (d = 0) sit in bin 0
