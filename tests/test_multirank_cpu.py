"""
World-size-2 tests of the sharded drivers on CPU (gloo): frames (RDF, S(q)) and
particles (Onsager) shard across ranks and the accumulators meet in one
all-reduce.  There is no GPU here, so the device engines are replaced — in this
test only — by oracle-backed stand-ins with the same interface; what is under
test is the host logic: the shard each rank takes, the reduction, and the
bit-exact / 1e-6 agreement of the reduced result with a single-rank run.
"""
import os
import subprocess
import sys

import numpy as np

# NOTE: nothing here imports torch.  `pytest -m gpu` imports every test module at collection, and a torch that is
# loaded before libmdx.so binds the library to the ROCm copy the wheel bundles: the gloo ranks
# are fresh `sys.executable` children (tests/helpers/gloo_rank.py), like the launcher's ranks in test_launch_cpu.

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleRdfEngine:
    def __init__(self, edges, exclusion=None, **kw):
        self.edges, self.exclusion = np.asarray(edges), exclusion
        self.n_bins = len(edges) - 1
        self._counts = np.zeros(self.n_bins, dtype=np.int64)
        self._grouping = {}
        self._drop = None

    def set_drop_axis(self, axis):
        self._drop = axis

    def set_grouping(self, which, offsets, masses):
        self._grouping[which] = None if offsets is None else (np.asarray(offsets), np.asarray(masses))

    def _points(self, which, pos):
        # centres of mass as the host path forms them (algorithm/molecule.py)
        g = self._grouping.get(which)
        if g is None:
            return pos
        offsets, m = g
        inverse = np.repeat(np.arange(len(offsets) - 1), np.diff(offsets))
        msum = np.bincount(inverse, weights=m)
        p = np.asarray(pos, dtype=np.float64)
        return np.stack([np.bincount(inverse, weights=m * p[:, k]) for k in range(3)], axis=1) / msum[:, None]

    def accumulate(self, pos1, pos2=None, boxes=None):
        from oracle.cbind import c_radial_histogram
        pos1 = np.asarray(pos1)
        a = [self._points(1, pos1[f]) for f in range(pos1.shape[0])]
        b = a if pos2 is None else [self._points(2, np.asarray(pos2)[f]) for f in range(pos1.shape[0])]
        pos1, pos2 = a, (None if pos2 is None else b)
        for f in range(len(pos1)):
            p1 = np.array(pos1[f], dtype=np.float32)
            p2 = p1 if pos2 is None else np.array(pos2[f], dtype=np.float32)
            box = None if boxes is None else np.array(np.asarray(boxes)[f], dtype=np.float32)
            if self._drop is not None:        # structure.py:761-766
                p1[:, self._drop] = 0
                p2[:, self._drop] = 0
                box[self._drop] = box[:3].max()
            c_radial_histogram(p1, p2, self.n_bins, (self.edges[0], self.edges[-1]), box,
                               exclusion=self.exclusion, counts=self._counts, n_threads=1)

    def counts(self):
        return self._counts.copy()

    def close(self):
        pass


def _centres(grouping, frames):
    """float32 centres of mass of rows sorted molecule by molecule (algorithm/molecule.py)."""
    if grouping is None:
        return np.asarray(frames, dtype=np.float32)
    offsets, m = grouping
    inverse = np.repeat(np.arange(len(offsets) - 1), np.diff(offsets))
    msum = np.bincount(inverse, weights=m)
    out = []
    for p in np.asarray(frames, dtype=np.float64):
        out.append(np.stack([np.bincount(inverse, weights=m * p[:, k]) for k in range(3)], axis=1) / msum[:, None])
    return np.asarray(out).astype(np.float32)


class OracleSqEngine:
    def __init__(self, wavevectors, group_sizes, pairs, **kw):
        self.q, self.sizes, self.pairs = np.asarray(wavevectors), list(group_sizes), pairs
        self._acc = np.zeros((len(pairs), len(self.q)))
        self._grouping = None

    def set_grouping(self, offsets, masses):
        self._grouping = None if offsets is None else (np.asarray(offsets), np.asarray(masses))

    def accumulate(self, pos):
        from oracle import fourier as of
        pos = _centres(self._grouping, pos)
        slices, idx = [], 0
        for n in self.sizes:
            slices.append(slice(idx, idx + n))
            idx += n
        mode = None if self.pairs[0][0] is None else "partial"
        for f in range(len(pos)):
            self._acc += of.ssf_frame_ref(self.q, np.asarray(pos[f], dtype=np.float64), slices, self.pairs, mode)

    def result(self):
        return self._acc.copy()

    def close(self):
        pass


class OracleIsfEngine:
    """Un-normalised lag sums as the device engine returns them, from the restated driver."""

    def __init__(self, wavevectors, group_sizes, pairs, n_lags, incoherent=False, **kw):
        self.q, self.sizes, self.pairs = np.asarray(wavevectors), list(group_sizes), pairs
        self.n_lags, self.incoherent = n_lags, incoherent
        self._frames = []
        self._grouping = None

    def set_grouping(self, offsets, masses):
        self._grouping = None if offsets is None else (np.asarray(offsets), np.asarray(masses))

    def accumulate(self, pos):
        self._frames.extend(_centres(self._grouping, pos))

    def result(self):
        from oracle import fourier as of
        frames = np.stack(self._frames)
        mode = None if self.pairs[0][0] is None else ("pair" if len(self.pairs) == 1 else "partial")
        ref = of.isf_run_ref(frames, self.sizes, self.q, self.n_lags, mode=mode,
                             incoherent=self.incoherent, sort=False, unique=False)
        norm = sum(self.sizes) * np.arange(len(frames), len(frames) - self.n_lags, -1)[:, None, None]
        return ref["cisf"] * norm, (ref["iisf"] * norm if self.incoherent else None)

    def close(self):
        pass


class OracleMsdEngine:
    def __init__(self, t_block, n_blocks, n_groups, **kw):
        self.tb, self.b, self.g = t_block, n_blocks, n_groups
        self._msd = np.zeros((n_groups, n_blocks, t_block))
        self._traj = np.zeros((n_groups, n_blocks, t_block, 3))
        self._acf = np.zeros((n_groups, n_blocks, t_block))

    def push(self, group, positions, first, count, zero_dims=0):
        from oracle import correlation as oc
        p = np.asarray(positions)[:self.tb * self.b, first:first + count].reshape(self.b, self.tb, count, 3).copy()
        for k in range(3):
            if (zero_dims >> k) & 1:
                p[..., k] = 0
        self._msd[group] += oc.msd_fft_ref(p, axis=1, average=False).sum(axis=-1)
        self._traj[group] += p.sum(axis=2)
        acf = oc.correlation_fft_ref(p, axis=1, vector=True).sum(axis=-1)       # [B, T_b], per lag mean
        self._acf[group] += acf * (self.tb - np.arange(self.tb))

    # the device-side frame preparation, as far as these CPU runs reach it (plain atom groups)
    has_grouping = False

    def set_grouping(self, offsets, masses):
        assert offsets is None

    def set_initial_images(self, images):
        pass

    def push_f32(self, group, positions, *, unwrap_dims=None, zero_dims=0, shift=None):
        assert unwrap_dims is None and shift is None
        p = np.asarray(positions, dtype=np.float64)
        self.push(group, p, 0, p.shape[1], zero_dims)

    def result(self, want_msd=True):
        return self._msd.copy(), self._traj.copy()

    def result_acf(self):
        return self._acf.copy()

    def close(self):
        pass


def _install_stand_ins():
    from mdhelper_amd import _core
    from mdhelper_amd.algorithm import correlation
    from oracle import correlation as oc
    _core.RdfEngine, _core.SqEngine, _core.MsdEngine = OracleRdfEngine, OracleSqEngine, OracleMsdEngine
    _core.IsfEngine = OracleIsfEngine
    correlation.msd_fft = oc.msd_fft_ref
    from mdhelper_amd.analysis import Onsager
    Onsager._hbm_share = 0.0          # no device to keep frames on: the group-by-group route


def _build_inputs():
    rng = np.random.default_rng(21)
    L = np.float32(14.0)
    frames = (rng.random((9, 300, 3)) * L).astype(np.float32)
    walk = 7.0 + np.cumsum(rng.normal(scale=0.2, size=(60, 25, 3)), axis=0)
    return frames, L, walk


def _analyses(comm):
    import mdhelper_amd
    from mdhelper_amd.analysis import (IntermediateScatteringFunction, Onsager,
                                       RadialDistributionFunction, StructureFactor)
    from mdhelper_amd.analysis.polymer import EndToEndVector
    frames, L, walk = _build_inputs()
    u = mdhelper_amd.ArrayUniverse(frames, [L, L, L, 90, 90, 90])
    rdf = RadialDistributionFunction(u.atoms, n_bins=40, range=(0.0, 6.0), exclusion=(1, 1), comm=comm).run()
    slow = RadialDistributionFunction(u.atoms[:100], u.atoms[100:], n_bins=40, range=(0.0, 6.0),
                                      groupings="residues", drop_axis="y", comm=comm).run()
    com = RadialDistributionFunction(u.atoms[:100], u.atoms[100:], n_bins=40, range=(0.0, 6.0),
                                     groupings="residues", comm=comm).run()
    sf = StructureFactor((u.atoms[:120], u.atoms[120:]), mode="partial", n_points=3, comm=comm).run()
    isf = IntermediateScatteringFunction((u.atoms[:120], u.atoms[120:]), mode="partial", n_points=3,
                                         n_lags=4, incoherent=True, comm=comm).run()
    ur = mdhelper_amd.ArrayUniverse(frames, [L, L, L, 90, 90, 90], resids=np.arange(300) // 3,
                                    masses=np.linspace(1.0, 9.0, 300))
    sfr = StructureFactor((ur.atoms[:120], ur.atoms[120:]), ("residues", "atoms"), mode="partial", n_points=3,
                          comm=comm).run()
    isfr = IntermediateScatteringFunction((ur.atoms[:120], ur.atoms[120:]), "residues", mode="partial",
                                          n_points=3, n_lags=3, incoherent=True, comm=comm).run()
    uw = mdhelper_amd.ArrayUniverse(walk, [14.0, 14.0, 14.0, 90, 90, 90])
    ons = Onsager((uw.atoms[:15], uw.atoms[15:]), temperature=1.0, reduced=True, n_blocks=2, comm=comm).run()
    # 25 particles = 5 chains of 5 (group 1) ... the first 20 as 4 chains of 5, the rest as 1 chain
    e2e = EndToEndVector((uw.atoms[:20], uw.atoms[20:]), n_chains=(4, 1), n_monomers=(5, 5), n_blocks=2,
                         comm=comm).run()
    return {"ssf_res": sfr.results.ssf, "cisf_res": isfr.results.cisf, "iisf_res": isfr.results.iisf,
            "acf": e2e.results.acf, "counts": rdf.results.counts, "rdf": rdf.results.rdf, "counts_slow": slow.results.counts,
            "counts_com": com.results.counts, "cisf": isf.results.cisf, "iisf": isf.results.iisf,
            "ssf": sf.results.ssf, "msd_self": ons.results.msd_self, "msd_cross": ons.results.msd_cross}


def _child(*args, env=None):
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "helpers", "gloo_rank.py"), *map(str, args)],
                            env={**os.environ, **(env or {})}, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                            text=True)


def test_world_size_2_matches_single_rank(tmp_path):
    # stand-ins are installed in child processes only, never in the pytest process
    port = 29500 + os.getpid() % 2000
    procs = [_child("rank", rank, 2, port, tmp_path) for rank in range(2)] + [_child("single", tmp_path)]
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0, out
    single = np.load(tmp_path / "single.npz")
    for rank in range(2):
        got = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(got["counts"], single["counts"])            # integer sums: bit-exact
        assert np.array_equal(got["counts_slow"], single["counts_slow"])
        assert np.array_equal(got["counts_com"], single["counts_com"])
        assert np.allclose(got["rdf"], single["rdf"], rtol=1e-12)
        assert np.allclose(got["ssf"], single["ssf"], rtol=1e-9, atol=1e-12)
        assert np.allclose(got["cisf"], single["cisf"], rtol=1e-9, atol=1e-12)     # wavevectors shard
        assert np.allclose(got["iisf"], single["iisf"], rtol=1e-9, atol=1e-12)
        assert np.allclose(got["msd_self"], single["msd_self"], rtol=1e-9, atol=1e-12)
        assert np.allclose(got["msd_cross"], single["msd_cross"], rtol=1e-9, atol=1e-10)
        assert np.allclose(got["acf"], single["acf"], rtol=1e-9, atol=1e-12)       # chains shard
        for name in ("ssf_res", "cisf_res", "iisf_res"):                            # device-COM wiring
            assert np.allclose(got[name], single[name], rtol=1e-9, atol=1e-12), name
    assert single["counts"].sum() > 0 and single["counts_com"].sum() > 0
