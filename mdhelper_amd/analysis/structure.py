"""
Bulk structural analysis on the GPU (operator surface of
``mdhelper.analysis.structure``).

Mirrors reference ``src/mdhelper/analysis/structure.py``:

* ``radial_histogram``               :32-104
* ``zeroth_order_hankel_transform``  :106-146, ``radial_fourier_transform`` :148-188
* ``calculate_coordination_numbers`` :190-285, ``calculate_structure_factor`` :287-442
* ``RadialDistributionFunction``     :444-1032
* ``StructureFactor``                :1034-1550

Names, argument order, defaults, result attributes and error behaviour follow
the reference.  The per-frame arithmetic (pair search + histogram; fused
``exp(i q.r)`` sums) runs in ``libmdx.so``; there is no NumPy fallback for it.
``results.units`` holds unit *names* (strings) because ``pint`` is not a
dependency here.
"""

from __future__ import annotations

import warnings
from itertools import combinations_with_replacement
from typing import Union

import numpy as np
from scipy.integrate import simpson
from scipy.signal import argrelextrema
from scipy.special import jv

from .. import _core
from ..algorithm.molecule import center_of_mass
from ..algorithm.unit import strip_unit
from ..algorithm.utility import get_closest_factors
from ..comm import shard_range
from ..universe import box_volumes
from .base import DynamicAnalysisBase, FrameBatcher, NumbaAnalysisBase

_GROUPINGS_RDF = {"atoms", "residues", "segments"}
_KB_KJ_PER_MOL_K = 8.31446261815324e-3   # N_A * k_B in kJ/(mol K)


def radial_histogram(pos1, pos2, n_bins: int, range, dims, *, exclusion=None) -> np.ndarray:
    r"""
    Radial histogram of minimum-image distances between two position sets
    (reference structure.py:32-104), computed on the GPU.

    Parameters
    ----------
    pos1, pos2 : array-like, shape (N_1, 3) / (N_2, 3) (a single (3,) point is accepted)
    n_bins : int
    range : (float, float)
    dims : array-like, shape (6,)
        Box lengths and angles (orthorhombic or triclinic).
    exclusion : (int, int), keyword-only, optional
        Pairs with ``i // exclusion[0] == j // exclusion[1]`` are dropped;
        ``(1, 1)`` drops self pairs.

    Returns
    -------
    histogram : numpy.ndarray of int64, shape (n_bins,)
    """
    edges = np.linspace(range[0], range[1], n_bins + 1)
    return _core.radial_histogram_device(pos1, pos2, n_bins, edges, dims, exclusion)


def zeroth_order_hankel_transform(r, f, q) -> np.ndarray:
    r""":math:`F_0(q)=2\pi\int f(r)J_0(qr)r\,dr` by Simpson's rule (reference :106-146)."""
    r, f, q = np.asarray(r), np.asarray(f), np.asarray(q)
    ht = 2 * np.pi * simpson(f * r * jv(0, np.outer(q, r)), x=r)
    if 0 in q:
        ht[q == 0] = 2 * np.pi * simpson(f * r, x=r)
    return ht


def radial_fourier_transform(r, f, q) -> np.ndarray:
    r""":math:`\hat f(q)=\frac{4\pi}{q}\int f(r)r\sin(qr)\,dr` by Simpson's rule (reference :148-188)."""
    r, f, q = np.asarray(r), np.asarray(f), np.asarray(q)
    with np.errstate(divide="ignore", invalid="ignore"):
        rft = 4 * np.pi * np.divide(simpson(f * r * np.sin(np.outer(q, r)), x=r), q)
    if 0 in q:
        rft[q == 0] = 4 * np.pi * simpson(f * r ** 2, x=r)
    return rft


def calculate_coordination_numbers(bins, rdf, rho: float, *, n_coord_nums: int = 2,
                                   n_dims: int = 3, threshold: float = 0.1) -> np.ndarray:
    r"""
    Coordination numbers :math:`n_k` between successive local minima of
    :math:`g(r)` (reference :190-285); ``numpy.nan`` where no minimum exists.
    """
    if n_dims not in {2, 3}:
        raise ValueError("Invalid number of dimensions.")
    bins, rdf = np.asarray(bins), np.asarray(rdf)

    def shell(lo, hi):
        r = bins[lo:hi]
        g = rdf[lo:hi]
        if n_dims == 3:
            return 4 * np.pi * rho * simpson(r ** 2 * g, x=r)
        return 2 * np.pi * rho * simpson(r * g, x=r)

    coord_nums = np.full(n_coord_nums, np.nan)
    i_min, = argrelextrema(rdf, np.less)
    i_min = i_min[rdf[i_min] >= threshold]
    if len(i_min):
        coord_nums[0] = shell(0, i_min[0] + 1)
        for i in np.arange(min(n_coord_nums, len(i_min)) - 1):
            coord_nums[i + 1] = shell(i_min[i], i_min[i + 1] + 1)
    else:
        warnings.warn("No local minima found.")
    return coord_nums


def calculate_structure_factor(r, g, equal: bool, rho: float, x_i: float = 1, x_j: float = None,
                               q=None, *, q_lower: float = None, q_upper: float = None,
                               n_q: int = 1_000, n_dims: int = 3, formalism: str = "FZ"):
    r"""(Partial) static structure factor from :math:`g_{ij}(r)` (reference :287-442)."""
    r, g = np.asarray(r), np.asarray(g)
    if q is None:
        if q_lower is None:
            q_lower = 2 * np.pi / r[-1]
        if q_upper is None:
            q_upper = 2 * np.pi / r[0]
        q = np.linspace(q_lower, q_upper,
                        int((q_upper - q_lower) / q_lower) if n_q is None else n_q)
    if n_dims == 3:
        transform = radial_fourier_transform
    elif n_dims == 2:
        transform = zeroth_order_hankel_transform
    else:
        raise ValueError("Invalid number of dimensions.")
    rho_sft = rho * transform(r, g - 1, q)
    if equal or formalism == "FZ":
        return q, 1 + rho_sft
    if formalism == "AL":
        return q, (x_i == x_j) + np.sqrt(x_i * x_j) * rho_sft
    if formalism == "general":
        return q, 1 + x_i * x_j * rho_sft
    raise ValueError("Invalid formalism.")


def _group_positions(group, grouping):
    return group.positions if grouping == "atoms" else center_of_mass(group, grouping)


def _is_array_trajectory(traj) -> bool:
    return hasattr(traj, "frame_block") and hasattr(traj, "box_block")


def _device_frames(traj, frames):
    """The listed frames as a float32 device array without a copy (a ``universe.DeviceTrajectory`` over float32
    frames, consecutive frames), else None: the caller then goes through host memory."""
    block = getattr(traj, "device_block", None)
    if block is None or traj.device_array.dtype != np.float32:
        return None
    return block(frames)


class RadialDistributionFunction(DynamicAnalysisBase):
    r"""
    Radial distribution function :math:`g_{ij}(r)` and related properties
    (reference structure.py:444-1032), with the pair histogram on the GPU.

    Parameters (as in the reference)
    --------------------------------
    ag1, ag2 : AtomGroup (``ag2=None`` → ``ag1``)
    n_bins : int, default 201
    range : (float, float), default (0.0, 15.0)
    drop_axis : {0, 1, 2, "x", "y", "z"}, keyword-only, optional
    norm : {"rdf", "density", None}, keyword-only
    exclusion : (int, int), keyword-only, optional
    groupings : str or (str, str), keyword-only
    reduced : bool, keyword-only
    n_batches : int, keyword-only — accepted, ignored: the GPU kernel never
        materialises a pair list, so there is nothing to batch (and none of the
        reference's "counts may be off by a few" caveat, structure.py:601-607)
    parallel : bool, keyword-only — accepted; the GPU is the parallelism
    verbose : bool, keyword-only
    algo : {"auto", "exact", "filter", "cell"}, keyword-only (extension)
    comm : communicator, keyword-only (extension) — frames shard across ranks,
        counts and volume meet in one all-reduce in ``_conclude``

    Results: ``results.edges``, ``results.bins``, ``results.counts`` (int64),
    ``results.rdf``, ``results.units``; after the ``calculate_*`` calls
    ``results.coordination_numbers``, ``results.pmf``, ``results.wavenumbers``,
    ``results.ssf``.
    """

    def __init__(self, ag1, ag2=None, n_bins: int = 201, range: tuple = (0.0, 15.0), *,
                 drop_axis: Union[int, str] = None, norm: str = "rdf", exclusion: tuple = None,
                 groupings: Union[str, tuple] = "atoms", reduced: bool = False,
                 n_batches: int = None, parallel: bool = False, verbose: bool = True,
                 algo: str = "auto", **kwargs) -> None:
        self.ag1 = ag1
        self.ag2 = ag1 if ag2 is None else ag2
        self.universe = self.ag1.universe
        if self.universe.dimensions is None:
            raise ValueError("Trajectory does not contain system dimension information.")

        super().__init__(self.universe.trajectory, parallel, verbose, **kwargs)

        if isinstance(groupings, str):
            if groupings not in _GROUPINGS_RDF:
                raise ValueError(f"Invalid grouping '{groupings}'. The options are "
                                 "'atoms', 'residues', and 'segments'.")
            self._groupings = 2 * [groupings]
        else:
            for g in groupings:
                if g not in _GROUPINGS_RDF:
                    raise ValueError(f"Invalid grouping '{g}'. The options are "
                                     "'atoms', 'residues', and 'segments'.")
            self._groupings = list(2 * tuple(groupings) if len(groupings) == 1 else groupings)

        self._drop_axis = ord(drop_axis) - 120 if isinstance(drop_axis, str) else drop_axis
        if self._drop_axis not in {0, 1, 2, None}:
            raise ValueError("Invalid axis to drop.")
        if norm not in {"rdf", "density", None}:
            raise ValueError("Invalid normalization.")

        self._n_bins = n_bins
        self._range = range
        self._norm = norm
        self._exclusion = exclusion
        self._reduced = reduced
        self._n_batches = n_batches
        self._verbose = verbose
        self._algo = algo

    # ------------------------------------------------------------------ protocol

    def _prepare(self) -> None:
        self.results.edges = np.linspace(*self._range, self._n_bins + 1)
        self.results.bins = (self.results.edges[:-1] + self.results.edges[1:]) / 2
        self.results.counts = np.zeros(self._n_bins, dtype=int)
        self.results.units = {"results.bins": "angstrom", "results.edges": "angstrom"}
        self._area_or_volume = 0.0
        self._same = (self.ag1 is self.ag2 or self.ag1 == self.ag2) \
            and self._groupings[0] == self._groupings[1]
        self._engine = _core.RdfEngine(self.results.edges, self._exclusion, algo=self._algo,
                                       dev=self._device)
        n1 = getattr(self.ag1, f"n_{self._groupings[0]}")
        n2 = getattr(self.ag2, f"n_{self._groupings[1]}")
        self._batch = FrameBatcher([n1] if self._same else [n1, n2], self._flush)
        self._frames_mine = shard_range(self.n_frames, self._comm.rank, self._comm.world_size)

    def _flush(self, blocks, boxes):
        self._engine.accumulate(blocks[0], None if self._same else blocks[1], boxes)

    def _frame_volume(self, dims):
        if self._drop_axis is None:
            return float(self._ts.volume)
        return float(np.delete(dims[:3], self._drop_axis).prod())

    def _single_frame(self) -> None:
        lo, hi = self._frames_mine
        if not lo <= self._frame_index < hi:
            return
        dims = np.array(self._ts.dimensions, dtype=np.float32)
        pos1 = _group_positions(self.ag1, self._groupings[0])
        pos2 = None if self._same else _group_positions(self.ag2, self._groupings[1])
        if self._drop_axis is not None:
            # avoid periodic images along the dropped dimension (reference :761-770)
            pos1 = np.array(pos1, dtype=np.float32)
            pos1[:, self._drop_axis] = 0
            if pos2 is not None:
                pos2 = np.array(pos2, dtype=np.float32)
                pos2[:, self._drop_axis] = 0
            dims[self._drop_axis] = dims[:3].max()
        # always tracked (the reference only does for norm="rdf", which leaves _get_rdf()
        # without a volume for the other norms)
        self._area_or_volume += self._frame_volume(dims)
        self._batch.add([pos1] if self._same else [pos1, pos2], dims)

    def _conclude(self):
        self._batch.flush()
        if self._comm.world_size > 1 and getattr(self._comm, "device_collectives", False):
            self._engine.allreduce(self._comm)
            counts = self._engine.counts()
        else:
            counts = self._comm.allreduce(self._engine.counts())
        self.results.counts[:] = counts
        if self._comm.world_size > 1:
            self._area_or_volume = float(
                self._comm.allreduce(np.array([self._area_or_volume], dtype=np.float64))[0])
        self._engine.close()

        # normalisation, reference :846-862
        norm = self.n_frames
        if self._norm is not None:
            if self._drop_axis is None:
                norm = norm * (4 * np.pi * np.diff(self.results.edges ** 3) / 3)
            else:
                norm = norm * (np.pi * np.diff(self.results.edges ** 2))
            if self._norm == "rdf":
                _N2 = getattr(self.ag2, f"n_{self._groupings[1]}")
                if self._exclusion:
                    _N2 -= self._exclusion[1]
                norm = norm * (getattr(self.ag1, f"n_{self._groupings[0]}") * _N2
                               * self.n_frames / self._area_or_volume)
        self.results.rdf = self.results.counts / norm

    # batched fast path for in-memory and file trajectories: frames go to the engine in
    # contiguous blocks instead of one Python iteration per frame; residue / segment centres
    # of mass are formed on the device (mdx_rdf_set_grouping)
    @staticmethod
    def _selection(ag, grouping):
        """(particle indices, CSR offsets or None, masses or None) of one side of the histogram."""
        idx = np.asarray(ag.indices)
        if grouping == "atoms":
            return idx, None, None
        from ..algorithm.molecule import _level_ids
        _, inverse = np.unique(_level_ids(ag, grouping), return_inverse=True)
        order = np.argsort(inverse, kind="stable")        # molecule by molecule, atom order kept
        offsets = np.concatenate(([0], np.cumsum(np.bincount(inverse))))
        return idx[order], offsets, np.asarray(ag.masses, dtype=np.float64)[order]

    def run(self, start=None, stop=None, step=None, frames=None, verbose=None, **kwargs):
        traj = self._trajectory
        fast = _is_array_trajectory(traj)
        if not fast:
            return super().run(start=start, stop=stop, step=step, frames=frames, verbose=verbose,
                               **kwargs)
        self._setup_frames(traj, start=start, stop=stop, step=step, frames=frames)
        self._prepare()
        numbers = self._frame_numbers()
        self.frames[:] = numbers
        self.times[:] = numbers * traj.dt
        lo, hi = self._frames_mine
        mine = numbers[lo:hi]
        block = self._batch.capacity
        i1, off1, m1 = self._selection(self.ag1, self._groupings[0])
        i2, off2, m2 = (i1, off1, m1) if self._same else self._selection(self.ag2, self._groupings[1])
        if off1 is not None:
            self._engine.set_grouping(1, off1, m1)
        if off2 is not None and not self._same:
            self._engine.set_grouping(2, off2, m2)
        # 2-D mode: the coordinate is zeroed and the cell stretched on the device (:761-766)
        self._engine.set_drop_axis(self._drop_axis)
        all1 = len(i1) == traj.n_atoms and np.array_equal(i1, np.arange(len(i1)))
        native = getattr(traj, "native", None)
        contiguous = len(mine) < 2 or bool(np.all(np.diff(mine) == 1))
        if native is not None:
            # trajectory file: raw frames stream file -> pinned memory -> HBM inside the library
            block = max(block, 4096, (4 << 30) // max(12 * traj.n_atoms, 1))
        else:
            # in-memory frames are handed over where they lie (slices, no copy): ~4 GiB per call (1 GiB when
            # the selection has to be gathered first); the library pipelines inside a call (copy of slab
            # k + 1 beside the kernels of slab k), every call boundary drains that pipeline once
            block = max(block, ((4 if contiguous else 1) << 30) // max(12 * traj.n_atoms, 1))
        for b0 in np.arange(0, len(mine), block):
            sel = mine[b0:b0 + block]
            boxes = traj.box_block(sel)
            if self._drop_axis is None:
                self._area_or_volume += float(box_volumes(boxes).sum())
            else:
                self._area_or_volume += float(
                    np.delete(np.asarray(boxes, dtype=np.float32)[:, :3], self._drop_axis, axis=1)
                    .prod(axis=1, dtype=np.float32).astype(float).sum())
            if native is not None:
                self._engine.accumulate_traj(native, sel, boxes, None if all1 else i1,
                                             None if self._same else i2, same=self._same)
                continue
            resident = _device_frames(traj, sel) if (all1 and self._same and off1 is None) else None
            if resident is not None:
                # float32 frames already in HBM (ArrayUniverse.from_device), every particle, one group: the
                # kernels read them where they lie
                d_boxes = _core.DeviceArray.from_host(
                    np.ascontiguousarray(np.broadcast_to(np.asarray(boxes, dtype=np.float32), (len(sel), 6))),
                    self._device)
                try:
                    self._engine.accumulate_device(resident.ptr, traj.n_atoms, None, traj.n_atoms, d_boxes.ptr,
                                                   len(sel))
                    self._engine.synchronize()
                finally:
                    d_boxes.free()
                continue
            pos = traj.frame_block(sel)
            p1 = pos if all1 else pos[:, i1]
            p2 = None if self._same else pos[:, i2]
            self._engine.accumulate(p1, p2, boxes)
        self._conclude()
        return self

    # ------------------------------------------------------------ post-processing

    def _get_rdf(self) -> np.ndarray:
        """:math:`g_{ij}(r)` whatever ``norm`` was (reference :864-891)."""
        if self._norm == "rdf":
            return self.results.rdf
        _N2 = getattr(self.ag2, f"n_{self._groupings[1]}")
        if self._exclusion:
            _N2 -= self._exclusion[1]
        if self._drop_axis is None:
            norm = 4 * np.diff(self.results.edges ** 3) / 3
        else:
            norm = np.diff(self.results.edges ** 2)
        return self._area_or_volume * self.results.counts / (
            np.pi * self.n_frames ** 2 * _N2 * norm
            * getattr(self.ag1, f"n_{self._groupings[0]}"))

    def calculate_coordination_numbers(self, rho: float, *, n_coord_nums: int = 2,
                                       threshold: float = 0.1) -> None:
        self.results.coordination_numbers = calculate_coordination_numbers(
            self.results.bins, self._get_rdf(), rho, n_coord_nums=n_coord_nums,
            n_dims=2 + (self._drop_axis is None), threshold=threshold)

    def calculate_pmf(self, temperature) -> None:
        r"""Potential of mean force :math:`w(r)=-k_BT\ln g(r)` (reference :925-959)."""
        self.results.units["results.pmf"] = "kilojoule / mole"
        temperature, unit_ = strip_unit(temperature, "kelvin")
        if self._reduced:
            if not isinstance(unit_, str):
                raise ValueError("'temperature' cannot have units when reduced=True.")
            kBT = temperature
        else:
            kBT = _KB_KJ_PER_MOL_K * temperature
        with np.errstate(divide="ignore"):
            self.results.pmf = -kBT * np.log(self._get_rdf())

    def calculate_structure_factor(self, rho: float, x_i: float = None, x_j: float = None,
                                   q=None, *, q_lower: float = None, q_upper: float = None,
                                   n_q: int = 1_000, formalism: str = "FZ") -> None:
        self.results.wavenumbers, self.results.ssf = calculate_structure_factor(
            self.results.bins, self._get_rdf(), self.ag1 == self.ag2, rho, x_i, x_j, q=q,
            q_lower=q_lower, q_upper=q_upper, n_q=n_q, n_dims=2 + (self._drop_axis is None),
            formalism=formalism)


def _mean_over_equal_wavenumbers(x, wavenumbers, unique) -> np.ndarray:
    """
    ``x[..., N_q] -> [..., N_unique]``: column ``u`` is the mean of the columns whose wavenumber is
    ``numpy.isclose`` to ``unique[u]`` — the reference's ``_conclude`` (structure.py:1536-1541,
    2116-2127), which evaluates ``isclose`` once per unique wavenumber over all columns (minutes of
    host time at the default 32 768-wavevector grid).  Here every column finds its unique
    wavenumber(s) by bisection; where each column belongs to exactly one (any grid), the means are
    one segmented sum.  Otherwise the reference's loop runs as it stands.
    """
    w = np.asarray(wavenumbers, dtype=float)
    q = np.asarray(unique, dtype=float)
    order_q = np.argsort(q, kind="stable")
    qs = q[order_q]
    tol = 1e-8 + 1e-5 * np.abs(w)                       # isclose(q, w): |q - w| <= atol + rtol |w|
    lo = np.searchsorted(qs, w - tol, side="left")
    hi = np.searchsorted(qs, w + tol, side="right")
    # bisection brackets candidates; the comparison itself is isclose's
    single = (hi - lo == 1)
    if single.all() and np.all(np.abs(qs[lo] - w) <= tol):
        member = order_q[lo]                            # unique index of every column
        cols = np.argsort(member, kind="stable")
        counts = np.bincount(member, minlength=len(q))
        if np.all(counts > 0):
            starts = np.concatenate(([0], np.cumsum(counts)[:-1]))
            return np.add.reduceat(np.take(x, cols, axis=-1), starts, axis=-1) / counts
    return np.stack([x[..., np.isclose(v, w)].mean(axis=-1) for v in q], axis=-1)


class StructureFactor(NumbaAnalysisBase):
    r"""
    Static / partial structure factor from particle positions (reference
    structure.py:1034-1550):

    .. math:: S_{\alpha\beta}(q)=\frac{2-\delta_{\alpha\beta}}{N}\left\langle
              \mathrm{Re}\,\rho_\alpha(\mathbf q)\rho_\beta^*(\mathbf q)\right\rangle,\quad
              \rho_\alpha(\mathbf q)=\sum_{j\in\alpha}e^{i\mathbf q\cdot\mathbf r_j}

    Parameters as in the reference (``groups``, ``groupings``, ``mode``, ``form``,
    ``dimensions``, ``n_points``, ``n_surfaces``, ``n_surface_points``, ``q_max``,
    ``wavevectors``, ``sort``, ``unique``, ``parallel``, ``verbose``).  ``form``
    ("exp" / "trig") selects between two algebraically identical host
    formulations in the reference; both map onto the same fused GPU kernel.
    ``parallel`` is accepted and ignored.
    """

    @staticmethod
    def ssf_trigonometric_2d(qrs: np.ndarray) -> np.ndarray:
        r"""
        Static structure factors (un-normalised) from a caller-supplied array of
        :math:`\mathbf q\cdot\mathbf r_j`, shape :math:`(N_q, N_r)`, in the trigonometric form
        :math:`(\sum_j\cos)^2+(\sum_j\sin)^2` per row (reference structure.py:1238-1271).  The row sums run
        on the device (``mdx_trig_rowsums``); the analysis itself never materialises ``qrs``.
        """
        c, s = _core.trig_rowsums_device(qrs)
        return c * c + s * s

    @staticmethod
    def psf_trigonometric_2d_2d(qrs1: np.ndarray, qrs2: np.ndarray) -> np.ndarray:
        r"""
        Partial structure factors (un-normalised) from two arrays of :math:`\mathbf q\cdot\mathbf r`, shapes
        :math:`(N_q, N_\alpha)` and :math:`(N_q, N_\beta)`:
        :math:`2(\sum_j\cos\sum_k\cos+\sum_j\sin\sum_k\sin)` per row (reference structure.py:1273-1317).
        """
        if np.shape(qrs1)[0] != np.shape(qrs2)[0]:
            raise ValueError("The two arrays must have one row per wavevector each.")
        c1, s1 = _core.trig_rowsums_device(qrs1)
        c2, s2 = _core.trig_rowsums_device(qrs2)
        return 2 * (c1 * c2 + s1 * s2)

    def __init__(self, groups, groupings: Union[str, tuple] = "atoms", *, mode: str = None,
                 form: str = "exp", dimensions=None, n_points: int = 32, n_surfaces: int = None,
                 n_surface_points: int = 8, q_max=None, wavevectors=None, sort: bool = True,
                 unique: bool = True, parallel: bool = False, verbose: bool = True,
                 **kwargs) -> None:
        self._groups = [groups] if hasattr(groups, "universe") else list(groups)
        self.universe = self._groups[0].universe
        super().__init__(self.universe.trajectory, verbose, **kwargs)

        self._n_groups = len(self._groups)
        valid = {"atoms", "residues"}
        if isinstance(groupings, str):
            if groupings not in valid:
                raise ValueError(f"Invalid grouping '{groupings}'. Valid values: "
                                 f"{', '.join(sorted(valid))}.")
            self._groupings = self._n_groups * [groupings]
        else:
            if self._n_groups != len(groupings):
                raise ValueError("The number of grouping values is not equal to "
                                 "the number of groups.")
            for g in groupings:
                if g not in valid:
                    raise ValueError(f"Invalid grouping '{g}'. Valid values: "
                                     f"{', '.join(sorted(valid))}.")
            self._groupings = list(groupings)

        if mode not in {None, "pair", "partial"}:
            raise ValueError("Invalid mode.")
        if form not in {"exp", "trig"}:
            raise ValueError("Invalid form.")
        self._mode = mode
        if self._mode == "pair" and not 1 <= len(self._groups) <= 2:
            raise ValueError("There must be exactly one or two groups when mode='pair'.")
        elif self._mode is None:
            if sum(g.n_atoms for g in self._groups) != self.universe.atoms.n_atoms:
                raise ValueError("The provided atom groups do not contain all atoms "
                                 "in the universe.")

        if dimensions is not None:
            if len(dimensions) != 3:
                raise ValueError("'dimensions' must have length 3.")
            self._dimensions = np.asarray(strip_unit(dimensions, "angstrom")[0], dtype=float)
        elif self.universe.dimensions is not None:
            self._dimensions = np.array(self.universe.dimensions[:3], dtype=float)
        elif wavevectors is None:
            raise ValueError("No system dimensions found or provided.")

        # wavevectors (reference :1376-1410); numpy.meshgrid's default 'xy' indexing fixes the row order
        if wavevectors is not None:
            self._wavevectors = np.asarray(wavevectors, dtype=float)
        elif np.allclose(self._dimensions, self._dimensions[0]):
            grid = 2 * np.pi * np.arange(n_points) / self._dimensions[0]
            self._wavevectors = np.stack(np.meshgrid(grid, grid, grid), -1).reshape(-1, 3)
            if n_surfaces:
                n_theta, n_phi = get_closest_factors(n_surface_points, 2, reverse=True)
                theta = np.linspace(np.pi / (2 * n_theta + 4),
                                    np.pi / 2 - np.pi / (2 * n_theta + 4), n_theta)
                phi = np.linspace(np.pi / (2 * n_phi + 4),
                                  np.pi / 2 - np.pi / (2 * n_phi + 4), n_phi)
                directions = np.stack(
                    (np.sin(theta) * np.cos(phi)[:, None], np.sin(theta) * np.sin(phi)[:, None],
                     np.tile(np.cos(theta)[None, :], (n_phi, 1))), axis=-1)
                self._wavevectors = np.vstack((
                    self._wavevectors,
                    np.einsum("o,tpd->otpd", grid[1:n_surfaces + 1], directions)
                    .reshape((n_surfaces * n_surface_points, 3))))
        else:
            self._wavevectors = np.stack(
                np.meshgrid(*[2 * np.pi * np.arange(n_points) / L for L in self._dimensions]),
                axis=-1).reshape(-1, 3)
        self._wavenumbers = np.linalg.norm(self._wavevectors, axis=1)

        if q_max is not None:
            q_max, _ = strip_unit(q_max, "angstrom^-1")
            keep = self._wavenumbers <= q_max
            self._wavevectors = self._wavevectors[keep]
            self._wavenumbers = self._wavenumbers[keep]

        self._Ns = np.fromiter((getattr(a, f"n_{g}") for a, g in zip(self._groups, self._groupings)),
                               dtype=int, count=self._n_groups)
        self._N = self._Ns.sum()
        self._slices = []
        index = 0
        for N in self._Ns:
            self._slices.append(slice(index, index + N))
            index += N

        self._form = form
        self._sort = sort
        self._unique = unique
        self._verbose = verbose

    def _prepare(self) -> None:
        self.results.pairs = (
            tuple(combinations_with_replacement(range(self._n_groups), 2))
            if self._mode == "partial"
            else ((0, self._n_groups - 1),) if self._mode == "pair"
            else ((None, None),))
        self.results.ssf = np.zeros((len(self.results.pairs), len(self._wavenumbers)))
        self.results.wavenumbers = (np.unique(self._wavenumbers.round(11))
                                    if self._unique else self._wavenumbers)
        self.results.units = {"results.wavenumbers": "angstrom^-1"}
        self._engine = _core.SqEngine(self._wavevectors, self._Ns, self.results.pairs,
                                      dev=self._device)
        self._batch = FrameBatcher(int(self._N), lambda p, b: self._engine.accumulate(p[0]),
                                   with_box=False)
        self._positions = np.empty((self._N, 3), dtype=np.float32)
        self._frames_mine = shard_range(self.n_frames, self._comm.rank, self._comm.world_size)

    def _single_frame(self) -> None:
        lo, hi = self._frames_mine
        if not lo <= self._frame_index < hi:
            return
        for g, gr, s in zip(self._groups, self._groupings, self._slices):
            self._positions[s] = _group_positions(g, gr)
        self._batch.add([self._positions])

    # batched fast path (in-memory and file trajectories, groupings="atoms"): whole blocks of
    # frames go to the engine, gathered in concatenated-group order, instead of one Python
    # iteration per frame.  Shared with IntermediateScatteringFunction.
    def run(self, start=None, stop=None, step=None, frames=None, verbose=None, **kwargs):
        traj = self._trajectory
        if not _is_array_trajectory(traj):
            return super().run(start=start, stop=stop, step=step, frames=frames, verbose=verbose,
                               **kwargs)
        self._setup_frames(traj, start=start, stop=stop, step=step, frames=frames)
        self._prepare()
        numbers = self._frame_numbers()
        self.frames[:] = numbers
        self.times[:] = numbers * traj.dt
        lo, hi = getattr(self, "_frames_mine", (0, len(numbers)))
        mine = numbers[lo:hi]
        if all(g == "atoms" for g in self._groupings):
            index = np.concatenate([np.asarray(g.indices) for g in self._groups])
        elif self._engine is not None:
            # residue / segment centres of mass are formed on the device: rows sorted molecule
            # by molecule; plain-atom groups enter as molecules of one particle and unit mass
            rows, sizes, masses = [], [], []
            for g, gr in zip(self._groups, self._groupings):
                idx, off, m = RadialDistributionFunction._selection(g, gr)
                rows.append(idx)
                sizes.append(np.ones(len(idx), dtype=np.int64) if off is None else np.diff(off))
                masses.append(np.ones(len(idx)) if m is None else m)
            index = np.concatenate(rows)
            self._engine.set_grouping(np.concatenate(([0], np.cumsum(np.concatenate(sizes)))),
                                      np.concatenate(masses))
        else:
            index = np.zeros(0, dtype=int)
        identity = len(index) == traj.n_atoms and np.array_equal(index, np.arange(len(index)))
        native = getattr(traj, "native", None)
        block = 4096 if native is not None else max(self._batch.capacity,
                                                     (1 << 30) // max(12 * traj.n_atoms, 1))
        for b0 in np.arange(0, len(mine), block):
            sel = mine[b0:b0 + block]
            if self._engine is None:       # an ISF rank without wavevectors of its own
                break
            resident = _device_frames(traj, sel) if (identity and not self._engine_has_grouping()) else None
            if native is not None:
                self._engine.accumulate_traj(native, sel, None if identity else index)
            elif resident is not None:
                self._engine.accumulate_device(resident.ptr, traj.n_atoms, len(sel))      # frames already in HBM
            else:
                pos = traj.frame_block(sel)
                self._engine.accumulate(pos if identity else pos[:, index])
        self._conclude()
        return self

    def _engine_has_grouping(self) -> bool:
        return any(g != "atoms" for g in self._groupings)

    def _conclude(self) -> None:
        self._batch.flush()
        if self._comm.world_size > 1 and getattr(self._comm, "device_collectives", False):
            self._engine.allreduce(self._comm)
            ssf = self._engine.result()
        else:
            ssf = self._comm.allreduce(self._engine.result())
        self._engine.close()
        # normalise by particles and frames (reference :1533; N = all particles of all groups)
        self.results.ssf = ssf / (self.n_frames * self._N)
        if self._unique:
            self.results.ssf = _mean_over_equal_wavenumbers(self.results.ssf, self._wavenumbers,
                                                            self.results.wavenumbers)
        if self._sort:
            order = np.argsort(self.results.wavenumbers)
            self.results.wavenumbers = self.results.wavenumbers[order]
            self.results.ssf = self.results.ssf[:, order]
        del self._positions


class IntermediateScatteringFunction(StructureFactor):
    r"""
    Coherent and incoherent (self) intermediate scattering functions and their
    partial versions (reference structure.py:1552-2127):

    .. math::

       F_{\alpha\beta}(q,t)=\frac{1}{N}\left\langle\mathrm{Re}\,
       \rho_\alpha(\mathbf q,t_0)\rho_\beta^*(\mathbf q,t_0+t)\right\rangle_{t_0}
       \;(+\,\alpha\leftrightarrow\beta),\qquad
       F_{\mathrm s,\alpha}(q,t)=\frac{1}{N}\left\langle\sum_{j\in\alpha}
       \cos\mathbf q\cdot[\mathbf r_j(t_0+t)-\mathbf r_j(t_0)]\right\rangle_{t_0}

    Parameters as for :class:`StructureFactor` plus ``dt`` (time between frames),
    ``n_lags`` (number of time lags, default: all analysed frames) and
    ``incoherent``.  Results: ``results.times``, ``results.wavenumbers``,
    ``results.pairs``, ``results.cisf`` ``[N_t, N_pairs or 1, N_q]`` and, when
    ``incoherent=True``, ``results.iisf`` ``[N_t, N_g or 1, N_q]``.

    The reference keeps a ring of ``n_lags`` frames on the host and re-evaluates a
    Fourier sum per (frame, lag) for the incoherent part; here both rings and all
    sums live on the GPU (``mdx_isf_*``).  Frames are consumed in order, so this
    analysis does not shard over frames.
    """

    def __init__(self, groups, groupings: Union[str, tuple] = "atoms", *, mode: str = None,
                 form: str = "exp", dimensions=None, dt=None, n_points: int = 32,
                 n_surfaces: int = None, n_surface_points: int = 8, q_max=None, wavevectors=None,
                 sort: bool = True, unique: bool = True, n_lags: int = None,
                 incoherent: bool = False, parallel: bool = False, verbose: bool = True,
                 **kwargs) -> None:
        super().__init__(groups, groupings, mode=mode, form=form, dimensions=dimensions,
                         n_points=n_points, n_surfaces=n_surfaces,
                         n_surface_points=n_surface_points, q_max=q_max, wavevectors=wavevectors,
                         sort=sort, unique=unique, parallel=parallel, verbose=verbose, **kwargs)
        self._dt = strip_unit(dt or self._trajectory.dt, "picosecond")[0]
        self._n_lags = n_lags
        self._incoherent = incoherent

    def _prepare(self) -> None:
        self._n_lags = self._n_lags or self.n_frames
        st = self._sliced_trajectory
        if hasattr(st, "frames"):
            df = np.diff(st.frames)
            if len(df) and (df[0] <= 0 or not np.allclose(df, df[0])):
                raise ValueError("The selected frames must be evenly spaced and proceed "
                                 "forward in time.")
            df = df[0] if len(df) else 1
        else:
            if st.step is not None and st.step <= 0:
                raise ValueError("The analysis must proceed forward in time.")
            df = st.step if st.step is not None else 1
        self.results.pairs = (
            tuple(combinations_with_replacement(range(self._n_groups), 2))
            if self._mode == "partial"
            else ((0, self._n_groups - 1),) if self._mode == "pair"
            else ((None, None),))
        self.results.times = df * self._dt * np.arange(self._n_lags)
        self.results.wavenumbers = (np.unique(self._wavenumbers.round(11))
                                    if self._unique else self._wavenumbers)
        self.results.units = {"results.times": "picosecond", "results.wavenumbers": "angstrom^-1"}
        # multi-GPU: the lag products of different wavevectors never meet, so every rank sees all
        # frames and owns a contiguous block of wavevectors (frames cannot shard: every lag is needed)
        self._q_mine = shard_range(len(self._wavevectors), self._comm.rank, self._comm.world_size)
        lo, hi = self._q_mine
        self._engine = None
        if hi > lo:
            self._engine = _core.IsfEngine(self._wavevectors[lo:hi], self._Ns, self.results.pairs,
                                           self._n_lags, self._incoherent, dev=self._device)
        self._batch = FrameBatcher(
            int(self._N), lambda p, b: self._engine.accumulate(p[0]) if self._engine else None,
            with_box=False, max_bytes=64 << 20)
        self._positions = np.empty((self._N, 3), dtype=np.float32)

    def _single_frame(self) -> None:
        for g, gr, s in zip(self._groups, self._groupings, self._slices):
            self._positions[s] = _group_positions(g, gr)
        self._batch.add([self._positions])

    def _conclude(self) -> None:
        self._batch.flush()
        n_q = len(self._wavevectors)
        n_p = 1 if self._mode is None else len(self.results.pairs)
        n_s = 1 if self._mode is None else self._n_groups
        cisf = np.zeros((self._n_lags, n_p, n_q))
        iisf = np.zeros((self._n_lags, n_s, n_q)) if self._incoherent else None
        lo, hi = self._q_mine
        if self._engine is not None:
            c, i = self._engine.result()
            self._engine.close()
            cisf[:, :, lo:hi] = c
            if self._incoherent:
                iisf[:, :, lo:hi] = i
        if self._comm.world_size > 1:      # disjoint blocks: the sum is a gather
            cisf = self._comm.allreduce(cisf)
            if self._incoherent:
                iisf = self._comm.allreduce(iisf)
        normalization = (self._N * np.arange(self.n_frames, self.n_frames - self._n_lags, -1)
                         [:, None, None])
        self.results.cisf = cisf / normalization
        if self._incoherent:
            self.results.iisf = iisf / normalization
        if self._unique:
            self.results.cisf = _mean_over_equal_wavenumbers(self.results.cisf, self._wavenumbers,
                                                             self.results.wavenumbers)
            if self._incoherent:
                self.results.iisf = _mean_over_equal_wavenumbers(self.results.iisf, self._wavenumbers,
                                                                 self.results.wavenumbers)
        if self._sort:
            order = np.argsort(self.results.wavenumbers)
            self.results.wavenumbers = self.results.wavenumbers[order]
            self.results.cisf = self.results.cisf[:, :, order]
            if self._incoherent:
                self.results.iisf = self.results.iisf[:, :, order]
        del self._positions
