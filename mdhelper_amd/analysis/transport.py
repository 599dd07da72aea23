"""
Transport properties on the GPU (operator surface of
``mdhelper.analysis.transport``).

Mirrors reference ``src/mdhelper/analysis/transport.py``:

* ``msd_fft`` / ``msd_shift`` aliases            :31-57
* ``calculate_transport_coefficients``           :59-286
* ``calculate_conductivity``                     :288-336
* ``calculate_electrophoretic_mobility``         :338-395
* ``calculate_transference_number``              :397-433
* ``Onsager``                                    :435-1322

The O(N_t log N_t)-per-particle work of ``Onsager._conclude`` (:1016-1059) runs in
``libmdx.so``: the per-particle self MSDs of a group are accumulated as ONE power
spectrum per (group, block) on the device (``mdx_msd_*``), and the collective /
cross terms are FFT correlations of the summed trajectories (``mdx_correlate``).
Fits and unit conversions are O(N_t) host NumPy exactly as in the reference.
"""

from __future__ import annotations

import itertools
import time
import warnings
from typing import Union

import numpy as np
from scipy import optimize

from .. import _core
from ..algorithm import correlation
from ..algorithm.molecule import center_of_mass
from ..algorithm.topology import make_whole_images, unwrap, wrap
from ..algorithm.unit import strip_unit
from ..comm import shard_range
from .base import SerialAnalysisBase

_KB_KJ_PER_MOL_K = 8.31446261815324e-3      # N_A k_B, kJ/(mol K)
_NA_E2 = 6.02214076e23 * 1.602176634e-19 ** 2   # N_A e^2 (C^2/mol), reference :331-334
_NA_E = 6.02214076e23 * 1.602176634e-19        # N_A e  (C/mol),   reference :391-393
_GROUPINGS = {"atoms", "residues", "segments"}


def msd_fft(*args, **kwargs) -> np.ndarray:
    """Alias of :func:`mdhelper_amd.algorithm.correlation.msd_fft`."""
    return correlation.msd_fft(*args, **kwargs)


def msd_shift(*args, **kwargs) -> np.ndarray:
    """Alias of :func:`mdhelper_amd.algorithm.correlation.msd_shift`."""
    return correlation.msd_shift(*args, **kwargs)


def _poly1(x, a, b):
    """First-order polynomial (reference fit/polynomial.py:82, ``poly1``)."""
    return a * x + b


def _fit_slope(x, y, scale, enforce_linear, what):
    """Slope of y(t) (linear scale) or prefactor of y = c t (log scale)."""
    if scale == "linear":
        return np.polyfit(x, y, 1)[0]
    if scale == "log":
        if enforce_linear:
            return float(np.exp(optimize.curve_fit(lambda t, b: _poly1(t, 1, b),
                                                   np.log(x), np.log(y))[0][0]))
        fit = np.polyfit(np.log(x), np.log(y), 1)
        if abs(1 - fit[0]) >= 0.01:
            warnings.warn(f"The slope for log({what}) vs. log(t) fit is {fit[0]:.6f}.")
        return float(np.exp(fit[1]))
    raise ValueError("Invalid data scaling.")


def calculate_transport_coefficients(
        time, msd_cross, msd_self, Ns, dimensions, kBT: float, start: int = 1, stop: int = None,
        scale: str = "log", *, start_self: int = None, stop_self: int = None,
        scale_self: str = None, enforce_linear: bool = True, verbose: bool = False):
    r"""
    Onsager coefficients :math:`L_{ij}`, their self parts and the self-diffusion
    coefficients :math:`D_i` from the long-time slopes of the (cross) MSDs:

    .. math:: L_{ij}=\frac{1}{k_BTV}\lim_{t\to\infty}\frac{d}{dt}\,\mathrm{MSD}^\mathrm{cross}_{ij}(t),
              \qquad L_{ii}^\mathrm{self}=\frac{N_iD_i}{k_BTV}

    Returns ``(L_ij[N_b, N_g, N_g], L_ii_self[N_b, N_g], D_i[N_b, N_g])``.
    """
    start_self = start if start_self is None else start_self
    stop_self = stop if stop_self is None else stop_self
    scale_self = scale if scale_self is None else scale_self
    msd_cross, msd_self = np.asarray(msd_cross), np.asarray(msd_self)
    time, dimensions = np.asarray(time), np.asarray(dimensions, dtype=float)
    if msd_self.ndim == 2:
        msd_self = msd_self[:, None]
        msd_cross = msd_cross[:, None]
    elif msd_self.ndim != 3:
        raise ValueError("The arrays containing the cross- and self-MSDs have invalid shapes.")
    n_groups, n_blocks = msd_self.shape[:2]
    L_ij = np.zeros((n_blocks, n_groups, n_groups))
    D_i = np.zeros((n_blocks, n_groups))
    rows, cols = np.triu_indices(n_groups)
    denom = kBT * dimensions[~np.isclose(dimensions, 0)].prod()

    def window(y, lo, hi):
        y = y[lo:hi]
        ok = np.isfinite(y) & (y > 0)
        return time[lo:hi][ok], y[ok]

    for b in range(n_blocks):
        for i in range(msd_cross.shape[0]):
            x, y = window(msd_cross[i, b] / denom, start, stop)
            L_ij[b, rows[i], cols[i]] = (_fit_slope(x, y, scale, enforce_linear, "MSDc")
                                         if len(x) > 1 else np.nan)
        L_ij[b] = L_ij[b] + L_ij[b].T - np.diag(np.diag(L_ij[b]))
        for i in range(n_groups):
            x, y = window(msd_self[i, b], start_self, stop_self)
            D_i[b, i] = (_fit_slope(x, y, scale_self, enforce_linear, "MSD")
                         if len(x) > 1 else np.nan)
    return L_ij, np.asarray(Ns) * D_i / denom, D_i


def calculate_conductivity(L_ij, z, *, reduced: bool = False) -> np.ndarray:
    r""":math:`\kappa=\sum_{ij}z_iz_jL_{ij}` (times :math:`N_Ae^2` unless ``reduced``)."""
    z = np.asarray(z, dtype=float)
    kappas = np.einsum("bij,ij->b", np.asarray(L_ij), z * z[:, None])
    return kappas if reduced else kappas * _NA_E2


def calculate_electrophoretic_mobility(L_ij, z, rho, *, reduced: bool = False) -> np.ndarray:
    r""":math:`\mu_i=\sum_jz_jL_{ij}/\rho_i` (times :math:`N_Ae` unless ``reduced``)."""
    z, rho = np.asarray(z, dtype=float), np.asarray(rho, dtype=float)
    mus = (np.asarray(L_ij) * z / rho[:, None]).sum(axis=-1)
    return mus if reduced else mus * _NA_E


def calculate_transference_number(L_ij, z) -> np.ndarray:
    r""":math:`t_i=z_i\sum_jz_jL_{ij}\,/\sum_{kl}z_kz_lL_{kl}`."""
    z = np.asarray(z, dtype=float)
    s = z * (np.asarray(L_ij) * z).sum(axis=-1)
    return s / s.sum(axis=-1, keepdims=True)


class Onsager(SerialAnalysisBase):
    r"""
    Onsager transport framework (Fong et al., Macromolecules 53, 9503 (2020)):
    self and cross mean squared displacements of the particles of several
    groups, and from their slopes :math:`L_{ij}`, :math:`D_i`, conductivity,
    electrophoretic mobilities and transference numbers (reference
    transport.py:435-1322).

    Parameters (as in the reference)
    --------------------------------
    groups : AtomGroup or sequence of AtomGroup
    groupings : {"atoms", "residues", "segments"} or one per group
    temperature : float, default 300 (K; the energy scale when ``reduced``)
    charges, dimensions, dt : keyword-only, optional
    n_blocks : int — trajectory blocks analysed independently
    center, center_atom, center_wrap : bool — subtract the system centre of mass
    fft : bool — FFT algorithm (GPU) or the direct O(N_t^2) definition (host NumPy)
    reduced, unwrap, verbose : bool
    comm : communicator, keyword-only (extension) — *particles* shard across ranks
        (a time correlation needs every lag, so frames cannot); the per-group
        accumulators meet in one all-reduce

    With ``unwrap=True`` every fragment of the first analysed frame is made whole before the starting
    positions are stored (reference transport.py:936-941): a universe that carries bonds
    (``ArrayUniverse(..., bonds=...)`` / ``FileUniverse``) is walked along them
    (``algorithm.topology.make_whole_images``), and the image flags that result are where the
    unwrapping starts — on the host path through ``_positions_old``, on the device paths through
    ``mdx_msd_set_initial_images``.  Displacements, hence every MSD, do not depend on it; what does is
    ``center=True, center_wrap=True`` with molecule groupings when a molecule is split across the
    boundary in that frame.  A universe without bonds has nothing to make whole.

    Results: ``results.pairs``, ``results.times``, ``results.msd_cross``
    ``[N_pairs, N_b, N_t]``, ``results.msd_self`` ``[N_g, N_b, N_t]``, ``results.units``;
    then ``results.L_ij``, ``results.L_ii_self``, ``results.D_i``,
    ``results.conductivities``, ``results.electrophoretic_mobilities``,
    ``results.transference_numbers`` after the ``calculate_*`` methods.
    """

    def __init__(self, groups, groupings: Union[str, tuple] = "atoms", temperature=300, *,
                 charges=None, dimensions=None, dt=None, n_blocks: int = 1, center: bool = False,
                 center_atom: bool = False, center_wrap: bool = False, fft: bool = True,
                 reduced: bool = False, unwrap: bool = False, verbose: bool = True,
                 **kwargs) -> None:
        self._groups = [groups] if hasattr(groups, "universe") else list(groups)
        self.universe = self._groups[0].universe
        super().__init__(self.universe.trajectory, verbose=verbose, **kwargs)

        self._n_groups = len(self._groups)
        if isinstance(groupings, str):
            if groupings not in _GROUPINGS:
                raise ValueError(f"Invalid grouping '{groupings}'. Valid values: "
                                 f"{', '.join(sorted(_GROUPINGS))}.")
            self._groupings = self._n_groups * [groupings]
        else:
            if self._n_groups != len(groupings):
                raise ValueError("The number of grouping values is not equal to the "
                                 "number of groups.")
            for g in groupings:
                if g not in _GROUPINGS:
                    raise ValueError(f"Invalid grouping '{g}'. Valid values: "
                                     f"{', '.join(sorted(_GROUPINGS))}.")
            self._groupings = list(groupings)

        temperature, unit_ = strip_unit(temperature, "temperature")
        if reduced:
            if not isinstance(unit_, str):
                raise TypeError("'temperature' cannot have units when reduced=True.")
            self._kBT = temperature
        else:
            self._kBT = _KB_KJ_PER_MOL_K * temperature

        if dimensions is not None:
            if len(dimensions) != 3:
                raise ValueError("'dimensions' must have length 3.")
            self._dimensions = np.asarray(strip_unit(dimensions, "angstrom")[0], dtype=float)
        elif self.universe.dimensions is not None:
            self._dimensions = np.array(self.universe.dimensions[:3], dtype=float)
        else:
            raise ValueError("No system dimensions found or provided.")

        self._dt, unit_ = strip_unit(dt or self._trajectory.dt, "picosecond")
        if reduced and not isinstance(unit_, str):
            raise TypeError("'dt' cannot have units when reduced=True.")

        if charges is not None:
            self._set_charges(charges, reduced)
        else:
            try:
                self._charges = np.fromiter(
                    (getattr(g, gr).charges[0] for g, gr in zip(self._groups, self._groupings)),
                    count=self._n_groups, dtype=float)
            except (AttributeError, IndexError, TypeError):
                self._charges = None

        self._Ns = tuple(getattr(a, f"n_{g}") for a, g in zip(self._groups, self._groupings))
        self._N = sum(self._Ns)
        self._slices = []
        index = 0
        for N in self._Ns:
            self._slices.append(slice(index, index + N))
            index += N

        self._rhos = None
        if np.all(~np.isclose(self._dimensions, 0)):
            self._rhos = np.asarray(self._Ns, dtype=float) / self._dimensions.prod()

        self._n_blocks = n_blocks
        self._center = center
        self._center_atom = center_atom
        self._center_wrap = center_wrap
        self._fft = fft
        self._reduced = reduced
        self._unwrap = unwrap
        self._verbose = verbose

    def _set_charges(self, charges, reduced):
        if len(charges) != self._n_groups:
            raise ValueError("The number of group charges is not equal to the number of groups.")
        charges, unit_ = strip_unit(charges, "elementary_charge")
        if reduced and not isinstance(unit_, str):
            raise TypeError("'charges' cannot have units when reduced=True.")
        self._charges = np.asarray(charges, dtype=float)

    # ------------------------------------------------------------------ protocol

    def _prepare(self) -> None:
        st = self._sliced_trajectory
        if hasattr(st, "frames"):
            df = np.diff(st.frames)
            if len(df) and (df[0] <= 0 or not np.allclose(df, df[0])):
                raise ValueError("The selected frames must be evenly spaced and proceed "
                                 "forward in time.")
        elif hasattr(st, "step") and st.step is not None and st.step <= 0:
            raise ValueError("The analysis must proceed forward in time.")

        self.results.pairs = tuple(
            itertools.combinations_with_replacement(range(self._n_groups), 2))

        # particle shard of every group owned by this rank (all of them on one rank)
        rank, world = self._comm.rank, self._comm.world_size
        self._own = [shard_range(n, rank, world) for n in self._Ns]
        self._own_slices = []
        index = 0
        for lo, hi in self._own:
            self._own_slices.append(slice(index, index + hi - lo))
            index += hi - lo
        self._from_file = getattr(self, "_from_file", False)
        self._positions = None if self._from_file else np.empty((self.n_frames, index, 3))

        self._images0 = None
        if self._unwrap:
            # every fragment of the first analysed frame is made whole before the starting positions
            # are stored (reference :936-941); universes without bonds have nothing to make whole
            first = st.frames[0] if hasattr(st, "frames") else (self.start or 0)
            self.universe.trajectory[first]
            images0 = make_whole_images(self.universe, self._dimensions)
            if images0.any():
                self._images0 = images0
            if not self._from_file:
                self._positions_old = (np.array(self.universe.atoms.positions, dtype=float)
                                       + images0 * self._dimensions)
                self._images = np.zeros((self.universe.atoms.n_atoms, 3), dtype=int)
                self._thresholds = self._dimensions / 2

        self._n_frames_block = self.n_frames // self._n_blocks
        self._n_frames = self._n_blocks * self._n_frames_block
        extra = self.n_frames - self._n_frames
        if extra > 0:
            warnings.warn(f"The trajectory is not divisible into {self._n_blocks:,} blocks, so "
                          f"the last {extra:,} frame(s) will be discarded. To maximize "
                          "performance, set appropriate starting and ending frames in run() so "
                          "that the number of frames to be analyzed is divisible by the number "
                          "of blocks.")

        step = self.step if self.step is not None else 1
        self.results.times = step * self._dt * np.arange(self._n_frames // self._n_blocks)
        self.results.msd_cross = np.empty(
            (len(self.results.pairs), self._n_blocks, self._n_frames_block), dtype=float)
        self.results.msd_self = np.empty(
            (self._n_groups, self._n_blocks, self._n_frames_block), dtype=float)
        self.results.units = {"results.times": "picosecond",
                              "results.msd_cross": "angstrom^2", "results.msd_self": "angstrom^2"}

    def _single_frame(self) -> None:
        positions = np.array(self.universe.atoms.positions, dtype=float)
        if self._unwrap:
            unwrap(positions, self._positions_old, self._dimensions,
                   thresholds=self._thresholds, images=self._images)

        frame = np.empty((self._N, 3)) if self._center else None
        for g, gr, s, own, (lo, hi) in zip(self._groups, self._groupings, self._slices,
                                           self._own_slices, self._own):
            group_pos = (positions[g.indices] if gr == "atoms"
                         else center_of_mass(g, gr, images=self._images[g.indices]
                                             if hasattr(self, "_images") else None))
            if frame is not None:
                frame[s] = group_pos
            self._positions[self._frame_index, own] = group_pos[lo:hi]

        # subtract the system centre of mass (reference :993-1014)
        if self._center:
            if self._center_atom:
                if self._center_wrap:
                    wrap(positions, self._dimensions)
                scom = center_of_mass(positions=positions, masses=self.universe.atoms.masses)
            else:
                ref = wrap(frame, self._dimensions, in_place=False) if self._center_wrap else frame
                scom = center_of_mass(
                    positions=ref,
                    masses=np.concatenate([getattr(g, gr).masses
                                           for g, gr in zip(self._groups, self._groupings)]))
            self._positions[self._frame_index] -= scom

    def _conclude(self) -> None:
        if self.n_frames != self._n_frames and self._positions is not None:
            self._positions = self._positions[:self._n_frames]
        delete_dimensions = np.isclose(self._dimensions, 0)
        zero_mask = int(sum(1 << k for k in range(3) if delete_dimensions[k]))
        B, Tb = self._n_blocks, self._n_frames_block
        multi = self._comm.world_size > 1

        # wall-clock split of one analysis for bench.py (off unless _profile is set: the marks wait for the device)
        self._timings, mark_t = {}, [time.perf_counter()]

        def mark(phase):
            if self._profile:
                _core.synchronize(self._device)
                now = time.perf_counter()
                self._timings[phase] = self._timings.get(phase, 0.0) + now - mark_t[0]
                mark_t[0] = now

        if self._fft:
            eng = _core.MsdEngine(Tb, B, self._n_groups, dev=self._device)
            mark("engine_create")
            if self._from_file:
                # The analysed frames are brought into HBM ONCE (trajectory file -> pinned ring -> HBM;
                # host memory through the ring's copy threads, or by DMA where it is page-locked; a
                # DeviceTrajectory is there already) and every group, and the system centre of mass, is
                # prepared from them on the device: particle gather, unwrapping, float64 widening,
                # molecule centres, shift (mdx_msd_push_frames_device).  Frames that would not leave room
                # for the engine stream group by group instead (mdx_msd_push_traj / _f32).
                numbers = self._frame_numbers()[:self._n_frames]
                native = getattr(self._trajectory, "native", None)
                unwrap_dims = self._dimensions if self._unwrap else None
                streamed = self._stream_host_groups(eng, numbers, native, zero_mask, unwrap_dims)
                resident = None if streamed else self._resident_frames(numbers, native)
                mark("frames_to_hbm")
                if resident is not None:
                    n_total = resident.shape[1]

                    def system_com(rows, masses, **kw):
                        return eng.system_com_device(resident, n_total, rows, masses, **kw)

                    def push(g, rows, **kw):
                        plain = (not eng.has_grouping and kw.get("shift") is None
                                 and kw.get("unwrap_dims") is None and len(rows)
                                 and np.array_equal(rows, np.arange(rows[0], rows[0] + len(rows))))
                        if plain and resident.dtype == np.float64:
                            # float64 frames in HBM, a contiguous range of particles, nothing to prepare:
                            # the correlation kernels read them where they lie
                            eng.push_device(g, resident.ptr, n_total, int(rows[0]), len(rows),
                                            kw.get("zero_dims", 0))
                        elif plain and resident.dtype == np.float32 and eng.reads_f32:
                            # ... and so do float32 frames where the first pass widens them as it stages them
                            # (mdx_msd_push_device_f32: the 400 x R2 transforms; no float64 copy of the group)
                            eng.push_device_f32(g, resident.ptr, n_total, int(rows[0]), len(rows),
                                                kw.get("zero_dims", 0))
                        else:
                            eng.push_frames_device(g, resident, n_total, rows, **kw)
                elif native is not None:
                    def system_com(rows, masses, **kw):
                        return eng.system_com_traj(native, numbers, rows, masses, **kw)

                    def push(g, rows, **kw):
                        eng.push_traj(g, native, numbers, rows, **kw)
                else:
                    # in-memory float32 frames: the same device stages, fed from host memory
                    block = self._trajectory.frame_block(numbers)

                    def system_com(rows, masses, **kw):
                        return eng.system_com_f32(block if rows is None else block[:, rows], masses, **kw)

                    def push(g, rows, **kw):
                        eng.push_f32(g, block[:, rows], **kw)
                def start_images(rows):
                    # image flags the rows start from (molecules made whole in the first frame)
                    # (the reference's first unwrap call moves a flag by sign(x - x_whole) only, reference
                    # topology.py:366-376: a particle k >= 2 images away from its made-whole place starts at
                    # +-1 like the host path's, not at k)
                    if self._unwrap:
                        eng.set_initial_images(None if self._images0 is None else
                                               np.sign(self._images0 if rows is None else self._images0[rows]))

                shift = None
                if self._center:
                    # system centre of mass per frame (reference :993-1014), every rank the same
                    wrap_dims = self._dimensions if self._center_wrap else None
                    if self._center_atom:
                        start_images(None)
                        shift = system_com(None, self.universe.atoms.masses,
                                           unwrap_dims=unwrap_dims, wrap_dims=wrap_dims)
                    elif wrap_dims is None or all(gr == "atoms" for gr in self._groupings):
                        start_images(np.concatenate([g.indices for g in self._groups]))
                        shift = system_com(np.concatenate([g.indices for g in self._groups]),
                                           np.concatenate([g.masses for g in self._groups]),
                                           unwrap_dims=unwrap_dims, wrap_dims=wrap_dims)
                    else:
                        # wrapped CENTRES of the molecules (reference :1004-1014: wrap(frame) acts on
                        # the residue / segment centres): per group on the device, mass-weighted here
                        from .structure import RadialDistributionFunction as _R
                        num, mass = 0.0, 0.0
                        for grp, gr in zip(self._groups, self._groupings):
                            if gr == "atoms":
                                eng.set_grouping(None, None)
                                rows, m = grp.indices, np.asarray(grp.masses, dtype=float)
                            else:
                                rows, off, m = _R._selection(grp, gr)
                                eng.set_grouping(off, m)
                            start_images(rows)
                            com = system_com(rows, m, unwrap_dims=unwrap_dims, wrap_dims=wrap_dims)
                            num = num + com * m.sum()
                            mass += m.sum()
                        eng.set_grouping(None, None)
                        shift = num / mass
                from .structure import RadialDistributionFunction
                for g, (grp, gr, (lo, hi)) in enumerate(zip(self._groups, self._groupings, self._own)):
                    if hi <= lo or streamed:
                        continue
                    if gr == "atoms":
                        eng.set_grouping(None, None)
                        rows = grp.indices[lo:hi]
                    else:
                        # molecules lo .. hi of this rank: rows sorted molecule by molecule, the
                        # float64 centres of the unwrapped particles are formed on the device
                        idx, off, m = RadialDistributionFunction._selection(grp, gr)
                        eng.set_grouping(off[lo:hi + 1] - off[lo], m[off[lo]:off[hi]])
                        rows = idx[off[lo]:off[hi]]
                    start_images(rows)
                    push(g, rows, unwrap_dims=unwrap_dims, zero_dims=zero_mask, shift=shift)
                if resident is not None and getattr(resident, "base", None) is None:
                    _core.synchronize(self._device)
                    resident.free()
            for g, own in enumerate(self._own_slices):
                if own.stop > own.start and not self._from_file:
                    eng.push(g, self._positions, own.start, own.stop - own.start, zero_mask)
            mark("prepare_and_push")
            cross = None
            if multi and getattr(self._comm, "device_collectives", False):
                eng.allreduce(self._comm)
                self_sum, traj = eng.result()
            else:
                self_sum, traj = eng.result()
                if multi:
                    self_sum = self._comm.allreduce(self_sum)
                    traj = self._comm.allreduce(traj)
            mark("result")
            if hasattr(eng, "cross") and (not multi or getattr(self._comm, "device_collectives", False)):
                # the summed trajectories of every rank are in HBM: all pairs in one batch
                cross = eng.cross(self.results.pairs)
            eng.close()
            msd = correlation.msd_fft
        else:
            # direct definition: per-particle MSDs and summed trajectories on the host
            cross = None
            self_sum = np.zeros((self._n_groups, B, Tb))
            traj = np.zeros((self._n_groups, B, Tb, 3))
            for g, own in enumerate(self._own_slices):
                if own.stop > own.start:
                    p = self._positions[:, own].reshape((B, Tb, own.stop - own.start, 3)).copy()
                    p[..., delete_dimensions] = 0
                    self_sum[g] = correlation.msd_shift(p, axis=1, average=False).sum(axis=-1)
                    traj[g] = p.sum(axis=2)
            if multi:
                self_sum = self._comm.allreduce(self_sum)
                traj = self._comm.allreduce(traj)
            msd = correlation.msd_shift

        for i, (i1, i2) in enumerate(self.results.pairs):
            if i1 == i2:
                if self._Ns[i1]:
                    self.results.msd_cross[i] = cross[i] if cross is not None else msd(traj[i1], axis=1)
                    self.results.msd_self[i1] = self_sum[i1] / self._Ns[i1]
                else:
                    self.results.msd_cross[i] = self.results.msd_self[i1] = np.nan
            elif self._Ns[i1] and self._Ns[i2]:
                self.results.msd_cross[i] = (cross[i] if cross is not None
                                             else msd(traj[i1], traj[i2], axis=1))
            else:
                self.results.msd_cross[i] = np.nan

        # account for dimensionality (reference :1056-1059)
        D = 2 * (~delete_dimensions).sum()
        self.results.msd_cross /= D
        self.results.msd_self /= D
        mark("cross_msds")

    def _stream_host_groups(self, eng, numbers, native, zero_mask, unwrap_dims) -> bool:
        """
        Plain atom groups over consecutive particles of an in-memory float32 trajectory, nothing to centre: a
        group's particles are all the engine needs of a frame, so they travel in column chunks — particles
        [a, a + c) of every frame through ``mdx_upload_rows`` (2-D copies the DMA engine reads out of the caller's
        pages) into one of two device blocks — and the chunk before is unwrapped, widened and transformed on the
        device meanwhile.  The analysis then takes the time of the host link plus one chunk (C4 from pageable
        memory: 225 - 235 ms, the link alone 208 ms) instead of upload + transforms.  Returns False when the case is
        another one (the caller brings whole frames into HBM; trajectory files: NOTES round 5 on why their pages
        are not handed to the DMA engine).
        """
        traj = self._trajectory
        if (not self._stream_columns or native is not None or self._center or self._comm.world_size != 1
                or hasattr(traj, "device_block")
                or self._hbm_share <= 0 or any(gr != "atoms" for gr in self._groupings)):
            return False
        block = traj.frame_block(numbers)
        if not (isinstance(block, np.ndarray) and block.dtype == np.float32 and block.flags.c_contiguous):
            return False
        spans = []
        for grp in self._groups:
            idx = np.asarray(grp.indices)
            if len(idx) == 0 or not np.array_equal(idx, np.arange(idx[0], idx[0] + len(idx))):
                return False
            spans.append((int(idx[0]), len(idx)))
        T, n_atoms = block.shape[0], block.shape[1]
        # ~1.5 GB per chunk, at least two chunks per group (the second upload hides the first chunk's transforms),
        # equal chunks of whole groups of 16 particles
        largest = max(c for _f, c in spans)
        n_chunks = max(2, -(-largest * 12 * T // (3 << 29)))
        chunk = max(16, -(-(-(-largest // n_chunks)) // 16) * 16)
        free = _core.device_info(self._device)["hbm_available_bytes"]
        if 2 * 12 * T * chunk > self._hbm_share * free:
            return False
        bufs = [_core.DeviceArray((T, chunk, 3), np.float32, self._device) for _ in range(2)]
        k = 0
        try:
            eng.set_grouping(None, None)
            for g, (first, count) in enumerate(spans):
                for a in range(0, count, chunk):
                    c = min(chunk, count - a)
                    # (a ragged last chunk: the same block, viewed as [T, c, 3])
                    buf = bufs[k & 1] if c == chunk else _core.DeviceArray.view(bufs[k & 1], (T, c, 3))
                    buf.upload_columns(block, first + a, c)
                    # the chunk pushed before this upload ran beside it; once it is done, the block it read is free
                    # for the upload after this one, and this chunk's kernels run beside that upload
                    eng.synchronize()
                    if self._unwrap:
                        eng.set_initial_images(None if self._images0 is None else
                                               np.sign(self._images0[first + a:first + a + c]))
                    if unwrap_dims is None and eng.reads_f32:
                        # nothing to prepare: the first pass reads the chunk's float32 rows where they lie
                        eng.push_device_f32(g, buf.ptr, c, 0, c, zero_mask)
                    else:
                        eng.push_frames_device(g, buf, c, None, unwrap_dims=unwrap_dims, zero_dims=zero_mask)
                    k += 1
            eng.synchronize()
        finally:
            for b in bufs:
                b.free()
        return True

    _stream_columns = True   # tests: False keeps whole frames in HBM where column chunks would be streamed
    _profile = False    # bench.py: split one analysis into phases (self._timings)
    _hbm_share = 0.3    # of the free HBM the analysed float32 frames may take to be kept whole

    def _resident_frames(self, numbers, native):
        """The analysed frames as ONE device array ``[T, n_atoms, 3]`` (float32, or the float64 of a
        DeviceTrajectory), or None when they would take more than 30 % of the free HBM (or particles
        shard across ranks from host memory: a rank then stages its own rows only)."""
        traj = self._trajectory
        dev = self._device
        if hasattr(traj, "device_block"):
            view = traj.device_block(numbers)
            if view is not None:
                return view
        elif self._comm.world_size > 1:
            return None
        need = len(numbers) * traj.n_atoms * 12
        if self._hbm_share <= 0 or need > self._hbm_share * _core.device_info(dev)["hbm_available_bytes"]:
            return None
        if native is not None:
            out = _core.DeviceArray((len(numbers), traj.n_atoms, 3), np.float32, dev)
            try:
                native.load_device(numbers, out.ptr, dev=dev)
            except Exception:
                out.free()
                raise
            return out
        return _core.DeviceArray.upload(traj.frame_block(numbers), dev)

    # in-memory trajectories, plain atom groups: copy the positions in one vectorised step
    def run(self, start=None, stop=None, step=None, frames=None, n_jobs: int = 1, verbose=None,
            **kwargs):
        traj = self._trajectory
        atoms_only = all(g == "atoms" for g in self._groupings)
        # trajectory files: nothing is staged on the host — unwrapping, residue / segment centres
        # of mass and the removal of the system centre of mass (of atoms, of group particles, or of
        # the wrapped molecule centres) happen on the device
        native = getattr(traj, "native", None) is not None
        # ... and so do in-memory frames (what an MDAnalysis memory reader holds) and frames resident in HBM
        batched = hasattr(traj, "frame_block") or hasattr(traj, "device_block")
        self._from_file = bool(self._fft and (native or batched))
        fast = self._from_file or (hasattr(traj, "frame_block") and atoms_only
                                   and not (self._unwrap or self._center))
        if not fast:
            return super().run(start=start, stop=stop, step=step, frames=frames, n_jobs=n_jobs,
                               verbose=verbose, **kwargs)
        self._setup_frames(traj, start=start, stop=stop, step=step, frames=frames)
        self._prepare()
        numbers = self._frame_numbers()
        self.frames[:] = numbers
        self.times[:] = numbers * traj.dt
        if self._from_file:
            self._conclude()
            return self
        block = traj.frame_block(numbers)
        for g, own, (lo, hi) in zip(self._groups, self._own_slices, self._own):
            self._positions[:, own] = block[:, g.indices[lo:hi]]
        self._conclude()
        return self

    # ------------------------------------------------------------ post-processing

    def calculate_transport_coefficients(self, start: int = 1, stop: int = None,
                                         scale: str = "log", *, start_self: int = None,
                                         stop_self: int = None, scale_self: str = None,
                                         enforce_linear: bool = True) -> None:
        if "msd_cross" not in self.results:
            raise RuntimeError("Call Onsager.run() before "
                               "Onsager.calculate_transport_coefficients().")
        self.results.L_ij, self.results.L_ii_self, self.results.D_i = \
            calculate_transport_coefficients(
                self.results.times, self.results.msd_cross, self.results.msd_self, self._Ns,
                self._dimensions, self._kBT, start, stop, scale, start_self=start_self,
                stop_self=stop_self, scale_self=scale_self, enforce_linear=enforce_linear,
                verbose=self._verbose)
        if not self._reduced:
            self.results.units["results.D_i"] = "angstrom^2 / picosecond"
            self.results.units["results.L_ii_self"] = self.results.units["results.L_ij"] = \
                "mole / (kilojoule * angstrom * picosecond)"

    def _need_L(self, who):
        if "L_ij" not in self.results:
            raise RuntimeError("Call Onsager.calculate_transport_coefficients() before "
                               f"Onsager.{who}().")

    def _charges_or_raise(self, charges):
        if charges is not None:
            self._set_charges(charges, self._reduced)
        if self._charges is None:
            raise ValueError("No charge number information available.")
        return self._charges

    def calculate_conductivity(self, *, charges=None) -> None:
        self._need_L("calculate_conductivity")
        z = self._charges_or_raise(charges)
        self.results.conductivities = calculate_conductivity(self.results.L_ij, z,
                                                             reduced=self._reduced)
        self.results.units["results.conductivities"] = \
            "coulomb^2 / (kilojoule * angstrom * picosecond)"

    def calculate_electrophoretic_mobility(self, *, charges=None, rhos=None) -> None:
        self._need_L("calculate_electrophoretic_mobility")
        z = self._charges_or_raise(charges)
        if rhos is not None:
            if len(rhos) != self._n_groups:
                raise ValueError("The number of group number densities is not equal to the "
                                 "number of groups.")
            rhos, unit_ = strip_unit(rhos, "angstrom**-3")
            if self._reduced and not isinstance(unit_, str):
                raise TypeError("'rhos' cannot have units when reduced=True.")
            self._rhos = np.asarray(rhos, dtype=float)
        if self._rhos is None:
            raise ValueError("No number density information available.")
        self.results.electrophoretic_mobilities = calculate_electrophoretic_mobility(
            self.results.L_ij, z, self._rhos, reduced=self._reduced)
        self.results.units["results.electrophoretic_mobilities"] = \
            "angstrom^2 * coulomb / (kilojoule * picosecond)"

    def calculate_transference_number(self, *, charges=None) -> None:
        self._need_L("calculate_transference_number")
        z = self._charges_or_raise(charges)
        self.results.transference_numbers = calculate_transference_number(self.results.L_ij, z)
