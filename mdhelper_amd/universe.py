"""
ArrayUniverse — the slice of MDAnalysis' ``Universe`` / ``AtomGroup`` /
``Timestep`` interface that the reference's hot-path classes touch
(SURVEY.md §8b), backed by in-memory NumPy arrays.

MDAnalysis is not a dependency of this package.  The analysis classes are
duck-typed: a real ``MDAnalysis.AtomGroup`` works wherever an ``AtomGroup`` of
this module does.  What the classes use:

``ag.universe``, ``ag.positions`` (float32[N, 3]), ``ag.indices``, ``ag.n_atoms`` /
``n_residues`` / ``n_segments``, ``ag.masses`` / ``charges``, ``ag == other``;
``universe.atoms``, ``universe.dimensions``, ``universe.trajectory`` with ``dt``,
``n_frames``, ``ts``, integer indexing, slicing and iteration;
``ts.frame``, ``ts.time``, ``ts.positions``, ``ts.dimensions``, ``ts.volume``.
"""

from __future__ import annotations

import numpy as np


def box_volumes(boxes) -> np.ndarray:
    """float64 volumes of cells given as rows (lx, ly, lz, alpha, beta, gamma)."""
    b = np.atleast_2d(np.asarray(boxes, dtype=np.float64))
    vol = b[:, 0] * b[:, 1] * b[:, 2]
    tri = ~np.all(b[:, 3:] == 90.0, axis=1)
    if tri.any():
        ca, cb, cg = (np.cos(np.radians(b[tri, k])) for k in (3, 4, 5))
        vol[tri] *= np.sqrt(np.maximum(0.0, 1 - ca * ca - cb * cb - cg * cg + 2 * ca * cb * cg))
    return vol


class Timestep:
    def __init__(self, trajectory, frame: int):
        self._trajectory = trajectory
        self.frame = int(frame)

    @property
    def positions(self):
        return self._trajectory.frame_positions(self.frame)

    @property
    def dimensions(self):
        return self._trajectory.frame_dimensions(self.frame)

    @property
    def volume(self):
        d = self.dimensions
        if d is None:
            return 0.0
        lx, ly, lz, al, be, ga = (float(x) for x in d)
        if al == be == ga == 90.0:
            return lx * ly * lz
        ca, cb, cg = (np.cos(np.radians(x)) for x in (al, be, ga))
        return lx * ly * lz * float(np.sqrt(max(0.0, 1 - ca * ca - cb * cb - cg * cg + 2 * ca * cb * cg)))

    @property
    def time(self):
        return self.frame * self._trajectory.dt

    @property
    def n_atoms(self):
        return self._trajectory.n_atoms


class FrameSelection:
    """What slicing / fancy-indexing a trajectory returns: iterable with ``len``."""

    def __init__(self, trajectory, frames=None, start=None, stop=None, step=None):
        self._trajectory = trajectory
        if frames is not None:
            self.frames = np.asarray(frames, dtype=int)
            self._frames = self.frames
        else:
            self.start, self.stop, self.step = start, stop, step
            self._frames = np.arange(start, stop, step)

    def __len__(self):
        return len(self._frames)

    def __iter__(self):
        for f in self._frames:
            yield self._trajectory[int(f)]


class ArrayTrajectory:
    """In-memory trajectory: positions float32[F, N, 3], dimensions float32[F or 1, 6] or None."""

    def __init__(self, positions, dimensions=None, dt: float = 1.0):
        pos = np.asarray(positions)
        if pos.ndim == 2:
            pos = pos[None]
        if pos.ndim != 3 or pos.shape[2] != 3:
            raise ValueError("positions must have shape (n_frames, n_atoms, 3).")
        self._positions = np.ascontiguousarray(pos, dtype=np.float32)
        if dimensions is not None:
            dim = np.asarray(dimensions, dtype=np.float32)
            if dim.ndim == 1:
                dim = dim[None]
            if dim.shape[1] == 3:
                dim = np.hstack((dim, np.full((dim.shape[0], 3), 90.0, dtype=np.float32)))
            if dim.shape[1] != 6 or dim.shape[0] not in (1, self._positions.shape[0]):
                raise ValueError("dimensions must have shape (6,), (1, 6) or (n_frames, 6).")
            dimensions = np.ascontiguousarray(dim, dtype=np.float32)
        self._dimensions = dimensions
        self.dt = float(dt)
        self.ts = Timestep(self, 0)

    @property
    def n_frames(self):
        return self._positions.shape[0]

    @property
    def n_atoms(self):
        return self._positions.shape[1]

    def frame_positions(self, frame):
        return self._positions[frame]

    def frame_dimensions(self, frame):
        d = self._dimensions
        if d is None:
            return None
        return d[frame if d.shape[0] > 1 else 0].copy()

    def __len__(self):
        return self.n_frames

    def __iter__(self):
        for f in range(self.n_frames):
            yield self[f]

    def __getitem__(self, item):
        if isinstance(item, (int, np.integer)):
            f = int(item)
            if f < 0:
                f += self.n_frames
            if not 0 <= f < self.n_frames:
                raise IndexError(f"frame {item} out of range")
            self.ts = Timestep(self, f)
            return self.ts
        if isinstance(item, slice):
            start, stop, step = item.indices(self.n_frames)
            return FrameSelection(self, start=start, stop=stop, step=step)
        arr = np.asarray(item)
        if arr.dtype == bool:
            arr = np.nonzero(arr)[0]
        return FrameSelection(self, frames=arr)

    def check_slice_indices(self, start, stop, step):
        return slice(start, stop, step).indices(self.n_frames)

    # batched access used by the GPU drivers (zero-copy for contiguous ranges)
    def frame_block(self, frames):
        frames = np.asarray(frames, dtype=int)
        if len(frames) and np.all(np.diff(frames) == 1):
            return self._positions[frames[0]:frames[-1] + 1]
        return self._positions[frames]

    def box_block(self, frames):
        if self._dimensions is None:
            return None
        if self._dimensions.shape[0] == 1:
            return np.broadcast_to(self._dimensions, (len(frames), 6))
        return self._dimensions[np.asarray(frames, dtype=int)]


class DeviceTrajectory(ArrayTrajectory):
    """Trajectory whose frames are resident in HBM: a ``_core.DeviceArray`` float32 or float64
    ``[n_frames, n_atoms, 3]`` (e.g. what a GPU MD engine left there).  ``Onsager`` consumes it without
    a host copy; single frames and host blocks are copied back on demand for everything else."""

    def __init__(self, device_array, dimensions=None, dt: float = 1.0):
        if len(device_array.shape) != 3 or device_array.shape[2] != 3:
            raise ValueError("device positions must have shape (n_frames, n_atoms, 3).")
        if device_array.dtype not in (np.float32, np.float64):
            raise ValueError("device positions must be float32 or float64.")
        self.device_array = device_array
        self._positions = None
        if dimensions is not None:
            dim = np.asarray(dimensions, dtype=np.float32)
            if dim.ndim == 1:
                dim = dim[None]
            if dim.shape[1] == 3:
                dim = np.hstack((dim, np.full((dim.shape[0], 3), 90.0, dtype=np.float32)))
            if dim.shape[1] != 6 or dim.shape[0] not in (1, device_array.shape[0]):
                raise ValueError("dimensions must have shape (6,), (1, 6) or (n_frames, 6).")
            dimensions = np.ascontiguousarray(dim, dtype=np.float32)
        self._dimensions = dimensions
        self.dt = float(dt)
        self.ts = Timestep(self, 0)

    @property
    def n_frames(self):
        return self.device_array.shape[0]

    @property
    def n_atoms(self):
        return self.device_array.shape[1]

    def frame_positions(self, frame):
        return self.device_array.to_host(int(frame), 1)[0].astype(np.float32, copy=False)

    def frame_block(self, frames):
        frames = np.asarray(frames, dtype=int)
        if len(frames) and np.all(np.diff(frames) == 1):
            return self.device_array.to_host(int(frames[0]), len(frames))
        return np.stack([self.device_array.to_host(int(f), 1)[0] for f in frames])

    def device_block(self, frames):
        """The listed frames as a device array without a copy, or None when they are not consecutive."""
        frames = np.asarray(frames, dtype=int)
        if len(frames) and np.all(np.diff(frames) == 1):
            return self.device_array.rows(int(frames[0]), len(frames))
        return None


class _Level:
    """Residue / segment view of an AtomGroup (masses, charges, counts)."""

    def __init__(self, group, ids):
        self._group = group
        self._ids = ids
        self._unique, self._inverse = np.unique(ids, return_inverse=True)

    def __len__(self):
        return len(self._unique)

    @property
    def n_residues(self):
        return len(self._unique)

    n_segments = n_residues

    @property
    def masses(self):
        return np.bincount(self._inverse, weights=self._group.masses, minlength=len(self._unique))

    @property
    def charges(self):
        return np.bincount(self._inverse, weights=self._group.charges, minlength=len(self._unique))


class AtomGroup:
    def __init__(self, universe, indices):
        self.universe = universe
        self.indices = np.asarray(indices, dtype=int)

    @property
    def positions(self):
        return self.universe.trajectory.ts.positions[self.indices]

    @property
    def n_atoms(self):
        return len(self.indices)

    @property
    def n_residues(self):
        return len(np.unique(self.universe._resids[self.indices]))

    @property
    def n_segments(self):
        return len(np.unique(self.universe._segids[self.indices]))

    @property
    def masses(self):
        return self.universe._masses[self.indices]

    @property
    def resindices(self):
        return self.universe._resids[self.indices]

    @property
    def segindices(self):
        return self.universe._segids[self.indices]

    @property
    def residues(self):
        return _Level(self, self.universe._resids[self.indices])

    @property
    def segments(self):
        return _Level(self, self.universe._segids[self.indices])

    @property
    def atoms(self):
        return self

    @property
    def fragments(self):
        """Connected components of the bond graph that hold atoms of this group, each as an
        AtomGroup of ALL its atoms in index order, ordered by first atom (MDAnalysis'
        ``AtomGroup.fragments``).  A universe without bonds: every atom is its own fragment."""
        labels = self.universe._fragment_labels()
        keep = np.unique(labels[self.indices])
        order = np.argsort(labels, kind="stable")
        bounds = np.searchsorted(labels[order], np.arange(labels.max() + 2))
        return tuple(AtomGroup(self.universe, order[bounds[k]:bounds[k + 1]]) for k in keep)

    def __getattr__(self, name):
        if name == "charges":
            if self.universe._charges is None:
                raise AttributeError("This universe has no charges.")
            return self.universe._charges[self.indices]
        raise AttributeError(name)

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, item):
        return AtomGroup(self.universe, np.atleast_1d(self.indices[item]))

    def __eq__(self, other):
        return (isinstance(other, AtomGroup) and other.universe is self.universe
                and np.array_equal(other.indices, self.indices))

    def __hash__(self):
        return hash((id(self.universe), self.indices.tobytes()))


class ArrayUniverse:
    """
    Parameters
    ----------
    positions : float[n_frames, n_atoms, 3]
    dimensions : float[6] | float[n_frames, 6] | None
        (lx, ly, lz, alpha, beta, gamma); three lengths are completed with 90 degree angles.
    dt : float
        Time between frames (ps, or reduced time).
    masses, charges, resids, segids : per-atom arrays, optional
    bonds : int[n_bonds, 2], optional
        Bonded atom pairs; they define ``atoms.fragments`` (what ``Onsager(unwrap=True)`` makes
        whole in the first analysed frame, reference transport.py:936-941).
    """

    def __init__(self, positions, dimensions=None, dt: float = 1.0, *, masses=None,
                 charges=None, resids=None, segids=None, bonds=None):
        self.trajectory = ArrayTrajectory(positions, dimensions, dt)
        self._init_topology(masses, charges, resids, segids, bonds)

    @classmethod
    def from_device(cls, device_array, dimensions=None, dt: float = 1.0, **topology):
        """Universe over frames resident in HBM (``_core.DeviceArray`` float32 / float64
        ``[n_frames, n_atoms, 3]``), see :class:`DeviceTrajectory`."""
        u = cls.__new__(cls)
        u.trajectory = DeviceTrajectory(device_array, dimensions, dt)
        u._init_topology(topology.get("masses"), topology.get("charges"), topology.get("resids"),
                         topology.get("segids"), topology.get("bonds"))
        return u

    def _fragment_labels(self):
        """Fragment number of every atom, fragments numbered by their first atom."""
        if self._fragments is None:
            n = self.trajectory.n_atoms
            if self._bonds is None or len(self._bonds) == 0:
                self._fragments = np.arange(n)
            else:
                from scipy.sparse import coo_matrix
                from scipy.sparse.csgraph import connected_components
                b = self._bonds
                graph = coo_matrix((np.ones(len(b), dtype=np.int8), (b[:, 0], b[:, 1])), shape=(n, n))
                _, raw = connected_components(graph, directed=False)
                # renumber by first appearance, so that fragments come out ordered by first atom
                _, first = np.unique(raw, return_index=True)
                rank = np.empty(len(first), dtype=int)
                rank[np.argsort(first)] = np.arange(len(first))
                self._fragments = rank[raw]
        return self._fragments

    def _init_topology(self, masses, charges, resids, segids, bonds=None):
        self._bonds = None
        self._fragments = None
        if bonds is not None:
            b = np.asarray(bonds, dtype=int).reshape(-1, 2)
            if len(b) and (b.min() < 0 or b.max() >= self.trajectory.n_atoms):
                raise ValueError("bond indices out of range.")
            self._bonds = b
        n = self.trajectory.n_atoms
        self._masses = np.ones(n) if masses is None else np.asarray(masses, dtype=float)
        self._charges = None if charges is None else np.asarray(charges, dtype=float)
        self._resids = np.arange(n) if resids is None else np.asarray(resids, dtype=int)
        self._segids = np.zeros(n, dtype=int) if segids is None else np.asarray(segids, dtype=int)
        self.atoms = AtomGroup(self, np.arange(n))

    @property
    def dimensions(self):
        return self.trajectory.ts.dimensions

    def select(self, mask_or_indices):
        idx = np.asarray(mask_or_indices)
        if idx.dtype == bool:
            idx = np.nonzero(idx)[0]
        return AtomGroup(self, idx)
