"""
mdhelper_amd.io — trajectory files read by the native ingest of ``libmdx.so``.

The reference analyses whatever ``MDAnalysis.Universe`` hands it, one Python
``Timestep`` at a time (``universe.trajectory[frame]``, reference
structure.py:796, transport.py:976-985); its own simulation writer produces
AMBER NetCDF trajectories (reference openmm/file.py:49-52, 160-188).  Here the
file is parsed by ``mdx_traj_open`` (``csrc/mdx_traj.hip``): AMBER NetCDF
(classic / 64-bit-offset containers) or CHARMM/NAMD DCD.  A
:class:`FileUniverse` offers the same duck-typed surface as
:class:`~mdhelper_amd.universe.ArrayUniverse`; the analysis classes detect the
native handle and stream frames file → pinned memory → HBM without a Python
loop over frames.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import byref, c_int, c_int64, c_void_p

import numpy as np

from ._lib import check, lib
from .universe import ArrayUniverse, FrameSelection, Timestep

FORMATS = {1: "NETCDF", 2: "DCD"}


def _ptr(a):
    return None if a is None else a.ctypes.data_as(c_void_p)


class TrajectoryFile:
    """Thin owner of an ``mdx_traj_t``."""

    def __init__(self, path):
        self.path = os.fspath(path)
        self.handle = c_void_p()
        check(lib().mdx_traj_open(byref(self.handle), self.path.encode()))
        nf, na, hb, ht, fmt = c_int64(), c_int64(), c_int(), c_int(), c_int()
        check(lib().mdx_traj_info(self.handle, byref(nf), byref(na), byref(hb), byref(ht), byref(fmt)))
        self.n_frames, self.n_atoms = nf.value, na.value
        self.has_box, self.has_time = bool(hb.value), bool(ht.value)
        self.format = FORMATS.get(fmt.value, "?")

    def close(self):
        if self.handle:
            lib().mdx_traj_close(self.handle)
            self.handle = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _frames(frames):
        return np.ascontiguousarray(np.atleast_1d(frames), dtype=np.int64)

    def read_positions(self, frames):
        f = self._frames(frames)
        out = np.empty((len(f), self.n_atoms, 3), dtype=np.float32)
        check(lib().mdx_traj_read_positions(self.handle, _ptr(f), len(f), _ptr(out)))
        return out

    def read_boxes(self, frames):
        f = self._frames(frames)
        out = np.empty((len(f), 6), dtype=np.float32)
        check(lib().mdx_traj_read_boxes(self.handle, _ptr(f), len(f), _ptr(out)))
        return out

    def read_times(self, frames):
        f = self._frames(frames)
        out = np.empty(len(f), dtype=np.float64)
        check(lib().mdx_traj_read_times(self.handle, _ptr(f), len(f), _ptr(out)))
        return out

    def load_device(self, frames, d_out, *, d_index=None, n_sel=0, dev=0):
        """Frames into HBM: ``d_out`` float32[len(frames)][n_sel or n_atoms][3] (raw pointer)."""
        f = self._frames(frames)
        check(lib().mdx_traj_load_device(self.handle, dev, _ptr(f), len(f), d_index, n_sel, d_out))


class FileTrajectory:
    """``universe.trajectory`` over a :class:`TrajectoryFile`."""

    def __init__(self, path, dt=None):
        self.file = TrajectoryFile(path)
        if dt is None:
            dt = 1.0
            if self.file.has_time and self.file.n_frames > 1:
                t = self.file.read_times([0, 1])
                if t[1] > t[0]:
                    dt = float(t[1] - t[0])
        self.dt = float(dt)
        self._cache_frame = -1
        self._cache_positions = None
        self.ts = Timestep(self, 0)

    @property
    def native(self) -> TrajectoryFile:
        return self.file

    @property
    def n_frames(self):
        return self.file.n_frames

    @property
    def n_atoms(self):
        return self.file.n_atoms

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

    def frame_positions(self, frame):
        if frame != self._cache_frame:
            self._cache_positions = self.file.read_positions([frame])[0]
            self._cache_frame = frame
        return self._cache_positions

    def frame_dimensions(self, frame):
        if not self.file.has_box:
            return None
        return self.file.read_boxes([frame])[0]

    # batched access used by the GPU drivers
    def frame_block(self, frames):
        return self.file.read_positions(frames)

    def box_block(self, frames):
        if not self.file.has_box:
            return None
        return self.file.read_boxes(frames)


class FileUniverse(ArrayUniverse):
    """
    Universe over an AMBER NetCDF or DCD trajectory file.

    Parameters
    ----------
    path : str
    dt : float, optional
        Time between frames; by default the spacing of the first two stored times.
    masses, charges, resids, segids, bonds : optional (as for ``ArrayUniverse``)
    """

    def __init__(self, path, dt=None, *, masses=None, charges=None, resids=None, segids=None,
                 bonds=None):
        self.trajectory = FileTrajectory(path, dt)
        self._init_topology(masses, charges, resids, segids, bonds)
