"""
Analysis base classes (operator surface of ``mdhelper.analysis.base``).

Mirrors reference ``src/mdhelper/analysis/base.py``: ``Hash`` (:72-113),
``SerialAnalysisBase`` (:115-210), ``NumbaAnalysisBase`` (:212-279),
``ParallelAnalysisBase`` (:281-507) and ``DynamicAnalysisBase`` (:509-584) keep
their names, ``run()`` signatures and ``save()``; underneath sits a small
restatement of ``MDAnalysis.analysis.base.AnalysisBase`` (frame selection,
``_prepare`` / ``_single_frame`` / ``_conclude`` protocol, ``results``), because
MDAnalysis is not a dependency here.

What changes is where the frames go: the classes built on these bases hand
*batches* of frames to a HIP engine instead of doing per-frame NumPy work, so
the process-pool / joblib / dask / numba-thread machinery of the reference has
nothing left to parallelise.  The keyword arguments that steered it
(``n_jobs``, ``module``, ``block``, ``method``, ``n_threads``) are accepted and
ignored.  Scaling past one GPU is by frame (or particle) sharding across ranks,
one process per GPU, with one all-reduce at the end (``comm=``).
"""

from __future__ import annotations

import logging
from datetime import datetime
from typing import Any, TextIO, Union

import numpy as np

from ..comm import SerialComm


class Hash(dict):
    """``dict`` with attribute access (reference base.py:72-113)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for arg in args:
            if not isinstance(arg, dict):
                raise TypeError("Positional arguments must be dictionaries.")
            for k, v in arg.items():
                self[k] = v
        for k, v in kwargs.items():
            self[k] = v

    def __getattr__(self, attr):
        if attr.startswith("__"):
            raise AttributeError(attr)
        return self.get(attr)

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, item):
        del self[item]

    def __contains__(self, key):
        return dict.__contains__(self, key)


class AnalysisBase:
    """
    Frame-iteration protocol of ``MDAnalysis.analysis.base.AnalysisBase``:
    ``run()`` = ``_setup_frames`` → ``_prepare`` → per frame (``_frame_index``,
    ``_ts``, ``frames[i]``, ``times[i]``, ``_single_frame``) → ``_conclude``.
    """

    def __init__(self, trajectory, verbose: bool = False, **kwargs):
        self._trajectory = trajectory
        self._verbose = verbose
        self.results = Hash()
        self._comm = kwargs.pop("comm", None) or SerialComm()
        self._device = int(kwargs.pop("device", 0))

    def _setup_frames(self, trajectory, start=None, stop=None, step=None, frames=None):
        if frames is not None:
            if not all(x is None for x in (start, stop, step)):
                raise ValueError("start/stop/step cannot be combined with frames")
            slicer = frames
        else:
            start, stop, step = trajectory.check_slice_indices(start, stop, step)
            slicer = slice(start, stop, step)
        self._sliced_trajectory = trajectory[slicer]
        self.start, self.stop, self.step = start, stop, step
        self.n_frames = len(self._sliced_trajectory)
        self.frames = np.zeros(self.n_frames, dtype=int)
        self.times = np.zeros(self.n_frames)

    def _frame_numbers(self) -> np.ndarray:
        """Trajectory frame numbers selected by ``_setup_frames``."""
        st = self._sliced_trajectory
        if hasattr(st, "frames"):
            return np.asarray(st.frames, dtype=int)
        return np.arange(self.start, self.stop, self.step)

    def _prepare(self):
        pass

    def _single_frame(self):
        raise NotImplementedError

    def _conclude(self):
        pass

    def run(self, start=None, stop=None, step=None, frames=None, verbose=None, **kwargs):
        verbose = getattr(self, "_verbose", False) if verbose is None else verbose
        self._setup_frames(self._trajectory, start=start, stop=stop, step=step, frames=frames)
        self._prepare()
        t0 = datetime.now()
        for i, ts in enumerate(self._sliced_trajectory):
            self._frame_index = i
            self._ts = ts
            self.frames[i] = ts.frame
            self.times[i] = ts.time
            self._single_frame()
        self._conclude()
        if verbose:
            logging.getLogger("mdhelper_amd").info("Analysis finished in %s.", datetime.now() - t0)
        return self


class SerialAnalysisBase(AnalysisBase):
    """Reference base.py:115-210 (``run`` :137-172, ``save`` :174-210)."""

    def __init__(self, trajectory, verbose: bool = False, **kwargs):
        super().__init__(trajectory, verbose, **kwargs)

    def run(self, start: int = None, stop: int = None, step: int = None,
            frames: Union[slice, np.ndarray] = None, n_jobs: int = 1, verbose: bool = None,
            **kwargs) -> "SerialAnalysisBase":
        return AnalysisBase.run(self, start, stop, step, frames, verbose, **kwargs)

    def save(self, file: Union[str, TextIO], archive: bool = True, compress: bool = True,
             **kwargs) -> None:
        data = {k: v for k, v in self.results.items() if k != "units"}
        if archive and compress:
            np.savez_compressed(file, **data, **kwargs)
        elif archive:
            np.savez(file, **data, **kwargs)
        else:
            for name, value in data.items():
                np.save(f"{file}_{name}", value, **kwargs)


class NumbaAnalysisBase(SerialAnalysisBase):
    """Reference base.py:212-279; ``n_threads`` has no meaning on the GPU and is ignored."""

    def run(self, start: int = None, stop: int = None, step: int = None,
            frames: Union[slice, np.ndarray] = None, n_threads: int = None,
            verbose: bool = None, **kwargs) -> "NumbaAnalysisBase":
        return AnalysisBase.run(self, start=start, stop=stop, step=step, frames=frames,
                                verbose=verbose, **kwargs)


class ParallelAnalysisBase(SerialAnalysisBase):
    """
    Reference base.py:281-507.  The reference fans frames out to worker
    processes (:396-501) and sums their per-frame results; here one process
    drives one GPU and the fan-out is the device's.  ``n_jobs``, ``module``,
    ``block`` and ``method`` are accepted for drop-in compatibility and ignored.
    """

    def _single_frame_parallel(self, frame: int, index: int) -> Any:
        raise NotImplementedError

    def run(self, start: int = None, stop: int = None, step: int = None,
            frames: Union[slice, np.ndarray] = None, verbose: bool = None, n_jobs: int = None,
            module: str = "multiprocessing", block: bool = True, method: str = None,
            **kwargs) -> "ParallelAnalysisBase":
        if method is not None and method not in {"fork", "forkserver", "spawn"}:
            raise ValueError("Invalid multiprocessing start method.")
        return AnalysisBase.run(self, start, stop, step, frames, verbose, **kwargs)


class DynamicAnalysisBase(ParallelAnalysisBase, SerialAnalysisBase):
    """Reference base.py:509-584: serial / parallel switch by the ``parallel`` flag."""

    def __init__(self, trajectory, parallel: bool, verbose: bool = False, **kwargs) -> None:
        self._parallel = parallel
        SerialAnalysisBase.__init__(self, trajectory, verbose=verbose, **kwargs)

    def run(self, start: int = None, stop: int = None, step: int = None,
            frames: Union[slice, np.ndarray] = None, verbose: bool = None,
            **kwargs) -> Union[SerialAnalysisBase, ParallelAnalysisBase]:
        return (ParallelAnalysisBase if self._parallel else SerialAnalysisBase).run(
            self, start=start, stop=stop, step=step, frames=frames, verbose=verbose, **kwargs)


class FrameBatcher:
    """
    Collects per-frame float32 position blocks (one per group) into contiguous
    batches and hands full batches to ``flush([positions[F, N_g, 3], ...], boxes[F, 6] | None)``.
    """

    def __init__(self, n_atoms, flush, with_box: bool = True, max_bytes: int = 256 << 20):
        sizes = [int(n) for n in np.atleast_1d(n_atoms)]
        self.capacity = int(max(1, min(4096, max_bytes // max(12 * sum(sizes), 1))))
        self._pos = [np.empty((self.capacity, n, 3), dtype=np.float32) for n in sizes]
        self._box = np.empty((self.capacity, 6), dtype=np.float32) if with_box else None
        self._n = 0
        self._flush = flush

    def add(self, positions, box=None):
        for buf, p in zip(self._pos, positions):
            buf[self._n] = p
        if self._box is not None:
            self._box[self._n] = box
        self._n += 1
        if self._n == self.capacity:
            self.flush()

    def flush(self):
        if self._n:
            self._flush([b[:self._n] for b in self._pos],
                        None if self._box is None else self._box[:self._n])
            self._n = 0
