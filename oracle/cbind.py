"""
oracle.cbind — ctypes binding of the C restatement (oracle/c/rdf_oracle.c).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).
"""
from __future__ import annotations

import ctypes
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_SO = _HERE / "_build" / "librdf_oracle.so"
_lib = None


def build(force: bool = False) -> pathlib.Path:
    src = _HERE / "c" / "rdf_oracle.c"
    if force or not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE / "c")], check=True, capture_output=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(str(build()))
        _lib.rdf_oracle_histogram.restype = ctypes.c_int
        _lib.rdf_oracle_histogram.argtypes = [
            ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double,
            ctypes.c_long, ctypes.c_long, ctypes.c_void_p, ctypes.c_int]
        _lib.rdf_oracle_max_threads.restype = ctypes.c_int
    return _lib


def max_threads() -> int:
    return int(lib().rdf_oracle_max_threads())


def c_radial_histogram(pos1, pos2, n_bins, range, dims, *, exclusion=None, n_threads=0,
                       counts=None):
    """Same contract as ``oracle.rdf.radial_histogram_ref`` (structure.py:32-104)."""
    p1 = np.ascontiguousarray(pos1, dtype=np.float32).reshape(-1, 3)
    p2 = np.ascontiguousarray(pos2, dtype=np.float32).reshape(-1, 3)
    box = None if dims is None else np.ascontiguousarray(dims, dtype=np.float32).reshape(6)
    edges = np.linspace(range[0], range[1], n_bins + 1)
    if counts is None:
        counts = np.zeros(n_bins, dtype=np.int64)
    e0, e1 = (0, 0) if exclusion is None else (int(exclusion[0]), int(exclusion[1]))
    rc = lib().rdf_oracle_histogram(
        p1.ctypes.data, p1.shape[0], p2.ctypes.data, p2.shape[0],
        None if box is None else box.ctypes.data, edges.ctypes.data, int(n_bins),
        float(range[0]), float(range[1]), e0, e1, counts.ctypes.data, int(n_threads))
    if rc != 0:
        raise RuntimeError(f"rdf_oracle_histogram failed ({rc})")
    return counts
