"""
oracle.cpu_bench — timing harness of the CPU restatements (``bench.py``'s ``cpu_baseline`` leg).

TEST / MEASUREMENT INFRASTRUCTURE (see ``oracle/__init__.py``): never imported by the product.

SURVEY.md §8(d) asks for the build's NumPy restatement of the reference path — pair list +
``numpy.histogram`` (reference src/mdhelper/analysis/structure.py:92-104) — timed (1) on one core,
which is what the reference's serial ``run()`` uses, and (2) over ``len(os.sched_getaffinity(0))``
worker processes, which is what ``parallel=True`` does (reference analysis/base.py:385-386; the
reference hands whole frames to its workers, here each worker takes a block of rows of the same frame
so that a bounded sample keeps every core busy for the same time).
"""

from __future__ import annotations

import os
import time

import numpy as np


def rdf_rows(frame, dims, lo, hi, n_bins, rng, exclusion):
    """Counts of rows [lo, hi) of one frame against all its particles, NumPy restatement."""
    from oracle.rdf import radial_histogram_ref
    return radial_histogram_ref(frame[lo:hi], frame, n_bins, rng, dims, exclusion=exclusion,
                                i_offset=lo)


def time_rdf_numpy(frame, dims, n_bins, rng, exclusion, rows: int, workers: int = 1):
    """
    Time the NumPy restatement on ``rows`` rows per worker of one frame.

    Returns ``(counts, pairs_covered, seconds)``: the summed counts of the sampled rows, the
    ordered pairs they cover (rows x N) and the time — for ``workers > 1`` the longest compute
    time of the concurrently running worker processes (interpreter start-up excluded).
    """
    n = frame.shape[0]
    rows = max(1, min(rows, n // max(workers, 1)))
    if workers <= 1:
        t0 = time.perf_counter()
        counts = rdf_rows(frame, dims, 0, rows, n_bins, rng, exclusion)
        return counts, rows * n, time.perf_counter() - t0
    import json
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1",
               PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "task.npz")
        np.savez(path, frame=frame, dims=dims, rng=np.asarray(rng, dtype=np.float64),
                 exclusion=np.asarray(exclusion if exclusion else (0, 0)), n_bins=n_bins)
        # fresh interpreters (the caller may hold GPU state): exact pids, joined below
        procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_bench", path, str(w * rows),
                                   str((w + 1) * rows)], stdout=subprocess.PIPE, env=env, cwd=root,
                                  text=True) for w in range(workers)]
        outs = [p.communicate()[0] for p in procs]
    if any(p.returncode != 0 for p in procs):
        raise RuntimeError("a NumPy baseline worker failed")
    parts = [json.loads(o.strip().splitlines()[-1]) for o in outs]
    counts = np.sum([np.asarray(p["counts"], dtype=np.int64) for p in parts], axis=0)
    return counts, workers * rows * n, max(p["seconds"] for p in parts)


def time_rdf_kdtree(frame, dims, n_bins, rng, exclusion, rows: int):
    """
    The neighbour-search route an installed reference takes at this size: ``capped_distance`` switches from
    brute force to a periodic k-d tree / cell grid for large inputs (MDAnalysis.lib.distances, methods
    'pkdtree' / 'nsgrid'; reference src/mdhelper/analysis/structure.py:93-96 passes ``max_cutoff`` = the
    range end).  Neither MDAnalysis nor its grid is here; scipy's ``cKDTree(boxsize=...)`` is the same
    algorithm family on one core: pairs within the range end of ``rows`` query rows against the tree of all
    particles, then ``numpy.histogram``.  float64 distances from float32-wrapped coordinates: counts can
    differ from the contract's by a pair on a bin edge, so equality with the C restatement is reported, not
    required.  Returns (counts, ordered pairs covered, seconds including the tree build).
    """
    from scipy.spatial import cKDTree
    n = frame.shape[0]
    rows = max(1, min(rows, n))
    L = np.asarray(dims[:3], dtype=np.float64)
    t0 = time.perf_counter()
    x = np.mod(frame.astype(np.float64), L)
    x[x >= L] = 0.0                              # mod can return L itself for tiny negative inputs
    tree = cKDTree(x, boxsize=L)
    sub = cKDTree(x[:rows], boxsize=L)
    sd = sub.sparse_distance_matrix(tree, float(rng[1]), output_type="ndarray")
    keep = np.ones(sd.shape[0], dtype=bool)
    if exclusion:
        keep = (sd["i"] // exclusion[0]) != (sd["j"] // exclusion[1])
    counts, _ = np.histogram(sd["v"][keep], bins=n_bins, range=rng)
    return counts.astype(np.int64), rows * n, time.perf_counter() - t0


if __name__ == "__main__":
    import json
    import sys
    task = np.load(sys.argv[1])
    lo, hi = int(sys.argv[2]), int(sys.argv[3])
    excl = tuple(int(x) for x in task["exclusion"])
    t0 = time.perf_counter()
    c = rdf_rows(task["frame"], task["dims"], lo, hi, int(task["n_bins"]),
                 tuple(float(x) for x in task["rng"]), excl if excl[0] else None)
    print(json.dumps({"counts": c.tolist(), "seconds": time.perf_counter() - t0}))
