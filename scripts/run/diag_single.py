"""Latency of the function-level drop-in (one frame per call): radial_histogram at C2 size and at 4 000 atoms."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import numpy as np
from mdhelper_amd import _core
from mdhelper_amd.analysis import structure
for N, L in ((32768, 68.94), (4000, 34.2), (131072, 109.4)):
    rng = np.random.default_rng(0)
    pos = (rng.random((N, 3)) * L).astype(np.float32)
    dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    for _ in range(3):
        structure.radial_histogram(pos, pos, 201, (0.0, 15.0), dims, exclusion=(1, 1))
    _core.synchronize(0)
    t0 = time.perf_counter(); n = 20
    for _ in range(n):
        c = structure.radial_histogram(pos, pos, 201, (0.0, 15.0), dims, exclusion=(1, 1))
    dt = (time.perf_counter() - t0) / n
    print(f"N={N}: {dt*1e3:.3f} ms per call, {1/dt:.0f} frames/s, binned {int(c.sum())}")
    # the same through an engine kept alive (create/destroy excluded)
    eng = _core.RdfEngine(np.linspace(0, 15, 202), (1, 1))
    eng.accumulate(pos, None, dims); eng.counts()
    t0 = time.perf_counter()
    for _ in range(n):
        eng.accumulate(pos, None, dims)
    eng.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"        engine kept: {dt*1e3:.3f} ms per 1-frame accumulate")
    eng.close()
