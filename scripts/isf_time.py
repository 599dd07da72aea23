"""ISF engine timing on one GPU (run through gpurun): 32 768 particles, 512 grid wavevectors,
two groups, frames fed from host memory in two calls; prints frames/s per configuration."""
import itertools
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from mdhelper_amd import _core

N, F, L = 32768, 96, 68.94
grid = 2 * np.pi * np.arange(8) / L
q = np.stack(np.meshgrid(grid, grid, grid), -1).reshape(-1, 3)
pairs = tuple(itertools.combinations_with_replacement(range(2), 2))
rng = np.random.default_rng(0)
pos = (rng.random((1, N, 3)) * L + np.cumsum(rng.normal(0, 0.3, (F, N, 3)), axis=0)).astype(np.float32)
for n_lags, inc in [(16, False), (16, True), (64, True)]:
    eng = _core.IsfEngine(q, [N // 2, N - N // 2], pairs, n_lags, inc)
    eng.accumulate(pos[:32])
    eng.result()
    t0 = time.perf_counter()
    eng.accumulate(pos[32:])
    eng.result()
    dt = time.perf_counter() - t0
    print(f"n_lags={n_lags} incoherent={inc}: {64 / dt:.1f} frames/s")
    eng.close()
