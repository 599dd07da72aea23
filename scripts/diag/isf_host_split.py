"""Where does the ISF host path lose against the resident one?  reset / accumulate / result timed apart."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mdhelper_amd import _core

N, L, F, n_lags = 32768, 68.94, 512, 64
grid = 2 * np.pi * np.arange(8) / L
q = np.stack(np.meshgrid(grid, grid, grid), -1).reshape(-1, 3)
d = _core.synth_random_walk(F, N, [L, L, L], 0.3, seed=2)
h = d.to_host()
eng = _core.IsfEngine(q, [N // 2, N - N // 2], ((0, 0), (0, 1), (1, 1)), n_lags, True)

def t(fn):
    t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); _core.synchronize(0); return (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3

for rep in range(4):
    a = t(eng.reset); b = t(lambda: eng.accumulate_device(d.ptr, N, F)); c = t(eng.result)
    print("resident: reset %.2f+%.2f  accumulate %.2f+%.2f  result %.2f+%.2f" % (a + b + c), flush=True)
for rep in range(4):
    a = t(eng.reset); b = t(lambda: eng.accumulate(h)); c = t(eng.result)
    print("host    : reset %.2f+%.2f  accumulate %.2f+%.2f  result %.2f+%.2f" % (a + b + c), flush=True)
