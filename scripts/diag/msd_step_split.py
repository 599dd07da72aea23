"""Host-side split of bench.py's MSD step (reset + pushes + result) for n_blocks = 1 and 8, timing on / off."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mdhelper_amd import _core

N, T = 10000, 100000
d = _core.synth_random_walk(T, N, [1, 1, 1], 0.1, seed=4, dtype=np.float64)
for B in (1, 8):
    for timing in (False, True):
        eng = _core.MsdEngine(T // B, B, 2, timing=timing)
        rows = []
        for rep in range(8):
            t = [time.perf_counter()]
            eng.reset(); t.append(time.perf_counter())
            eng.push_device(0, d.ptr, N, 0, N // 2); t.append(time.perf_counter())
            eng.push_device(1, d.ptr, N, N // 2, N // 2); t.append(time.perf_counter())
            eng.result(); t.append(time.perf_counter())
            rows.append(np.diff(t) * 1e3)
        rows = np.array(rows)
        print(f"B={B} timing={timing}: reset push0 push1 result (ms), last 5 steps mean:", np.round(rows[3:].mean(axis=0), 2),
              "total", round(rows[3:].sum(axis=1).mean(), 2), flush=True)
        eng.close()
