"""cProfile of Onsager(...).run() on HBM-resident float64 frames at C4 size: where the milliseconds beside the engine go.
    python scripts/diag/onsager_profile.py"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import mdhelper_amd  # noqa: E402
from mdhelper_amd import _core  # noqa: E402
from mdhelper_amd.analysis import Onsager  # noqa: E402

N, T, L = 10000, 100000, 50.0
dims = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
charges = np.r_[np.ones(N // 2), -np.ones(N - N // 2)]
d64 = _core.synth_random_walk(T, N, [1.0, 1.0, 1.0], 0.1, seed=4, dev=0, dtype=np.float64)
u = mdhelper_amd.ArrayUniverse.from_device(d64, dims, charges=charges)


def analysis():
    return Onsager((u.atoms[:N // 2], u.atoms[N // 2:]), temperature=1, reduced=True, n_blocks=1, verbose=False).run()


for _ in range(3):
    analysis()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    analysis()
    ts.append((time.perf_counter() - t0) * 1e3)
print("ms per analysis", [round(x, 2) for x in ts])
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    analysis()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
st.sort_stats("tottime").print_stats(18)
