"""Per-step wall and kernel times of the C4 MSD step over 40 synchronised steps (first argument: n_blocks).
Run through gpurun: python scripts/msd_step_times.py [n_blocks].  The first step after an idle period runs at a
lower clock; DESIGN.md quotes the mean of steps 5..39."""
import sys, time, os
sys.path.insert(0, ".")
import numpy as np
from mdhelper_amd import _core, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N, T = 10000, 100000
traj = _core.synth_random_walk(T, N, [1.0, 1.0, 1.0], 0.1, seed=4, dev=0, dtype=np.float64)
eng = _core.MsdEngine(T // B, B, 2, dev=0, timing=True)
ts, ks = [], []
for s in range(40):
    t0 = time.perf_counter()
    eng.reset()
    eng.push_device(0, traj.ptr, N, 0, N // 2)
    eng.push_device(1, traj.ptr, N, N // 2, N // 2)
    _lib.check(_lib.lib().mdx_device_synchronize(0))
    ts.append((time.perf_counter() - t0) * 1e3)
    ks.append(eng.stats()["kernel_ms"])
print("wall ms:", " ".join(f"{t:.1f}" for t in ts))
print("kern ms:", " ".join(f"{t:.1f}" for t in ks))
