"""Does the memory-side cache carry Y when the whole chip works on a few super groups at a time?  C4 pushed in chunks of `c`
atoms (one super group = 42.67 atoms = 210 MB of Y).  Needs a build whose pass A may split a super group into up to R2 column
ranges (the product caps it at 64): fsplit cap `sh.r2` in msdfft::launch."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mdhelper_amd import _core

T, N = 100000, 10000
traj = _core.synth_random_walk(T, N, [1.0, 1.0, 1.0], 0.1, seed=4, dtype=np.float64)
eng = _core.MsdEngine(T, 1, 2)
ref = None
for c in (5000, 1280, 640, 320, 128, 64):
    def run():
        eng.reset()
        for g, first in ((0, 0), (1, N // 2)):
            for a in range(0, N // 2, c):
                eng.push_device(g, traj.ptr, N, first + a, min(c, N // 2 - a))
        eng.synchronize()
    for _ in range(4):
        run()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); run(); ts.append((time.perf_counter() - t0) * 1e3)
    msd = eng.result()[0]
    if ref is None:
        ref = msd
    print(f"chunks of {c:5d} atoms ({2 * ((N // 2 + c - 1) // c):4d} pushes): median {np.median(ts):7.2f} ms  min {min(ts):7.2f}   same MSD {np.allclose(msd, ref, rtol=1e-10)}", flush=True)
