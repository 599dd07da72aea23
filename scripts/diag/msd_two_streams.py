"""Would pushing the two groups of C4 on two streams (tails of one group's launches filled by the other's) gain anything?
Two engines (own stream, own buffers), one group each, against one engine with both groups."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mdhelper_amd import _core

T, N = 100000, 10000
traj = _core.synth_random_walk(T, N, [1.0, 1.0, 1.0], 0.1, seed=4, dtype=np.float64)
one = _core.MsdEngine(T, 1, 2)
two = [_core.MsdEngine(T, 1, 1), _core.MsdEngine(T, 1, 1)]


def run_one():
    one.reset()
    one.push_device(0, traj.ptr, N, 0, N // 2)
    one.push_device(1, traj.ptr, N, N // 2, N // 2)
    one.synchronize()


def run_two():
    for e in two:
        e.reset()
    two[0].push_device(0, traj.ptr, N, 0, N // 2)
    two[1].push_device(0, traj.ptr, N, N // 2, N // 2)
    for e in two:
        e.synchronize()


for name, fn in (("one engine, one stream ", run_one), ("two engines, two streams", run_two), ("one engine, one stream ", run_one),
                 ("two engines, two streams", run_two)):
    for _ in range(6):
        fn()
    _core.synchronize(0)
    ts = []
    for _ in range(12):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{name}: median {np.median(ts):.2f} ms  min {min(ts):.2f}  max {max(ts):.2f}", flush=True)
a = one.result()[0]
b0, b1 = two[0].result()[0], two[1].result()[0]
print("same MSD:", np.allclose(a[0], b0[0], rtol=1e-12), np.allclose(a[1], b1[0], rtol=1e-12))
