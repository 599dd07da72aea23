"""S(q) rate by wavevector-set shape (diagnostic): full grids take the regular quad form, q_max-filtered
(spherical) subsets the general quad form or the column kernel.  python scripts/run/diag_sq_forms.py"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from mdhelper_amd import _core

N, L = 32768, 68.94
sizes = [N // 2, N - N // 2]
pairs = ((0, 0), (0, 1), (1, 1))


def grid(n):
    g = 2 * np.pi * np.arange(n) / L
    return np.stack(np.meshgrid(g, g, g), -1).reshape(-1, 3)


def rate(q, frames):
    traj = _core.synth_random_walk(frames, N, [L, L, L], 0.3, seed=5, dev=0)
    eng = _core.SqEngine(q, sizes, pairs, dev=0, timing=True)
    eng.accumulate_device(traj.offset(0), N, frames)
    eng.result()
    eng.reset()
    t0 = time.perf_counter()
    eng.accumulate_device(traj.offset(0), N, frames)
    eng.result()
    dt = time.perf_counter() - t0
    st = eng.stats()
    eng.close()
    return frames * float(N) * len(q) / dt, st


for name, q, frames in (
        ("full 8^3", grid(8), 2000),
        ("full 16^3", grid(16), 400),
        ("full 32^3", grid(32), 80),
        ("full 10^3", grid(10), 1000),
        ("full 20^3", grid(20), 200),
        ("32^3, |q| <= 0.5 q_axis_max (sphere octant)", None, 600),
        ("32^3, |q| <= 1.0 q_axis_max (sphere octant)", None, 120),
        ("16^3, |q| <= 1.0 q_axis_max", None, 600)):
    if q is None:
        n = 32 if name.startswith("32") else 16
        g = grid(n)
        cut = (0.5 if "0.5" in name else 1.0) * 2 * np.pi * (n - 1) / L
        q = g[np.linalg.norm(g, axis=1) <= cut]
    r, st = rate(q, frames)
    print(f"{name:48s} {len(q):6d} wavevectors  {r / 1e12:6.2f} T terms/s  {st}", flush=True)
