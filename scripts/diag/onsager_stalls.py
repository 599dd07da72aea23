"""Where do the seconds go when Onsager(...).run() is called again and again at C4 size?  Times every
engine call (with a device synchronise behind it) over a few analyses on HBM-resident float64 / float32 frames."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import mdhelper_amd
from mdhelper_amd import _core
from mdhelper_amd.analysis import Onsager

N, T = 10000, 100000
log = []

def wrap(cls, name):
    fn = getattr(cls, name)
    def timed(self, *a, **k):
        t0 = time.perf_counter()
        out = fn(self, *a, **k)
        t1 = time.perf_counter()
        _core.synchronize(0)
        log.append((name, (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3))
        return out
    setattr(cls, name, timed)

for name in ("__init__", "push_device", "push_frames_device", "result", "close"):
    wrap(_core.MsdEngine, name)

dims = np.array([50, 50, 50, 90, 90, 90], dtype=np.float32)
for dtype in (np.float64, np.float32):
    d = (_core.synth_random_walk(T, N, [1, 1, 1], 0.1, seed=4, dtype=np.float64) if dtype == np.float64
         else _core.synth_random_walk(T, N, [1, 1, 1], 0.1, seed=4, wrap=False))
    u = mdhelper_amd.ArrayUniverse.from_device(d, dims)
    for rep in range(6):
        log.clear()
        t0 = time.perf_counter()
        Onsager((u.atoms[:N // 2], u.atoms[N // 2:]), temperature=1, reduced=True, verbose=False).run()
        total = (time.perf_counter() - t0) * 1e3
        print(np.dtype(dtype).name, rep, f"total {total:8.1f} ms ", " ".join(f"{n}:{a:.1f}+{b:.1f}" for n, a, b in log), flush=True)
    d.free()

# the engine alone: result() after the pushes have finished
d = _core.synth_random_walk(T, N, [1, 1, 1], 0.1, seed=4, dtype=np.float64)
for B in (1, 8):
    eng = _core.MsdEngine(T // B, B, 2)
    for rep in range(4):
        eng.reset()
        log.clear()
        eng.push_device(0, d.ptr, N, 0, N // 2)
        eng.push_device(1, d.ptr, N, N // 2, N // 2)
        eng.result()
        print("engine B", B, rep, " ".join(f"{n}:{a:.1f}+{b:.1f}" for n, a, b in log), flush=True)
    eng.close()
d.free()
