"""Host -> HBM rate of mdx_upload_rows for column ranges of float32[T][N][3] (T = 100 000, N = 10 000): how wide must a
column chunk be for the strided host reads to keep the link busy?  Pageable (ring + copy threads) and page-locked."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mdhelper_amd import _core, _lib

T, N = 100000, 10000
d = _core.synth_random_walk(T, N, [1, 1, 1], 0.1, seed=4, wrap=False)
h = d.to_host()
d.free()
for pinned in (False, True):
    if pinned:
        _lib.check(_lib.lib().mdx_host_register(0, h.ctypes.data, h.nbytes))
    for c in (625, 1250, 2500, 5000, 10000):
        buf = _core.DeviceArray((T, c, 3), np.float32)
        buf.upload_columns(h, 0, c)
        t0 = time.perf_counter()
        n = 0
        for first in range(0, N, c):
            buf.upload_columns(h, first, c); n += 1
        dt = time.perf_counter() - t0
        print(f"{'page-locked' if pinned else 'pageable   '} columns of {c:5d} particles ({12 * c / 1024:6.1f} KB rows): {12e-9 * T * N / dt:6.1f} GB/s", flush=True)
        buf.free()
    if pinned:
        _lib.check(_lib.lib().mdx_host_unregister(0, h.ctypes.data))
