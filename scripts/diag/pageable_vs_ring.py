"""Rate of the runtime's own pageable hipMemcpy against the library's pinned ring, both directions, 1 GiB
(the decision of NOTES round 5 about commit 0d788a6: the ring stays as a performance choice or goes)."""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mdhelper_amd import _core, _lib  # noqa: E402

lib = _lib.lib()
rt = _lib.runtime()
hip = ctypes.CDLL(rt["libamdhip64"])          # the copy libmdx.so is bound to (already mapped)
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
n = 1 << 30
h = np.random.default_rng(0).integers(0, 255, n, dtype=np.uint8)
back = np.zeros_like(h)
d = _core.DeviceArray((n,), np.uint8)
out = {"bytes": n, "libamdhip64": rt["libamdhip64"]}


def best(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return n / min(ts) / 1e9


out["h2d_runtime_pageable_GBps"] = best(lambda: hip.hipMemcpy(d.ptr, h.ctypes.data, n, 1))
out["h2d_ring_GBps"] = best(lambda: _lib.check(lib.mdx_upload(0, d.ptr, h.ctypes.data, n)))
out["d2h_runtime_pageable_GBps"] = best(lambda: hip.hipMemcpy(back.ctypes.data, d.ptr, n, 2))
assert np.array_equal(back, h)
back[:] = 0
out["d2h_ring_GBps"] = best(lambda: _lib.check(lib.mdx_memcpy_d2h(0, back.ctypes.data, d.ptr, n)))
assert np.array_equal(back, h)
d.free()
print(json.dumps(out))
