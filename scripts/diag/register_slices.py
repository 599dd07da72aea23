"""What explicit page-locking costs per slice: hipHostRegister / hipHostUnregister of slices of a pageable float32
[T][N][3] array, and the strided column copy of C4's 12 GB done that way — every slice registered, copied by a 2-D
DMA, unregistered, several slices in flight on threads of their own — against the runtime's implicit route.
    python scripts/diag/register_slices.py [T] [N]"""
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mdhelper_amd import _core, _lib  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
lib = _lib.lib()
hip = ctypes.CDLL(_lib.runtime()["libamdhip64"])
V, Z, I, U = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint
hip.hipHostRegister.argtypes = [V, Z, U]
hip.hipHostUnregister.argtypes = [V]
hip.hipMemcpy2DAsync.argtypes = [V, Z, V, Z, Z, Z, I, V]
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(V), U]
hip.hipStreamSynchronize.argtypes = [V]
hip.hipSetDevice.argtypes = [I]
d = _core.synth_random_walk(T, N, [50, 50, 50], 0.1, seed=4, wrap=True)
h = d.to_host()
d.free()
gb = h.nbytes / 1e9
res = {"T": T, "N": N, "GB": round(gb, 2)}
base = h.ctypes.data
PAGE = 4096

# (1) register / unregister cost by slice size
for mb in (32, 128, 512):
    n = mb << 20
    lo = (base + PAGE - 1) & ~(PAGE - 1)
    ts_r, ts_u = [], []
    for k in range(4):
        p = lo + k * n
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(p, n, 0)
        t1 = time.perf_counter()
        rc2 = hip.hipHostUnregister(p)
        t2 = time.perf_counter()
        assert rc == 0 and rc2 == 0, (rc, rc2)
        ts_r.append((t1 - t0) * 1e3)
        ts_u.append((t2 - t1) * 1e3)
    res[f"register_{mb}MB_ms"] = [round(x, 2) for x in ts_r]
    res[f"unregister_{mb}MB_ms"] = [round(x, 2) for x in ts_u]

# (2) the column copy: quarter chunks, slices of rows registered / copied / unregistered by n_thr threads
c = N // 4
buf = _core.DeviceArray((T, c, 3), np.float32)
row_bytes, stride = 12 * c, 12 * N


def column_chunk(first, n_thr, rows_per):
    n_slices = -(-T // rows_per)
    nxt = [0]
    lock = threading.Lock()
    err = []

    def worker():
        hip.hipSetDevice(0)
        s = V()
        hip.hipStreamCreateWithFlags(ctypes.byref(s), 1)
        while True:
            with lock:
                k = nxt[0]
                nxt[0] += 1
            if k >= n_slices:
                break
            r0 = k * rows_per
            nr = min(rows_per, T - r0)
            a = base + r0 * stride + 12 * first
            b = a + (nr - 1) * stride + row_bytes
            lo, hi = a & ~(PAGE - 1), (b + PAGE - 1) & ~(PAGE - 1)
            rc = hip.hipHostRegister(lo, hi - lo, 0)
            if rc:
                err.append(("reg", rc))
                break
            rc = hip.hipMemcpy2DAsync(buf.ptr.value + r0 * row_bytes, row_bytes, a, stride, row_bytes, nr, 1, s)
            rc = rc or hip.hipStreamSynchronize(s)
            rc2 = hip.hipHostUnregister(lo)
            if rc or rc2:
                err.append(("copy", rc, rc2))
                break

    ths = [threading.Thread(target=worker) for _ in range(n_thr)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    return err


for n_thr, rows_per in ((1, 4096), (2, 4096), (4, 4096), (4, 1024), (8, 1024), (8, 4096)):
    rates = []
    for rep in range(2):
        t0 = time.perf_counter()
        errs = []
        for first in range(0, N, c):
            errs += column_chunk(first, n_thr, rows_per)
        _core.synchronize(0)
        rates.append(round(gb / (time.perf_counter() - t0), 1))
    res[f"explicit_{n_thr}thr_{rows_per}rows"] = rates if not errs else str(errs[:2])
    # slices overlap by pages at their edges when rows_per * stride is not a page multiple: registration of a page
    # that a neighbour slice holds fails -> reported above
ok = np.array_equal(buf.to_host(), h[:, N - c:])
res["last_chunk_bytes_equal"] = bool(ok)
print(json.dumps(res), flush=True)
