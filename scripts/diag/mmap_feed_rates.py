"""Can the DMA engine read a trajectory file where the page cache holds it?  A NetCDF file (float32[T][N][3] records,
12 N + 52 bytes apart) is mapped read-only and handed to the runtime's pageable copies (hipMemcpy2D pins the pages it
is given on the fly): whole frames, column chunks, against the pread -> pinned ring -> DMA route (mdx_traj_load_device)
and against the same copies out of an anonymous (malloc'd) array.  Also: hipMemcpy2DAsync on a stream of its own.
    python scripts/diag/mmap_feed_rates.py [T] [N]"""
import ctypes
import json
import mmap
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mdhelper_amd import _core, _lib  # noqa: E402
from mdhelper_amd.io import TrajectoryFile  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
lib = _lib.lib()
hip = ctypes.CDLL(_lib.runtime()["libamdhip64"])
V, Z, I = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
hip.hipMemcpy2D.argtypes = [V, Z, V, Z, Z, Z, I]
hip.hipMemcpy2DAsync.argtypes = [V, Z, V, Z, Z, Z, I, V]
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(V), ctypes.c_uint]
hip.hipStreamSynchronize.argtypes = [V]
d = _core.synth_random_walk(T, N, [50, 50, 50], 0.1, seed=4, wrap=True)
h = d.to_host()
gb = h.nbytes / 1e9
res = {"T": T, "N": N, "GB": round(gb, 2)}


def rate(fn):
    t0 = time.perf_counter()
    rc = fn()
    _core.synchronize(0)
    dt = time.perf_counter() - t0
    return round(gb / dt, 1) if not rc else f"hip error {rc}"


c = N // 4
buf = _core.DeviceArray((T, c, 3), np.float32)


def columns(base, pitch):
    rc = 0
    for first in range(0, N, c):
        rc = rc or hip.hipMemcpy2D(buf.ptr, 12 * c, base + 12 * first, pitch, 12 * c, T, 1)
    return rc


res["anon_whole_2D"] = [rate(lambda: hip.hipMemcpy2D(d.ptr, 12 * N, h.ctypes.data, 12 * N, 12 * N, T, 1)) for _ in range(2)]
res["anon_columns_quarter"] = [rate(lambda: columns(h.ctypes.data, 12 * N)) for _ in range(2)]
s = V()
assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0


def async_cols():
    rc = 0
    for first in range(0, N, c):
        rc = rc or hip.hipMemcpy2DAsync(buf.ptr, 12 * c, h.ctypes.data + 12 * first, 12 * N, 12 * c, T, 1, s)
    return rc or hip.hipStreamSynchronize(s)


res["anon_columns_quarter_async_stream"] = [rate(async_cols) for _ in range(2)]

tmp = tempfile.NamedTemporaryFile(suffix=".nc", delete=False)
tmp.close()
try:
    bench.write_amber_netcdf_fast(tmp.name, h, np.array([50, 50, 50, 90, 90, 90], dtype=np.float32))
    size = os.path.getsize(tmp.name)
    pitch = 12 * N + 52
    first = size - T * pitch + 4                      # coordinates of record 0
    tf = TrajectoryFile(tmp.name)
    frames = np.arange(T)
    res["file_pread_ring_load_device"] = [rate(lambda: tf.load_device(frames, d.ptr, dev=0)) for _ in range(2)]
    want = d.to_host()
    fd = os.open(tmp.name, os.O_RDONLY)
    for name, flags in (("shared", mmap.MAP_SHARED), ("private", mmap.MAP_PRIVATE)):
        mm = mmap.mmap(fd, size, flags=flags, prot=mmap.PROT_READ)
        arr = np.frombuffer(mm, dtype=np.uint8)
        base = arr.ctypes.data + first
        res[f"mmap_{name}_whole_2D"] = [rate(lambda: hip.hipMemcpy2D(d.ptr, 12 * N, base, pitch, 12 * N, T, 1))
                                        for _ in range(2)]
        got = d.to_host().view(">f4").astype(np.float32)
        res[f"mmap_{name}_bytes_equal"] = bool(np.array_equal(got, want))
        res[f"mmap_{name}_columns_quarter"] = [rate(lambda: columns(base, pitch)) for _ in range(2)]
        del arr
        mm.close()
    os.close(fd)
    tf.close()
finally:
    os.unlink(tmp.name)
print(json.dumps(res), flush=True)
