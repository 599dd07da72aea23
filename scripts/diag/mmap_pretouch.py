"""First-touch cost of a mapped trajectory file as a DMA source (scripts/diag/mmap_feed_rates.py: 33 GB/s the first
time, 57.6 GB/s the second): is it the population of the mapping's page tables?  Fresh mappings, (a) touched by
eight threads first (one byte per page), (b) madvise(MADV_POPULATE_READ), then the first hipMemcpy2D out of them.
    python scripts/diag/mmap_pretouch.py [T] [N]"""
import ctypes
import json
import mmap
import os
import sys
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mdhelper_amd import _core, _lib  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
lib = _lib.lib()
hip = ctypes.CDLL(_lib.runtime()["libamdhip64"])
V, Z, I = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
hip.hipMemcpy2D.argtypes = [V, Z, V, Z, Z, Z, I]
d = _core.synth_random_walk(T, N, [50, 50, 50], 0.1, seed=4, wrap=True)
h = d.to_host()
gb = h.nbytes / 1e9
res = {"T": T, "N": N, "GB": round(gb, 2), "cpus": os.cpu_count()}
tmp = tempfile.NamedTemporaryFile(suffix=".nc", delete=False)
tmp.close()
try:
    bench.write_amber_netcdf_fast(tmp.name, h, np.array([50, 50, 50, 90, 90, 90], dtype=np.float32))
    size = os.path.getsize(tmp.name)
    pitch = 12 * N + 52
    first = size - T * pitch + 4
    fd = os.open(tmp.name, os.O_RDONLY)

    def copy_rate(base):
        t0 = time.perf_counter()
        rc = hip.hipMemcpy2D(d.ptr, 12 * N, base, pitch, 12 * N, T, 1)
        _core.synchronize(0)
        return round(gb / (time.perf_counter() - t0), 1) if rc == 0 else f"hip error {rc}"

    # (0) untouched
    mm = mmap.mmap(fd, size, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ)
    arr = np.frombuffer(mm, dtype=np.uint8)
    res["untouched_first_copy"] = copy_rate(arr.ctypes.data + first)
    del arr
    mm.close()
    # (a) touched by threads
    for n_thr in (8, 16):
        mm = mmap.mmap(fd, size, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ)
        arr = np.frombuffer(mm, dtype=np.uint8)
        per = -(-size // n_thr // 4096) * 4096

        def touch(k):
            arr[k * per:min(size, (k + 1) * per):4096].max()

        t0 = time.perf_counter()
        ths = [threading.Thread(target=touch, args=(k,)) for k in range(n_thr)]
        [t.start() for t in ths]
        [t.join() for t in ths]
        res[f"touch_{n_thr}_threads_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        res[f"touched_{n_thr}_first_copy"] = copy_rate(arr.ctypes.data + first)
        res[f"touched_{n_thr}_second_copy"] = copy_rate(arr.ctypes.data + first)
        del arr
        mm.close()
    # (b) MADV_POPULATE_READ (Linux >= 5.14)
    mm = mmap.mmap(fd, size, flags=mmap.MAP_SHARED, prot=mmap.PROT_READ)
    arr = np.frombuffer(mm, dtype=np.uint8)
    t0 = time.perf_counter()
    try:
        mm.madvise(22)
        res["madv_populate_read_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
    except OSError as e:
        res["madv_populate_read_ms"] = f"refused: {e}"
    res["populated_first_copy"] = copy_rate(arr.ctypes.data + first)
    del arr
    mm.close()
    os.close(fd)
finally:
    os.unlink(tmp.name)
print(json.dumps(res), flush=True)
