"""What the host link gives for the 12 GB of BASELINE C4 (float32[100 000][10 000][3], pageable), by route:
the runtime's own pageable hipMemcpy / hipMemcpy2D (pins the caller's pages on the fly), the library's pinned ring
(mdx_upload whole, mdx_upload_rows column chunks), and the NetCDF file route (mdx_traj_load_device).
    python scripts/diag/host_feed_rates.py [io_threads ...]     (one child process per thread count)"""
import ctypes
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] != "child":
    for th in sys.argv[1:]:
        out = subprocess.run([sys.executable, __file__, "child"], env={**os.environ, "MDX_IO_THREADS": th},
                             capture_output=True, text=True)
        print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-2000:], flush=True)
    sys.exit(0)

import bench  # noqa: E402
from mdhelper_amd import _core, _lib  # noqa: E402
from mdhelper_amd.io import TrajectoryFile  # noqa: E402

T, N = 100000, 10000
lib = _lib.lib()
hip = ctypes.CDLL(_lib.runtime()["libamdhip64"])
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
hip.hipMemcpy2D.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                            ctypes.c_size_t, ctypes.c_int]
d = _core.synth_random_walk(T, N, [1, 1, 1], 0.1, seed=4, wrap=False)
h = d.to_host()
gb = h.nbytes / 1e9
res = {"io_threads": int(os.environ.get("MDX_IO_THREADS", "8")), "GB": gb}


def rate(fn, reps=2):
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        _core.synchronize(0)
        best = min(best, time.perf_counter() - t0)
    return round(gb / best, 1)


res["runtime_hipMemcpy_whole"] = rate(lambda: hip.hipMemcpy(d.ptr, h.ctypes.data, h.nbytes, 1))
res["ring_upload_whole"] = rate(lambda: _lib.check(lib.mdx_upload(0, d.ptr, h.ctypes.data, h.nbytes)))
c = 1250
buf = _core.DeviceArray((T, c, 3), np.float32)


def ring_rows():
    for first in range(0, N, c):
        buf.upload_columns(h, first, c)


def runtime_2d():
    for first in range(0, N, c):
        hip.hipMemcpy2D(buf.ptr, 12 * c, h.ctypes.data + 12 * first, 12 * N, 12 * c, T, 1)


res["ring_upload_rows_1250"] = rate(ring_rows)
res["runtime_hipMemcpy2D_1250"] = rate(runtime_2d, reps=1)
buf.free()
tmp = tempfile.NamedTemporaryFile(suffix=".nc", delete=False)
tmp.close()
try:
    bench.write_amber_netcdf_fast(tmp.name, h, np.array([50, 50, 50, 90, 90, 90], dtype=np.float32))
    tf = TrajectoryFile(tmp.name)
    frames = np.arange(T)
    res["file_load_device"] = rate(lambda: tf.load_device(frames, d.ptr, dev=0))
    tf.close()
finally:
    os.unlink(tmp.name)
d.free()
print(json.dumps(res), flush=True)
