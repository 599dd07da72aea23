"""How long do hipMalloc / first touch / hipFree of multi-GB blocks take on this box?  (Why an analysis
object per call — the reference's usage — must not return its large buffers to the driver every time.)"""
import sys, time
from ctypes import byref, c_void_p
sys.path.insert(0, ".")
from mdhelper_amd import _core
from mdhelper_amd._lib import check, lib

def t(fn):
    t0 = time.perf_counter(); fn(); _core.synchronize(0); return (time.perf_counter() - t0) * 1e3

for gb in (1, 12, 24):
    n = gb << 30
    for rep in range(4):
        p = c_void_p()
        a = t(lambda: check(lib().mdx_malloc(0, n, byref(p))))
        b = t(lambda: check(lib().mdx_memset(0, p, 0, n)))
        c = t(lambda: check(lib().mdx_memset(0, p, 0, n)))
        d = t(lambda: check(lib().mdx_free(0, p)))
        print(f"{gb:3d} GB rep {rep}: malloc {a:8.1f} ms  first memset {b:8.1f} ms  second memset {c:7.1f} ms  free {d:8.1f} ms", flush=True)
