import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from mdhelper_amd import _core
from mdhelper_amd._lib import lib, check
from oracle import rdf as orf
rng = np.random.default_rng(12)
F, N, L = 24, 3000, 31.0
pos = rng.uniform(0, L, (1, N, 3)) + np.cumsum(rng.normal(0, 0.3, (F, N, 3)), axis=0)
pos = np.mod(pos, L).astype(np.float32)
f = 17
box = np.array([[L + 0.01 * f, L, L - 0.02 * f, 90, 90, 90]], dtype=np.float32)
edges = np.linspace(0, 12, 151)
ref = orf.rdf_run_ref(pos[f:f+1], box, 150, (0.0, 12.0), exclusion=(1, 1))["counts"]
n_pad = 3072
saved = 0
for it in range(12):
    e = _core.RdfEngine(edges, (1, 1), algo="cell")
    e.accumulate(pos[f:f+1], None, box)
    c = e.counts()
    pw = np.zeros((n_pad, 4), np.float32); po = np.zeros((n_pad, 4), np.float32)
    check(lib().mdx_rdf_debug_sorted(e.handle, 0, n_pad, pw.ctypes.data, po.ctypes.data))
    bad = not np.array_equal(c, ref)
    print(it, "bad" if bad else "ok", (c - ref)[:2])
    if bad and saved < 2:
        np.savez(f"gpurun_out/sorted_bad{saved}.npz", pw=pw, po=po, box=box); saved += 1
    if not bad:
        np.savez("gpurun_out/sorted_ok.npz", pw=pw, po=po, box=box)
    e.close()
