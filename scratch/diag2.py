import sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from mdhelper_amd import _core
from oracle import rdf as orf
rng = np.random.default_rng(12)
F, N, L = 24, 3000, 31.0
pos = rng.uniform(0, L, (1, N, 3)) + np.cumsum(rng.normal(0, 0.3, (F, N, 3)), axis=0)
pos = np.mod(pos, L).astype(np.float32)
lengths = np.array([[L + 0.01 * f, L, L - 0.02 * f] for f in range(F)], dtype=np.float32)
dims = np.hstack([lengths, np.full((F, 3), 90.0, dtype=np.float32)])
edges = np.linspace(0, 12, 151)
ref = orf.rdf_run_ref(pos, dims, 150, (0.0, 12.0), exclusion=(1, 1))["counts"]
for algo in ("cell", "filter"):
    bad = []
    for it in range(20):
        e = _core.RdfEngine(edges, (1, 1), algo=algo)
        e.accumulate(pos, None, dims)
        c = e.counts(); e.close()
        nz = np.nonzero(c - ref)[0]
        if len(nz):
            bad.append((it, nz.tolist(), (c - ref)[nz].tolist()))
    print("batched", algo, bad)
refs = [orf.rdf_run_ref(pos[f:f+1], dims[f:f+1], 150, (0.0, 12.0), exclusion=(1, 1))["counts"] for f in range(F)]
bad = []
for it in range(10):
    for f in range(F):
        e = _core.RdfEngine(edges, (1, 1), algo="cell")
        e.accumulate(pos[f:f+1], None, dims[f:f+1])
        c = e.counts(); e.close()
        nz = np.nonzero(c - refs[f])[0]
        if len(nz):
            bad.append((it, f, nz.tolist(), (c - refs[f])[nz].tolist()))
print("per-frame cell", bad)
# same engine reused
e = _core.RdfEngine(edges, (1, 1), algo="cell")
bad = []
for it in range(20):
    e.reset()
    e.accumulate(pos, None, dims)
    c = e.counts()
    nz = np.nonzero(c - ref)[0]
    if len(nz):
        bad.append((it, nz.tolist(), (c - ref)[nz].tolist()))
print("same engine", bad)
