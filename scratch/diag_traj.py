import sys, tempfile, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import mdhelper_amd
from mdhelper_amd.analysis import RadialDistributionFunction
from oracle import rdf as orf
from trajfiles import write_amber_netcdf
rng = np.random.default_rng(12)
F, N, L = 24, 3000, 31.0
pos = rng.uniform(0, L, (1, N, 3)) + np.cumsum(rng.normal(0, 0.3, (F, N, 3)), axis=0)
pos = np.mod(pos, L).astype(np.float32)
lengths = np.array([[L + 0.01 * f, L, L - 0.02 * f] for f in range(F)], dtype=np.float32)
dims = np.hstack([lengths, np.full((F, 3), 90.0, dtype=np.float32)])
d = tempfile.mkdtemp()
write_amber_netcdf(os.path.join(d, "r.nc"), pos, lengths)
uf = mdhelper_amd.FileUniverse(os.path.join(d, "r.nc"))
um = mdhelper_amd.ArrayUniverse(pos, dims)
kw = dict(n_bins=150, range=(0.0, 12.0), exclusion=(1, 1))
a = RadialDistributionFunction(uf.atoms, **kw).run().results.counts
b = RadialDistributionFunction(um.atoms, **kw).run().results.counts
ref = orf.rdf_run_ref(pos, dims, 150, (0.0, 12.0), exclusion=(1, 1))["counts"]
print("a-ref", np.nonzero(a - ref)[0], (a - ref)[np.nonzero(a - ref)[0]])
print("b-ref", np.nonzero(b - ref)[0], (b - ref)[np.nonzero(b - ref)[0]])
from mdhelper_amd import _core
edges = np.linspace(0, 12, 151)
for algo in ("exact", "filter", "cell"):
    per = []
    for f in range(F):
        e = _core.RdfEngine(edges, (1, 1), algo=algo)
        e.accumulate(pos[f:f+1], None, dims[f:f+1])
        c = e.counts(); e.close()
        r = orf.rdf_run_ref(pos[f:f+1], dims[f:f+1], 150, (0.0, 12.0), exclusion=(1, 1))["counts"]
        if not np.array_equal(c, r):
            nz = np.nonzero(c - r)[0]
            per.append((f, nz.tolist(), (c - r)[nz].tolist()))
    print(algo, per)
