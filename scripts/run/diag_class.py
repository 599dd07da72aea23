"""Where the operator-surface path spends its time (run on the GPU box): phases of
RadialDistributionFunction(...).run() on an in-memory C2 universe."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import mdhelper_amd
from mdhelper_amd import _core
from mdhelper_amd.analysis import RadialDistributionFunction
from mdhelper_amd.analysis import structure

N, F, L = 32768, 3000, 68.94
box = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
d = _core.synth_random_walk(F, N, box[:3], 0.3, seed=2)
h = d.to_host(); d.free()
u = mdhelper_amd.ArrayUniverse(h, box)
edges = np.linspace(0, 15, 202)

def stamp(label, t0):
    _core.synchronize(0)
    print(f"  {label:28s} {1e3*(time.perf_counter()-t0):8.2f} ms")
    return time.perf_counter()

for rep in range(3):
    print("rep", rep)
    t0 = time.perf_counter()
    r = RadialDistributionFunction(u.atoms, exclusion=(1, 1), verbose=False)
    t = stamp("constructor", t0)
    r._setup_frames(r._trajectory); t = stamp("_setup_frames", t)
    r._prepare(); t = stamp("_prepare (engine create)", t)
    eng = r._engine
    hb = np.tile(box, (F, 1))
    t = time.perf_counter()
    eng.accumulate(h, None, hb); eng.synchronize(); t = stamp("accumulate all frames", t)
    c = eng.counts(); t = stamp("counts", t)
    eng.close(); t = stamp("close", t)
    t = time.perf_counter()
    RadialDistributionFunction(u.atoms, exclusion=(1, 1), verbose=False).run(); t = stamp("whole .run()", t)
    # blocks of the class path
    e2 = _core.RdfEngine(edges, (1, 1)); t = stamp("engine create", t)
    for b0 in range(0, F, 682):
        e2.accumulate(h[b0:b0 + 682], None, hb[b0:b0 + 682])
    e2.synchronize(); t = stamp("accumulate in 682-frame calls", t)
    e2.close()
