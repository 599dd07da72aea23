"""Small systems: brute-force tile kernel (algo=filter) against the cell-sorted culled kernel (algo=cell),
counts compared bit for bit, frames/s of each.  Where should algo=auto switch?"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mdhelper_amd import _core

for N in (16, 48, 100, 200, 333, 512, 777, 1000, 1023):
    L = 68.94 * (N / 32768.0) ** (1.0 / 3.0)
    F = max(2000, min(200000, int(4e8 / (N * N))))
    box = np.array([L, L, L, 90, 90, 90], dtype=np.float32)
    d = _core.synth_random_walk(F, N, box[:3], 0.3, seed=7)
    d_boxes = _core.DeviceArray.from_host(np.tile(box, (F, 1)))
    for rng in ((0.0, min(15.0, 0.45 * L)), (0.0, float(np.float32(L)) / 2)):
        edges = np.linspace(rng[0], rng[1], 202)
        res = {}
        for algo in ("filter", "cell"):
            eng = _core.RdfEngine(edges, (1, 1), algo=algo)
            eng.accumulate_device(d.ptr, N, None, N, d_boxes.ptr, F)
            eng.synchronize()
            eng.reset()
            t0 = time.perf_counter()
            eng.accumulate_device(d.ptr, N, None, N, d_boxes.ptr, F)
            eng.synchronize()
            dt = time.perf_counter() - t0
            res[algo] = (eng.counts(), F / dt)
            eng.close()
        same = np.array_equal(res["filter"][0], res["cell"][0])
        print(f"N={N:5d} range=(0,{rng[1]:6.2f}) F={F:6d}  filter {res['filter'][1]:12.0f} f/s  cell {res['cell'][1]:12.0f} f/s  "
              f"ratio {res['cell'][1] / res['filter'][1]:5.2f}  counts equal: {same}  binned {int(res['cell'][0].sum())}", flush=True)
    d.free(); d_boxes.free()
