"""Per-chunk host times of Onsager's column streaming at C4 (pageable float32): upload / push / synchronize."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mdhelper_amd import _core

T, N, chunk = 100000, 10000, 1248
d = _core.synth_random_walk(T, N, [1, 1, 1], 0.1, seed=4, wrap=False)
h = d.to_host()
d.free()
for rep in range(3):
    eng = _core.MsdEngine(T, 1, 2)
    bufs = [_core.DeviceArray((T, chunk, 3), np.float32) for _ in range(2)]
    k = 0
    rows = []
    t_all = time.perf_counter()
    for g, (first, count) in enumerate(((0, 5000), (5000, 5000))):
        for a in range(0, count, chunk):
            c = min(chunk, count - a)
            t0 = time.perf_counter()
            if k >= 2:
                eng.synchronize()
            t1 = time.perf_counter()
            buf = bufs[k & 1] if c == chunk else _core.DeviceArray.view(bufs[k & 1], (T, c, 3))
            buf.upload_columns(h, first + a, c)
            t2 = time.perf_counter()
            eng.push_frames_device(g, buf, c, None)
            t3 = time.perf_counter()
            rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
            k += 1
    eng.synchronize()
    total = (time.perf_counter() - t_all) * 1e3
    print(f"rep {rep}: total {total:.1f} ms; per chunk (sync, upload, push) ms:", " ".join(f"({a:.1f},{b:.1f},{c:.1f})" for a, b, c in rows), flush=True)
    eng.close()
    for b in bufs:
        b.free()
