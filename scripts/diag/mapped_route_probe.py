"""The opt-in mapped route of mdx_traj_load_device (explicit lock / copy / unlock per slice): bytes against the host
reader, rates (first read of the fresh file, repeats) against the pread ring, and the truncation probe of
scripts/diag/mmap_truncate_probe.py on it — nothing may stay registered after a call, so a file that loses its tail
AFTERWARDS must leave the device usable.  Every step prints before it starts; run under `timeout`.
    python scripts/diag/mapped_route_probe.py [T] [N]
(The record of what was run: `TrajectoryFile(path, mapped=True)` belongs to the patch that was measured with this script
and not committed — NOTES.md round 5; the script does not run against the library as it is.)"""
import faulthandler
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
faulthandler.enable()
faulthandler.dump_traceback_later(150, exit=True)
import bench  # noqa: E402
from mdhelper_amd import _core  # noqa: E402
from mdhelper_amd.io import TrajectoryFile  # noqa: E402


def say(*a):
    print(*a, flush=True)


T = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
d = _core.synth_random_walk(T, N, [50, 50, 50], 0.1, seed=4, wrap=True)
h = d.to_host()
gb = h.nbytes / 1e9
res = {"T": T, "N": N, "GB": round(gb, 2)}
tmp = tempfile.NamedTemporaryFile(suffix=".nc", delete=False)
tmp.close()
bench.write_amber_netcdf_fast(tmp.name, h, np.array([50, 50, 50, 90, 90, 90], dtype=np.float32))
frames = np.arange(T)


def rate(t):
    t0 = time.perf_counter()
    t.load_device(frames, d.ptr, dev=0)
    _core.synchronize(0)
    return round(gb / (time.perf_counter() - t0), 1)


say("1 mapped: first read of the fresh file, then repeats")
tm = TrajectoryFile(tmp.name, mapped=True)
res["mapped_GBps"] = [rate(tm) for _ in range(3)]
res["mapped_bytes_equal"] = bool(np.array_equal(d.to_host(), h))
say("2 pread ring")
tr = TrajectoryFile(tmp.name)
res["pread_ring_GBps"] = [rate(tr) for _ in range(3)]
res["ring_bytes_equal"] = bool(np.array_equal(d.to_host(), h))
tr.close()
say(json.dumps(res))
say("3 truncate the file after the mapped reads")
os.truncate(tmp.name, os.path.getsize(tmp.name) // 2)
say("4 device still alive?")
x = _core.DeviceArray.from_host(np.arange(10.0))
say("4 ok", x.to_host()[:3])
say("5 mapped read of all frames (expect OSError)")
try:
    tm.load_device(frames, d.ptr, dev=0)
    say("5 no error?!")
except OSError as e:
    say("5 OSError:", str(e)[:90])
say("6 mapped read of the first third")
third = np.arange(T // 3)
tm.load_device(third, d.ptr, dev=0)
say("6 ok", bool(np.array_equal(_core.DeviceArray.view(d, (T // 3, N, 3)).to_host(), h[:T // 3])))
tm.close()
os.unlink(tmp.name)
say("done")
