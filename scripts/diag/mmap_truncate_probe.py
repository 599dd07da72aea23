"""Which call hangs when a mapped trajectory file loses its tail?  (tests/test_gpu_traj.py::
test_load_columns_of_a_file_that_lost_its_tail_is_an_io_error hung on the first mapped build.)  Every step prints
before it starts; run under `timeout`.    python scripts/diag/mmap_truncate_probe.py [unmap]"""
import faulthandler
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
faulthandler.enable()
faulthandler.dump_traceback_later(40, exit=True)
from mdhelper_amd import _core  # noqa: E402
from mdhelper_amd.io import TrajectoryFile  # noqa: E402
from trajfiles import write_amber_netcdf  # noqa: E402


def say(*a):
    print(*a, flush=True)


F, N, L = 64, 6000, 38.0
rng = np.random.default_rng(1)
pos = rng.uniform(0, L, (F, N, 3)).astype(np.float32)
d = tempfile.mkdtemp()
path = os.path.join(d, "cut.nc")
write_amber_netcdf(path, pos, (L, L, L))
t = TrajectoryFile(path)
out = _core.DeviceArray((F, 3000, 3), np.float32)
say("1 load all (maps the file)")
t.load_columns_device(np.arange(F), 1000, 3000, out.ptr)
say("1 ok", np.array_equal(out.to_host(), pos[:, 1000:4000]))
say("2 truncate")
os.truncate(path, os.path.getsize(path) - 20 * 12 * N)
say("3 load all again (expect OSError)")
try:
    t.load_columns_device(np.arange(F), 1000, 3000, out.ptr)
    say("3 no error?!")
except OSError as e:
    say("3 OSError:", str(e)[:100])
say("4 device still alive? small kernel-free copy")
x = _core.DeviceArray.from_host(np.arange(10.0))
say("4 ok", x.to_host()[:3])
say("5 load the first 32 frames")
t.load_columns_device(np.arange(32), 1000, 3000, out.ptr)
say("5 ok", np.array_equal(_core.DeviceArray.view(out, (32, 3000, 3)).to_host(), pos[:32, 1000:4000]))
t.close()
say("done")
