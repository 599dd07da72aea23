"""Child of tests/test_multirank_cpu.py: one gloo rank (or the single-rank run) of the sharded drivers on CPU.
torch is imported HERE, in a process of its own, never in the pytest process."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))

import test_multirank_cpu as tm  # noqa: E402

mode = sys.argv[1]
if mode == "rank":
    rank, world, port, out_dir = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import torch.distributed as dist
    from torch_comm import TorchDistComm
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tm._install_stand_ins()
    res = tm._analyses(TorchDistComm())
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()
elif mode == "single":
    tm._install_stand_ins()
    from mdhelper_amd.comm import SerialComm
    np.savez(os.path.join(sys.argv[2], "single.npz"), **tm._analyses(SerialComm()))
else:
    raise SystemExit(f"unknown mode {mode}")
