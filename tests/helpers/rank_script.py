"""Child of tests/test_launch_cpu.py: one rank started by mdhelper_amd.launch.launch (CPU only)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from mdhelper_amd.launch import Rendezvous, SocketComm  # noqa: E402

mode = sys.argv[1]
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])

if mode == "collectives":
    rdzv = Rendezvous.from_env()
    uid = rdzv.bcast(bytes(range(128)) if rank == 0 else None)
    ints = rdzv.allreduce(np.arange(5, dtype=np.int64) * (rank + 1))
    flt = rdzv.allreduce(np.array([[0.5 * rank, -1.0 - rank]]), op="sum")
    mx = rdzv.allreduce(np.array([float(rank), -float(rank)]), op="max")
    rdzv.barrier()
    names = rdzv.gather(f"r{rank};".encode())
    big = rdzv.allreduce(np.full(300_000, rank + 1.0))            # > one socket buffer
    rdzv.close()
    if rank == 0:
        print(json.dumps({"uid_ok": uid == bytes(range(128)), "ints": ints.tolist(), "flt": flt.tolist(),
                          "max": mx.tolist(), "names": names.decode(), "big": float(big.sum()),
                          "local_rank": os.environ["LOCAL_RANK"]}))
elif mode == "fail":
    rdzv = Rendezvous.from_env()
    if rank == 1:
        sys.exit(3)
    rdzv.allreduce(np.zeros(1))          # rank 0 would wait here forever: the launcher must stop it
elif mode == "analyses":
    import test_multirank_cpu as tm
    tm._install_stand_ins()
    comm = SocketComm()
    res = tm._analyses(comm)
    np.savez(os.path.join(sys.argv[2], f"rank{rank}.npz"), **res)
    comm.barrier()
    comm.close()
    if rank == 0:
        print(json.dumps({"done": True}))
elif mode == "rccl_or_socket":
    # no GPU here: ncclGetUniqueId fails on rank 0, every rank must land on the socket communicator and agree
    from mdhelper_amd.comm import rccl_comm_or_socket
    rdzv = Rendezvous.from_env()
    comm, kind = rccl_comm_or_socket(0, rdzv, timeout=30.0)
    total = comm.allreduce(np.array([rank + 1], dtype=np.int64))
    comm.barrier()
    comm.close()
    if rank == 0:
        print(json.dumps({"kind": kind, "class": type(comm).__name__, "total": int(total[0])}))
elif mode == "rccl_stuck":
    # ncclCommInitRank that never returns on rank 1 (a stand-in that sleeps): every rank must print the reason and
    # leave with comm.RCCL_TIMEOUT_EXIT — never carry on over the socket beside a thread blocked in RCCL
    import time
    from mdhelper_amd import _core, comm as mcomm

    class FakeRccl:
        device_collectives = True

        def __init__(self, r, w, uid, dev):
            if r == 1:
                time.sleep(3600)

        @staticmethod
        def unique_id():
            return bytes(128)

    _core.RcclComm = FakeRccl
    rdzv = Rendezvous.from_env()
    mcomm.rccl_comm_or_socket(0, rdzv, timeout=2.0)
    print(json.dumps({"continued": True}))          # must not be reached
elif mode == "dry_run":
    pass
