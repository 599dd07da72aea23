"""
Test-only communicator over an initialised ``torch.distributed`` process group (``gloo`` on CPU): the host copies
of the accumulators are all-reduced.  It lives under tests/ and not in the product package: mdhelper_amd has no
torch in its path (its ranks meet over ``launch.Rendezvous`` and reduce through RCCL inside libmdx.so), and a
torch imported beside libmdx.so brings a second ROCm runtime into the process (``_lib.runtime()`` refuses that).
The reference's counterpart is the gather-and-sum of ``ParallelAnalysisBase.run`` (reference
src/mdhelper/analysis/base.py:396-501, ``np.vstack(...).sum(axis=0)`` at analysis/structure.py:842).
"""
import numpy as np


class TorchDistComm:
    device_collectives = False

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised.")
        if dist.get_backend(group) != "gloo":
            raise RuntimeError("TorchDistComm is the CPU (gloo) test communicator; GPU ranks use mdhelper_amd's "
                               "RcclComm (comm.rccl_comm_from_env)")
        self._dist = dist
        self._group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)

    def allreduce(self, arr, op="sum"):
        import torch
        a = np.ascontiguousarray(arr)
        t = torch.from_numpy(a.copy())
        red = self._dist.ReduceOp.SUM if op == "sum" else self._dist.ReduceOp.MAX
        self._dist.all_reduce(t, op=red, group=self._group)
        return t.numpy().astype(a.dtype, copy=False)

    def barrier(self):
        self._dist.barrier(group=self._group)
