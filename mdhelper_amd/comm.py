"""
Communicators for frame- / particle-sharded runs (one process per GPU).

* ``SerialComm``     — world of one (default).
* ``RcclComm``       — RCCL over xGMI through ``libmdx.so`` (``mdx_comm_*``); the
                       accumulators are all-reduced in HBM (``engine.allreduce``).
* ``launch.SocketComm`` — host all-reduce over the node-local rendezvous socket (no torch,
                       no RCCL): ranks that share one GPU in tests.

No torch here (one process, one ROCm runtime: ``_lib.runtime()``); the gloo communicator
of the world_size-2 CPU tests lives with them (``tests/helpers/torch_comm.py``).

The reference's counterpart is the per-frame gather-and-sum of
``ParallelAnalysisBase.run`` (reference src/mdhelper/analysis/base.py:396-501,
``np.vstack(...).sum(axis=0)`` at analysis/structure.py:842).
"""

from __future__ import annotations

import numpy as np


def shard_range(n: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous, balanced share [lo, hi) of ``n`` items for ``rank``."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class SerialComm:
    rank = 0
    world_size = 1
    device_collectives = False

    def allreduce(self, arr, op="sum"):
        return np.asarray(arr)

    def barrier(self):
        pass


def rccl_comm_from_env(device: int | None = None, rdzv=None):
    """
    Build an ``RcclComm`` for one rank of a one-process-per-GPU job (``launch.launch`` or
    ``torch.distributed.run``): rank 0 creates the RCCL unique id and ships its 128 bytes through
    the node-local ``launch.Rendezvous`` (control plane only); the data plane is RCCL inside
    ``libmdx.so``.  The rendezvous stays attached as ``comm.rdzv``.
    """
    import os

    from ._core import RcclComm
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if device is None:
        device = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and rdzv is None:
        return RcclComm(0, 1, RcclComm.unique_id(), device)
    if rdzv is None:
        from .launch import Rendezvous
        rdzv = Rendezvous(rank, world)
    uid = rdzv.bcast(RcclComm.unique_id() if rank == 0 else None)
    comm = RcclComm(rank, world, uid, device)
    comm.rdzv = rdzv
    return comm


RCCL_TIMEOUT_EXIT = 3      # exit code of a rank whose ncclCommInitRank never returned


def rccl_comm_or_socket(device: int, rdzv, timeout: float = 180.0):
    """
    ``(communicator, kind)`` for one rank of a one-process-per-GPU job: the RCCL communicator of
    ``rccl_comm_from_env`` when EVERY rank could build it (``kind == "rccl"``); ``launch.SocketComm``
    over the same rendezvous on every rank, with the reason in ``kind`` ("host-socket (RCCL
    unavailable: ...)"), when ``ncclGetUniqueId`` / ``ncclCommInitRank`` RETURNED AN ERROR on some rank.
    The decision is one all-reduce over the rendezvous, so the ranks cannot disagree.  What travels
    either way is the accumulators of an analysis (a histogram, a few spectra), once, at its end —
    never coordinates; the kernels do not change.

    An ``ncclCommInitRank`` that does not RETURN within ``timeout`` seconds is different: a thread of
    this process is then still inside RCCL on the device, and neither running the analysis beside it
    nor tearing HIP down under it at interpreter exit is safe.  Every rank learns of it in the same
    all-reduce, prints the reason and leaves with exit code ``RCCL_TIMEOUT_EXIT`` through ``os._exit``
    (no teardown under the blocked thread, no destruction of a half-built communicator).  The job
    fails loudly; it never continues on the socket in that case, and nothing is re-executed.

    For measurement harnesses and long jobs that must end with a result where RCCL reports that it
    cannot start; code that requires RCCL calls ``rccl_comm_from_env`` and lets the error surface.
    """
    import os
    import sys
    import threading

    from . import _core
    from .launch import SocketComm
    rank, world = rdzv.rank, rdzv.world
    err, uid = None, b""
    if rank == 0:
        try:
            uid = bytes(_core.RcclComm.unique_id())
        except Exception as exc:                       # noqa: BLE001 - reported through `kind`
            err, uid = f"ncclGetUniqueId: {exc}", b""
    uid = rdzv.bcast(uid if rank == 0 else None)
    comm, stuck = None, False
    if uid:
        box = {}

        def init():
            try:
                box["comm"] = _core.RcclComm(rank, world, uid, device)
            except Exception as exc:                   # noqa: BLE001
                box["err"] = f"ncclCommInitRank: {exc}"

        th = threading.Thread(target=init, daemon=True)
        th.start()
        th.join(timeout)
        if th.is_alive():
            stuck = True
            err = f"ncclCommInitRank did not return within {timeout:g} s on rank {rank}"
        else:
            comm, err = box.get("comm"), box.get("err")
    elif err is None:
        err = "rank 0 could not create the RCCL unique id"
    flags = rdzv.allreduce(np.array([1 if comm is not None else 0, 1 if stuck else 0], dtype=np.int64))
    built, n_stuck = int(flags[0]), int(flags[1])
    if built == world:
        comm.rdzv = rdzv
        return comm, "rccl"
    reasons = rdzv.gather(((err or "") + "\n").encode()).decode().splitlines()
    if n_stuck:
        why = next((r for r in reasons if "did not return" in r), "ncclCommInitRank did not return")
        sys.stderr.write(f"mdhelper_amd: rank {rank}: RCCL did not start ({why[:200]}; {n_stuck} of {world} ranks "
                         f"still inside ncclCommInitRank); leaving with exit code {RCCL_TIMEOUT_EXIT}\n")
        sys.stderr.flush()
        sys.stdout.flush()
        # no interpreter teardown on any rank: a thread is blocked in RCCL on the stuck ranks' devices, and on the
        # others a communicator whose peers never arrived must not be destroyed (ncclCommDestroy can block too)
        os._exit(RCCL_TIMEOUT_EXIT)
    # (a communicator that some ranks built and others did not is left alone: destroying it can block on the
    # peers that never arrived)
    why = next((r for r in reasons if r), "unknown")
    return SocketComm(rdzv), f"host-socket (RCCL unavailable: {why[:200]})"
