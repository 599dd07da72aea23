"""
One process per GPU on one node, without torch: process launch, rendezvous and a
host-side communicator.

The reference fans frames out to worker processes (``multiprocessing.Pool`` /
joblib / dask) and sums what they return on the parent (reference
src/mdhelper/analysis/base.py:385-386, :491-501; analysis/structure.py:841-844).
Here the workers are one process per GPU and the sum is one RCCL all-reduce; this
module is the control plane around that:

* ``launch(n, argv)``    — a parent that never touches the GPU starts ``n`` fresh
                           children (``subprocess``; no ``exec`` of a process that
                           has initialised HIP), one per device, hands them
                           ``RANK / LOCAL_RANK / WORLD_SIZE / MDX_RDZV_KEY``, relays
                           rank 0's stdout and the worst return code.
* ``Rendezvous``         — a tiny collective service among the ranks of the node on an
                           abstract-namespace unix socket hosted by rank 0 (nothing on
                           disk, nothing left behind when the processes die).  It ships
                           the 128-byte ``ncclUniqueId`` and serves the handful of host
                           scalars a run exchanges (timings, checks).
* ``SocketComm``         — a communicator (``rank``, ``world_size``, ``allreduce``,
                           ``barrier``) over that service for runs whose ranks cannot
                           form an RCCL communicator (several ranks sharing ONE GPU in
                           tests: RCCL refuses duplicate devices).  Host copies of the
                           accumulators are summed; ``device_collectives = False``.

Under ``torch.distributed.run`` (the driver's launch line) the same ``Rendezvous``
is used; its key is then derived from the launcher's pid and ``MASTER_PORT``.
"""

from __future__ import annotations

import json
import os
import socket
import struct
import subprocess
import sys
import tempfile
import threading
import time

import numpy as np

_OP_BCAST, _OP_SUM_F64, _OP_MAX_F64, _OP_SUM_I64, _OP_GATHER = range(5)
_HDR = struct.Struct("<IQ")


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send_frame(sock, op, payload):
    sock.sendall(_HDR.pack(op, len(payload)) + payload)


def _recv_frame(sock):
    op, n = _HDR.unpack(_recv_exact(sock, _HDR.size))
    return op, _recv_exact(sock, n)


def _combine(op, payloads):
    if op == _OP_BCAST:
        return payloads[0]
    if op == _OP_GATHER:
        return b"".join(payloads)
    dtype = np.int64 if op == _OP_SUM_I64 else np.float64
    arrs = [np.frombuffer(p, dtype=dtype) for p in payloads]
    if len({a.size for a in arrs}) != 1:
        raise ValueError("ranks entered an all-reduce with different sizes")
    out = arrs[0].copy()
    for a in arrs[1:]:                       # rank order: the float sum is reproducible
        out = np.maximum(out, a) if op == _OP_MAX_F64 else out + a
    return out.tobytes()


def rendezvous_key() -> str:
    """``MDX_RDZV_KEY`` (set by ``launch``), else one all workers of a ``torch.distributed.run``
    agent share: the agent's pid and its ``MASTER_PORT``."""
    key = os.environ.get("MDX_RDZV_KEY")
    if key:
        return key
    return f"{os.getppid()}-{os.environ.get('MASTER_PORT', '0')}"


class Rendezvous:
    """Collective byte/array exchange among ``world`` local processes; rank 0 hosts the service."""

    def __init__(self, rank: int, world: int, key: str | None = None, timeout: float = 300.0):
        self.rank, self.world = int(rank), int(world)
        self._addr = "\0mdx-rdzv-" + (key or rendezvous_key())
        self._server = None
        self._thread = None
        self._error = None
        self._timeout = float(timeout)
        if self.rank == 0:
            self._server = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            self._server.bind(self._addr)
            self._server.listen(self.world)
            self._thread = threading.Thread(target=self._serve, daemon=True)
            self._thread.start()
        self._sock = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        deadline = time.monotonic() + timeout
        while True:
            try:
                self._sock.connect(self._addr)
                break
            except (ConnectionRefusedError, FileNotFoundError):
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rank {rank}: no rendezvous service at {self._addr[1:]!r} "
                                       f"after {timeout:.0f} s (is rank 0 running?)")
                time.sleep(0.02)
        self._sock.settimeout(timeout)
        self._sock.sendall(struct.pack("<I", self.rank))

    @classmethod
    def from_env(cls, **kw):
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), **kw)

    def _serve(self):
        conns = [None] * self.world
        try:
            # every rank has to show up before the constructor's deadline (a rank that died on its way
            # must not leave rank 0 blocked in accept), and only processes of this user are let in: an
            # abstract socket has no file permissions, the peer's credentials stand in for them
            deadline = time.monotonic() + self._timeout
            missing = self.world
            while missing:
                left = deadline - time.monotonic()
                if left <= 0:
                    raise TimeoutError(f"{missing} of {self.world} ranks did not reach the rendezvous "
                                       f"within {self._timeout:.0f} s")
                self._server.settimeout(left)
                try:
                    c, _addr = self._server.accept()
                except socket.timeout:
                    continue
                _pid, uid, _gid = struct.unpack("3i", c.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED,
                                                                  struct.calcsize("3i")))
                if uid != os.getuid():
                    c.close()                # not ours: the slot stays open
                    continue
                c.settimeout(self._timeout)
                (r,) = struct.unpack("<I", _recv_exact(c, 4))
                if not 0 <= r < self.world or conns[r] is not None:
                    raise ValueError(f"unexpected rank {r} at the rendezvous")
                conns[r] = c
                missing -= 1
            for c in conns:
                c.settimeout(None)           # collectives wait as long as the slowest rank computes
            while True:
                frames = []
                for c in conns:
                    try:
                        frames.append(_recv_frame(c))
                    except ConnectionError:
                        return               # a rank left: the job is over
                ops = {f[0] for f in frames}
                if len(ops) != 1:
                    raise ValueError(f"ranks entered different collectives: {sorted(ops)}")
                op = ops.pop()
                out = _combine(op, [f[1] for f in frames])
                for c in conns:
                    _send_frame(c, op, out)
        except Exception as exc:             # surfaces on rank 0's next call; peers see EOF
            self._error = exc
        finally:
            for c in conns:
                if c is not None:
                    c.close()
            self._server.close()

    def _exchange(self, op, payload):
        if self._error is not None:
            raise RuntimeError(f"rendezvous service failed: {self._error}")
        try:
            _send_frame(self._sock, op, payload)
            rop, out = _recv_frame(self._sock)
        except (ConnectionError, OSError) as exc:
            # the service closes every connection when it fails: rank 0 reports WHY it did (its own
            # socket sees the EOF first), the other ranks the lost connection
            if self._thread is not None:
                self._thread.join(timeout=2)
            if self._error is not None:
                raise RuntimeError(f"rendezvous service failed: {self._error}") from exc
            raise
        assert rop == op
        return out

    def bcast(self, data: bytes | None) -> bytes:
        """Rank 0's bytes, on every rank."""
        return self._exchange(_OP_BCAST, data if self.rank == 0 else b"")

    def gather(self, data: bytes) -> bytes:
        """Concatenation of every rank's bytes in rank order, on every rank."""
        return self._exchange(_OP_GATHER, data)

    def allreduce(self, arr, op="sum"):
        a = np.ascontiguousarray(arr)
        if a.dtype == np.int64 and op == "sum":
            out = np.frombuffer(self._exchange(_OP_SUM_I64, a.tobytes()), dtype=np.int64)
            return out.reshape(a.shape).copy()
        code = _OP_MAX_F64 if op == "max" else _OP_SUM_F64
        out = np.frombuffer(self._exchange(code, a.astype(np.float64).tobytes()), dtype=np.float64)
        return out.reshape(a.shape).copy()

    def barrier(self):
        self._exchange(_OP_GATHER, b"")

    def close(self):
        try:
            self._sock.close()
        except OSError:
            pass
        if self._thread is not None:
            self._thread.join(timeout=5)


class SocketComm:
    """Host-side communicator over a ``Rendezvous`` (see the module docstring)."""

    device_collectives = False

    def __init__(self, rdzv: Rendezvous | None = None):
        self.rdzv = rdzv or Rendezvous.from_env()
        self.rank, self.world_size = self.rdzv.rank, self.rdzv.world

    def allreduce(self, arr, op="sum"):
        a = np.asarray(arr)
        out = self.rdzv.allreduce(a, op)
        return out if a.dtype in (np.int64, np.float64) else out.astype(a.dtype)

    def barrier(self):
        self.rdzv.barrier()

    def close(self):
        self.rdzv.close()


def visible_device_count(timeout: float = 300.0) -> int:
    """HIP devices visible to a child of this process, counted IN a child so that the caller
    stays free of any GPU state (it is about to start one process per device)."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from mdhelper_amd import _lib\n"
            "try:\n    print('MDX_DEVICES', _lib.device_count())\n"
            "except Exception:\n    print('MDX_DEVICES', 0)\n") % os.path.dirname(
                os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout)
    for line in out.stdout.splitlines():
        if line.startswith("MDX_DEVICES"):
            return int(line.split()[1])
    raise RuntimeError("could not count HIP devices: " + out.stderr.strip()[-400:])


def launch(n: int, argv: list[str], *, env: dict | None = None, share_devices: bool = False,
           timeout: float | None = None, poll: float = 0.05, check_devices: bool = True):
    """
    Start ``n`` ranks of ``argv`` (run with this interpreter) and wait for them.

    Returns ``(returncode, rank0_stdout)``.  The first rank that fails ends the job: the other
    children (exactly the pids started here) are terminated and its return code is reported.
    ``share_devices``: allow more ranks than devices (rank r uses device ``r % devices``; the ranks
    then need a ``SocketComm`` — RCCL cannot put two ranks on one GPU); otherwise fewer visible
    devices than ranks is an error naming the count.  ``check_devices=False`` skips the count (and the
    child process that takes it) for ranks that do not use a GPU.
    """
    n = int(n)
    # (check_devices=False: ranks that touch no GPU — bench.py --dry-run — are started without counting any)
    n_dev = visible_device_count() if check_devices else n
    if n_dev < n and not share_devices:
        raise RuntimeError(f"{n} ranks requested but only {n_dev} HIP device(s) are visible "
                           f"(one process per GPU; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES apply)")
    base = dict(os.environ if env is None else env)
    for name in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        base.pop(name, None)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on these hosts
    base["MDX_RDZV_KEY"] = f"{os.getpid()}-{time.monotonic_ns()}"
    base["MDX_VISIBLE_DEVICES"] = str(max(n_dev, 1))
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    try:
        for r in range(n):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n))
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=e,
                                          stdout=out0 if r == 0 else sys.stderr, stdin=subprocess.DEVNULL))
        deadline = None if timeout is None else time.monotonic() + timeout
        rc = 0
        live = set(range(n))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    sys.stderr.write(f"[mdx launch] rank {r} exited with code {code}; stopping the job\n")
            if rc != 0 or (deadline is not None and time.monotonic() > deadline):
                if rc == 0:
                    rc = 124
                    sys.stderr.write(f"[mdx launch] time limit of {timeout:.0f} s reached\n")
                break
            time.sleep(poll)
    finally:
        for p in procs:                      # exact pids only
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    out0.seek(0)
    text = out0.read()
    out0.close()
    return rc, text


def last_json_line(text: str):
    """The last line of ``text`` that parses as a JSON object (rank 0's result line), or None."""
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{"):
            try:
                return json.loads(line)
            except ValueError:
                continue
    return None
