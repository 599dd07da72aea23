"""
The torch-free launcher and rendezvous (mdhelper_amd/launch.py) on CPU: collectives among fresh child
processes, failure handling, and the sharded analysis drivers at world_size 2 through the launcher +
SocketComm (device engines replaced by oracle stand-ins IN THE CHILDREN only), equal to a single rank.
The reference's counterpart: the worker fan-out and parent-side sum of analysis/base.py:385-386,
491-501 and analysis/structure.py:841-844.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tests", "helpers", "rank_script.py")


def test_rendezvous_collectives_three_ranks():
    from mdhelper_amd import launch
    rc, text = launch.launch(3, [SCRIPT, "collectives"], share_devices=True, timeout=120)
    assert rc == 0, text
    out = launch.last_json_line(text)
    assert out["uid_ok"] and out["local_rank"] == "0"
    assert out["ints"] == (np.arange(5) * 6).tolist()
    assert out["flt"] == [[1.5, -6.0]]
    assert out["max"] == [2.0, 0.0]
    assert out["names"] == "r0;r1;r2;"
    assert out["big"] == 300_000 * 6.0


def test_rccl_or_socket_falls_back_on_every_rank_when_rccl_cannot_start():
    """bench.py's multi-rank communicator: without a usable RCCL (no GPU here) every rank must end up on the
    rendezvous socket, with the reason recorded, and the job must still reduce correctly."""
    from mdhelper_amd import _lib, launch
    if _lib.device_count() > 0:
        pytest.skip("a GPU is present: RCCL starts")
    rc, text = launch.launch(3, [SCRIPT, "rccl_or_socket"], share_devices=True, timeout=180)
    assert rc == 0, text
    out = launch.last_json_line(text)
    assert out["class"] == "SocketComm" and out["total"] == 6
    assert out["kind"].startswith("host-socket (RCCL unavailable: ncclGetUniqueId")


def test_rccl_init_that_never_returns_ends_the_job_with_its_exit_code():
    """A rank whose ncclCommInitRank does not return has a thread inside RCCL on its device: the job must
    end — every rank, with the reason on stderr and comm.RCCL_TIMEOUT_EXIT — instead of continuing over the
    socket."""
    from mdhelper_amd import comm
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from mdhelper_amd import launch\n"
            "rc, text = launch.launch(3, [%r, 'rccl_stuck'], share_devices=True, timeout=120)\n"
            "print('RC', rc); print(text)\n") % (ROOT, SCRIPT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=e)
    assert f"RC {comm.RCCL_TIMEOUT_EXIT}" in out.stdout, out.stdout + out.stderr
    assert "continued" not in out.stdout
    assert "ncclCommInitRank did not return within 2 s on rank 1" in out.stderr
    assert "leaving with exit code" in out.stderr


@pytest.mark.parametrize("flags", [[], ["--workload", "msd", "--shard-fixed"], ["--workload", "sq", "--shard-fixed"]])
def test_bench_dry_run_walks_the_eight_rank_control_plane(flags):
    """``bench.py --gpus 8 --dry-run``: launcher, rendezvous, id broadcast, shard plan and a host all-reduce for
    the rank count the driver's scaling run uses, engines left out, no GPU touched — what stays untested on
    the real node is RCCL itself."""
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run", *flags],
                         capture_output=True, text=True, timeout=300, env=e)
    assert out.returncode == 0, out.stderr
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["dry_run"] and line["n_gpus"] == 8 and line["rccl"] is False
    assert line["id_broadcast_ok"] and line["plan_covers_the_work"]
    assert line["ranks_in_plan"] == list(range(8)) and line["devices"] == list(range(8))
    unit = line["unit"]
    spans = [tuple(x) for p in line["plan"] for x in p[unit]]
    if "--shard-fixed" in flags:
        # contiguous, disjoint, balanced shares
        sizes = [sum(hi - lo for lo, hi in p[unit]) for p in line["plan"]]
        assert max(sizes) - min(sizes) <= (2 if unit == "particles" else 1)
        assert len(set(spans)) == len(spans)
    else:
        assert len(set(spans)) == 1                      # weak scaling: every rank its own full batch
    assert line["launcher"].startswith("bench.py")


def test_bench_dry_run_under_torch_distributed_run():
    """The driver's launch line (torch.distributed.run, one rank per GPU) reaches the same rendezvous."""
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--dry-run", "--shard-fixed"],
                         capture_output=True, text=True, timeout=600, env=e)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["plan_covers_the_work"] and line["id_broadcast_ok"]
    assert [p["frames"] for p in line["plan"]] == [[[0, 5000]], [[5000, 10000]]]


def test_launcher_stops_the_job_when_a_rank_fails():
    from mdhelper_amd import launch
    # rank 1 exits with code 3 while rank 0 sits in a collective: the job must end, non-zero, promptly
    # (either the launcher sees rank 1's code first, or rank 0 sees the service go away and fails)
    rc, _text = launch.launch(2, [SCRIPT, "fail"], share_devices=True, timeout=120)
    assert rc in (1, 3)


def test_launcher_refuses_more_ranks_than_devices():
    from mdhelper_amd import _lib, launch
    n = _lib.device_count()
    ranks = max(2, n + 1)
    with pytest.raises(RuntimeError, match=rf"only {n} HIP device"):
        launch.launch(ranks, [SCRIPT, "collectives"])
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks)],
                         capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert out.returncode != 0 and f"only {n} HIP device" in out.stderr and out.stdout.strip() == ""


def test_world_size_2_through_the_launcher_matches_single_rank(tmp_path):
    from mdhelper_amd import launch
    rc, text = launch.launch(2, [SCRIPT, "analyses", str(tmp_path)], share_devices=True, timeout=600)
    assert rc == 0 and launch.last_json_line(text) == {"done": True}
    rc, _ = launch.launch(1, [SCRIPT, "analyses", str(tmp_path / "one")], share_devices=True, timeout=600) \
        if (tmp_path / "one").mkdir() is None else (1, "")
    assert rc == 0
    single = np.load(tmp_path / "one" / "rank0.npz")
    for rank in range(2):
        got = np.load(tmp_path / f"rank{rank}.npz")
        for name in ("counts", "counts_slow", "counts_com"):
            assert np.array_equal(got[name], single[name]), name          # integer sums: bit-exact
        for name in ("rdf", "ssf", "cisf", "iisf", "msd_self", "msd_cross", "acf", "ssf_res", "cisf_res",
                     "iisf_res"):
            assert np.allclose(got[name], single[name], rtol=1e-9, atol=1e-10), name
    assert single["counts"].sum() > 0


def test_rendezvous_under_torch_distributed_run():
    """The driver's launch line: no MDX_RDZV_KEY, the key comes from the agent's pid and MASTER_PORT."""
    port = 29600 + os.getpid() % 1000
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MDX_RDZV_KEY")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), SCRIPT, "collectives"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    from mdhelper_amd import launch
    res = launch.last_json_line(out.stdout)
    assert res["uid_ok"] and res["ints"] == (np.arange(5) * 3).tolist() and res["names"] == "r0;r1;"


def test_rendezvous_does_not_wait_forever_for_a_rank_that_never_came():
    """A rank that died before connecting must not leave rank 0 blocked in accept(): the service gives
    up at the constructor's deadline and rank 0's next collective says why."""
    import os
    import time
    from mdhelper_amd.launch import Rendezvous
    r0 = Rendezvous(0, 2, key=f"test-missing-{os.getpid()}", timeout=1.0)
    time.sleep(1.3)
    with pytest.raises(RuntimeError, match="did not reach the rendezvous"):
        r0.barrier()
    r0.close()


def test_rendezvous_reports_the_service_error_on_rank_zero():
    """Ranks entering different collectives: the service fails and closes every connection; rank 0 —
    whose own socket sees the EOF first — reports the stored reason, not a bare ConnectionError."""
    import os
    import threading
    from mdhelper_amd.launch import Rendezvous
    key = f"test-mismatch-{os.getpid()}"
    out = {}

    def rank1():
        r1 = Rendezvous(1, 2, key=key, timeout=20.0)
        try:
            r1.allreduce(np.arange(3, dtype=np.int64))
        except Exception as exc:           # EOF or reset: the service went away
            out["r1"] = exc
        r1.close()

    r0 = Rendezvous(0, 2, key=key, timeout=20.0)
    t = threading.Thread(target=rank1)
    t.start()
    with pytest.raises(RuntimeError, match="different collectives"):
        r0.barrier()
    t.join(timeout=20)
    assert isinstance(out.get("r1"), (ConnectionError, OSError))
    r0.close()
