"""
N-rank runs of the real engines on the GPU box, started by bench.py's own launcher (one fresh process
per rank, rendezvous over the node-local socket).  The box has ONE GPU and RCCL refuses two ranks on one
device, so the 2-rank runs use ``--share-devices`` (host all-reduce of the accumulators over the
rendezvous socket); what they prove is the launcher, the rank -> shard mapping and the reset / reduce
protocol of RdfEngine, SqEngine and MsdEngine: a FIXED frame / particle set gives the same result for
1 and 2 ranks (counts bit-identical).  The RCCL data plane itself runs with one rank
(``MDX_FORCE_COMM=1``), and with 2 real ranks wherever two devices are visible.
Reference counterpart: analysis/base.py:385-386, 491-501; analysis/structure.py:841-844.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _bench(*flags, env=None, timeout=900):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MDX_RDZV_KEY")}
    e.update(env or {})
    out = subprocess.run([sys.executable, BENCH, "--no-cpu-baseline", "--no-extras", *flags],
                         capture_output=True, text=True, timeout=timeout, env=e)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


RDF = ("--workload", "rdf", "--shard-fixed", "--frames", "37", "--atoms", "6000", "--steps", "2", "--warmup", "1")
SQ = ("--workload", "sq", "--shard-fixed", "--frames", "21", "--atoms", "5000", "--steps", "2", "--warmup", "1")
MSD = ("--workload", "msd", "--shard-fixed", "--frames", "4096", "--atoms", "101", "--steps", "2", "--warmup", "1")


def test_two_ranks_equal_one_rank_rdf_sq_msd():
    one = _bench(*RDF)
    two = _bench(*RDF, "--gpus", "2", "--share-devices")
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1 and two["scaling"] == "strong"
    assert two["result_digest"] == one["result_digest"]               # u64 counts: bit-identical
    assert two["value"] > 0 and len(two["per_rank_frames_per_sec"]) == 2
    assert all(v > 0 for v in two["per_rank_frames_per_sec"])
    assert two["comm"].startswith("host-socket") and two["rccl_ranks"] is None
    assert "launcher" in two
    one = _bench(*SQ)
    two = _bench(*SQ, "--gpus", "2", "--share-devices")
    assert np.allclose(two["result_digest"], one["result_digest"], rtol=1e-9, atol=1e-9)
    assert np.isclose(two["checksum"], one["checksum"], rtol=1e-9)
    one = _bench(*MSD)
    two = _bench(*MSD, "--gpus", "3", "--share-devices")
    assert two["n_gpus"] == 3
    assert np.allclose(two["result_digest"], one["result_digest"], rtol=1e-9, atol=1e-12)


def test_rccl_data_plane_single_rank_through_bench():
    out = _bench(*RDF, env={"MDX_FORCE_COMM": "1"})
    assert out["comm"] == "rccl" and out["rccl_ranks"] == 1
    assert out["result_digest"] == _bench(*RDF)["result_digest"]


def test_rccl_refusing_to_start_leaves_every_rank_on_the_socket():
    """Two ranks on ONE device ask for RCCL (``MDX_BENCH_TRY_RCCL=1``): RCCL refuses the duplicate device, and
    ``comm.rccl_comm_or_socket`` must bring every rank to the rendezvous socket with the reason in the line —
    the path a multi-GPU run takes if RCCL cannot start on its node — and the result must not change."""
    one = _bench(*RDF)
    two = _bench(*RDF, "--gpus", "2", "--share-devices",
                 env={"MDX_BENCH_TRY_RCCL": "1", "MDX_RCCL_INIT_TIMEOUT": "60"}, timeout=400)
    assert two["comm"].startswith("host-socket (RCCL unavailable: ncclCommInitRank"), two["comm"]
    assert two["rccl_ranks"] is None and two["n_gpus"] == 2
    assert two["result_digest"] == one["result_digest"]


def test_two_real_rccl_ranks_when_two_devices_are_visible():
    from mdhelper_amd import launch
    if launch.visible_device_count() < 2:
        pytest.skip("one device visible: RCCL cannot place two ranks on it")
    for flags in (RDF, SQ, MSD):
        one = _bench(*flags)
        two = _bench(*flags, "--gpus", "2")
        assert two["comm"] == "rccl" and two["rccl_ranks"] == 2
        if flags is RDF:
            assert two["result_digest"] == one["result_digest"]
        else:
            assert np.allclose(two["result_digest"], one["result_digest"], rtol=1e-9, atol=1e-9)


def test_more_ranks_than_devices_is_refused_with_the_count():
    from mdhelper_amd import launch
    n = launch.visible_device_count()
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, BENCH, "--gpus", str(n + 1)], capture_output=True, text=True,
                         timeout=300, env=e)
    assert out.returncode != 0 and f"only {n} HIP device" in out.stderr and out.stdout.strip() == ""
