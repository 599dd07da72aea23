"""L2 / L1 behaviour of the RDF kernels at C2(i) (diagnostic): python scripts/run/diag_rdf_cache.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import make_counters as mc  # noqa: E402

frames = 2000
args = ["--workload", "rdf", "--frames", str(frames), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"]
out = {}
for ctrs in (["TCC_HIT", "TCC_MISS", "TCC_REQ", "TCC_READ"],
             ["TCC_EA0_RDREQ", "TCC_TAG_STALL", "TCC_BUSY", "TCC_CYCLE"],
             ["TCP_TOTAL_CACHE_ACCESSES", "TCP_TCC_READ_REQ", "TCP_PENDING_STALL_CYCLES", "TCP_TCR_TCP_STALL_CYCLES"],
             ["TCP_TOTAL_ACCESSES", "TCP_TCC_READ_REQ_LATENCY", "TCP_TCC_NC_READ_REQ", "TCP_TCC_UC_READ_REQ"]):
    try:
        line, res, calls, dur = mc.run_pmc("rdf_cache", ctrs, args)
    except SystemExit as exc:
        print(ctrs, "failed:", exc, flush=True)
        continue
    for k in res:
        if "rdf_cell" in k:
            name = "pair" if "pair" in k else "sort"
            for c in ctrs:
                out.setdefault(name, {})[c] = res[k][c] / frames
print(json.dumps(out, indent=1))
