"""LDS side of the MSD passes at C4 (diagnostic): python scripts/run/diag_msd_lds.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import make_counters as mc  # noqa: E402

args = ["--workload", "msd", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"]
out = {}
for ctrs in (["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS"],
             ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"],
             ["GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAIT_INST_ANY"]):
    try:
        line, res, calls, dur = mc.run_pmc("msd_lds", ctrs, args)
    except SystemExit as exc:
        print(ctrs, "failed:", exc, flush=True)
        continue
    for k in res:
        if "msd_fft" in k:
            name = "passA" if "cols" in k else "passB"
            for c in ctrs:
                out.setdefault(name, {})[c] = res[k][c] / max(len(calls[k]), 1)
            out[name]["dispatches"] = len(calls[k])
            out[name]["ms"] = dur[k] / 1e6 / max(len(calls[k]), 1)
print(json.dumps(out, indent=1))
