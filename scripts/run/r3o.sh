#!/bin/bash
# RDF ingest legs against the resident rate for pipeline slabs of 128 / 256 / 512 MiB (one box)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3o
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "traj or ingest or host or pipelin or universe" > gpurun_out/r3o/pytest.log 2>&1
rc=$?; tail -n 2 gpurun_out/r3o/pytest.log; if [ $rc -ne 0 ]; then exit $rc; fi
for mb in 128 256 512 256; do
  MDX_RDF_PIPE_MB=$mb timeout -k 10 300 python bench.py --workload ingest > gpurun_out/r3o/ingest_$mb.json 2>/dev/null || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/r3o/ingest_$mb.json"))
e=d.get("extra",{}).get("ingest") or d.get("ingest") or d
print($mb, {a:(round(b["frames_per_sec"]),round(b["ratio_to_resident"],3)) for a,b in e.items() if isinstance(b,dict) and "frames_per_sec" in b}, round(e.get("resident_frames_per_sec",0)))
PY
done
