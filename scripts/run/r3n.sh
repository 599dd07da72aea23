#!/bin/bash
# RDF: one stream (new default) against MDX_RDF_OVERLAP=1 on one box: parity, headline + extras, C5 size, C1-like
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r3n
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "rdf or radial or c2 or c5 or beyond or traj or smoke or launch" > gpurun_out/r3n/pytest.log 2>&1
rc=$?; tail -n 3 gpurun_out/r3n/pytest.log; if [ $rc -ne 0 ]; then exit $rc; fi
for rep in 1 2; do
for mode in serial overlap; do
  if [ $mode = overlap ]; then export MDX_RDF_OVERLAP=1; else unset MDX_RDF_OVERLAP; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3n/bench_${mode}_$rep.json 2>/dev/null || exit 1
  timeout -k 10 300 python bench.py --atoms 131072 --frames 1000 --steps 2 --no-extras --no-cpu-baseline > gpurun_out/r3n/c5_${mode}_$rep.json 2>/dev/null || exit 1
done; done
unset MDX_RDF_OVERLAP
python - <<'PY'
import json
for rep in (1,2):
  for mode in ("serial","overlap"):
    d=json.load(open(f"gpurun_out/r3n/bench_{mode}_{rep}.json")); c=json.load(open(f"gpurun_out/r3n/c5_{mode}_{rep}.json"))
    ing={a:round(b["frames_per_sec"]) for a,b in d["extra"]["ingest"].items() if isinstance(b,dict) and "frames_per_sec" in b}
    print(mode, rep, "C2(i)", round(d["frames_per_sec"]), "frac", round(d["roofline"]["frac"],4), "wide", round(d["extra"]["rdf_wide"]["frames_per_sec"]), "C5", round(c["frames_per_sec"]), ing)
PY
