#!/bin/bash
# locked runs for large pageable uploads (HostStager::copy_run_locked) against the ring (MDX_RING_UPLOAD=1), alternating
out=gpurun_out/r5q; mkdir -p $out
timeout -k 10 300 python scripts/diag/pageable_vs_ring.py > $out/pvr_locked.json 2>> $out/err.log; cat $out/pvr_locked.json
MDX_RING_UPLOAD=1 timeout -k 10 300 python scripts/diag/pageable_vs_ring.py > $out/pvr_ring.json 2>> $out/err.log; cat $out/pvr_ring.json
for rep in 1 2; do
  for mode in ring locked; do
    if [ $mode = ring ]; then export MDX_RING_UPLOAD=1; else unset MDX_RING_UPLOAD; fi
    timeout -k 10 300 python bench.py --workload ingest > $out/ingest_${mode}_$rep.json 2>> $out/err.log
    timeout -k 10 300 python bench.py --workload sq --steps 5 --warmup 2 --no-cpu-baseline > $out/sq_${mode}_$rep.json 2>> $out/err.log
    timeout -k 10 300 python bench.py --workload isf --steps 2 --warmup 1 --no-cpu-baseline > $out/isf_${mode}_$rep.json 2>> $out/err.log
  done
done
unset MDX_RING_UPLOAD
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5q/ingest_*.json")):
    g = json.load(open(f))["extra"]["ingest"]
    print(f.split("/")[-1], "rdf host %.3f pinned %.3f class_mem %.3f of resident" % (g["rdf_host"]["ratio_to_resident"], g["rdf_host_pinned"]["ratio_to_resident"], g["rdf_class_memory"]["ratio_to_resident"]))
for f in sorted(glob.glob("gpurun_out/r5q/sq_*.json")) + sorted(glob.glob("gpurun_out/r5q/isf_*.json")):
    d = json.load(open(f)); g = d.get("ingest", {})
    print(f.split("/")[-1], {k: (round(v.get("frames_per_sec", 0)), round(v.get("ratio_to_link_bound", v.get("ratio_to_resident", 0)), 3)) for k, v in g.items() if isinstance(v, dict) and "frames_per_sec" in v})
PY
