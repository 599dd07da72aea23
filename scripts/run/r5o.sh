#!/bin/bash
# copy threads of the pinned ring: file legs (pread) and host legs by thread count
mkdir -p gpurun_out/io
for n in 8 12 16; do
  MDX_IO_THREADS=$n timeout -k 10 300 python bench.py --workload msd --steps 3 --warmup 5 --no-cpu-baseline > gpurun_out/io/msd_$n.json 2>/dev/null
  MDX_IO_THREADS=$n timeout -k 10 300 python bench.py --workload ingest > gpurun_out/io/ingest_$n.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/io/msd_$n.json"))["onsager"]
g=json.load(open("gpurun_out/io/ingest_$n.json"))["extra"]["ingest"]
print("threads $n: onsager host %.0f pinned %.0f file %.0f (first %.0f) ms; h2d ring %.1f GB/s | rdf host %.3f file %.3f class_file %.3f of resident, file->hbm %.1f GB/s" % (
  d["class_host_f32"]["ms_per_analysis"], d["class_host_f32_pinned"]["ms_per_analysis"], d["class_file"]["ms_per_analysis"], d["class_file"]["first_analysis_ms"], d["h2d_pageable_ring_GB_per_sec"],
  g["rdf_host"]["ratio_to_resident"], g["rdf_file"]["ratio_to_resident"], g["rdf_class_file"]["ratio_to_resident"], g["file_to_hbm"]["GB_per_sec"]))
PY
done
