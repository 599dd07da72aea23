#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/final
timeout -k 10 300 python bench.py > gpurun_out/final/bench_final.json 2> gpurun_out/final/bench_final.err || { tail -5 gpurun_out/final/bench_final.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/final/bench_final.json')); r=d['roofline']
print(d['frames_per_sec'], d['value'], d['ms_per_step'], r['frac'], r['frac_evaluations'], r['frac_binned'], r['work']['evaluations_per_binned_unordered_pair'], r['kernel_ms_per_launch'], r['valu']['clock_hz'])
e=d['extra']; print(e['sq']['frames_per_sec'], e['sq']['roofline']['frac'], e['msd']['ms_per_step'], e['rdf_wide']['frames_per_sec'], e['rdf_wide']['roofline']['frac'])
print({k:(round(v['frames_per_sec']),round(v['ratio_to_resident'],3)) for k,v in e['ingest'].items() if isinstance(v,dict) and 'frames_per_sec' in v})"
