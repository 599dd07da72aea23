#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/final
export TMPDIR=/tmp
O=gpurun_out/final
timeout -k 10 900 python scripts/make_counters.py msd_c4 msd_tcc > $O/make_counters_msd.log 2>&1 || { tail -20 $O/make_counters_msd.log; exit 1; }
cp gpurun_out/counters/counters.json profiles/counters.json
python - <<'PY'
import sys; sys.path.insert(0,'scripts')
import make_counters as m
m.run_stats("msd_c4", ["--workload", "msd", "--steps", "3", "--no-cpu-baseline"])
PY
timeout -k 10 300 python bench.py > $O/bench_final.json 2> $O/bench_final.err || { tail -5 $O/bench_final.err; exit 1; }
timeout -k 10 200 python bench.py --workload msd --steps 20 --no-cpu-baseline > $O/bench_msd_20steps.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --workload msd --blocks 8 --steps 10 --no-cpu-baseline > $O/bench_msd8.json 2>/dev/null || exit 1
python -c "
import json; O='gpurun_out/final/'
d=json.load(open(O+'bench_final.json')); r=d['roofline']
print(d['frames_per_sec'], d['value'], d['ms_per_step'], r['frac'], r['frac_evaluations'], r['frac_binned'], r['kernel_ms_per_launch'], r['valu']['clock_hz'])
e=d['extra']; print(e['sq']['frames_per_sec'], e['sq']['roofline']['frac'], e['msd']['ms_per_step'], e['msd']['roofline']['frac'], e['msd']['roofline']['traffic'], e['rdf_wide']['frames_per_sec'], e['rdf_wide']['roofline']['frac'])
print({k:(round(v['frames_per_sec']),round(v['ratio_to_resident'],3)) for k,v in e['ingest'].items() if isinstance(v,dict) and 'frames_per_sec' in v})
for n in ('bench_msd_20steps','bench_msd8'):
    x=json.load(open(O+n+'.json')); print(n, x['ms_per_step'], x['roofline']['frac'])"
