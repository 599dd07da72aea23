#!/bin/bash
# usage: scripts/dbg_sweep.sh VAR "v1 v2 ..." <bench args>   (GPU box)
var=$1; vals=$2; shift 2
for v in $vals; do
  env $var=$v python bench.py "$@" --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$var=$v', 'ms/step %.2f' % d['ms_per_step'], 'kernel_ms %.2f' % (r.get('kernel_ms_per_step') or r.get('kernel_ms_per_launch') or 0), d.get('frames_per_sec',''))"
done
