#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace statistics of bench.py
# workloads; summaries land in gpurun_out/prof_<tag>/ and are copied to profiles/ by hand.
# usage: scripts/profile_gpu.sh <tag> <bench args...>
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o "$tag" -- python3 "$root/bench.py" "$@" > "$out/bench.json" 2> "$out/bench.err"
rc=$?
# keep only the small summaries (the per-dispatch trace can be large)
find "$out" -name "*kernel_stats.csv" -exec cp {} "$out/${tag}_kernel_stats.csv" \; 2>/dev/null
find "$out" -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
cat "$out/bench.json"
exit $rc
