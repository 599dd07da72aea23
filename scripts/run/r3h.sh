#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/single
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/single -o single -- python3 $GRAFT_REPO_ROOT/scripts/run/diag_single.py > $GRAFT_REPO_ROOT/gpurun_out/single/out.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/single/out.txt | tail -6
find $GRAFT_REPO_ROOT/gpurun_out/single -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-60,150-260 | head -12
