#!/bin/bash
# kernel trace + stats of the default bench on the GPU box -> gpurun_out/prof_<tag>/
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
TAG=${1:-trace}
shift || true
OUT=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $OUT/prof_$TAG.log 2>&1
cat $OUT/prof_$TAG/*/*_kernel_stats.csv | cut -c1-200
