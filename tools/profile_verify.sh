#!/bin/bash
# kernel trace + stats of todhip_verify_device on one bench frame (tools/verify_ticks.py) -> gpurun_out/prof_verify/
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_verify -- python3 tools/verify_ticks.py > $OUT/prof_verify.log 2>&1
cat $OUT/prof_verify/*/*_kernel_stats.csv | cut -c1-160
