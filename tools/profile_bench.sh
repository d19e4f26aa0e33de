#!/bin/bash
# kernel trace + stats of the default bench.py pipeline (all three stages, no extras): the matcher launch average here must
# agree with roofline.launch_ms of the bench line. Result: gpurun_out/prof_full; copy the kernel_stats.csv into profiles/.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/prof_full
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_full -- python3 bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --extras= > $OUT/prof_full.log 2>&1
tail -c 600 $OUT/prof_full.log
