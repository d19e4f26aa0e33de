#!/bin/bash
# quick look at the float matcher: parity tests, then the kernel trace of tools/time_l2.py -- as shipped, and with
# TODHIP_L2_NO_CANDIDATES=1 (pass 2 finds nothing: the bare GEMM + min tree; results of that run are garbage)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 500 python -m pytest tests/test_l2_gpu.py -x -q -m gpu 2>&1 | tail -3
for mode in shipped nocand; do
  rm -rf $OUT/prof_l2q_$mode
  if [ $mode = nocand ]; then export TODHIP_L2_NO_CANDIDATES=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_l2q_$mode -- python3 tools/time_l2.py > $OUT/prof_l2q_$mode.log 2>&1
  echo "$mode: $(grep match_l2 $OUT/prof_l2q_$mode.log)"
  python3 - $OUT/prof_l2q_$mode <<'PY'
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True))[-1:]:
    for row in csv.DictReader(open(f)):
        if "l2_" in row["Name"] and "prepare" not in row["Name"]: print("  %-70s calls %s avg %.1f us" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
done
