#!/bin/bash
# quick look at the float matcher: parity tests, then the kernel trace of tools/time_l2.py with the wide and the narrow GEMM
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout -k 10 500 python -m pytest tests/test_l2_gpu.py -x -q -m gpu 2>&1 | tail -3
for wide in 1 0; do
  rm -rf $OUT/prof_l2q$wide
  TODHIP_L2_WIDE=$wide rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_l2q$wide -- python3 tools/time_l2.py > $OUT/prof_l2q$wide.log 2>&1
  echo "wide=$wide: $(grep match_l2 $OUT/prof_l2q$wide.log)"
  python3 - $OUT/prof_l2q$wide <<'PY'
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True))[-1:]:
    for row in csv.DictReader(open(f)):
        if "l2_gemm" in row["Name"]: print("  %-70s calls %s avg %.1f us" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
done
