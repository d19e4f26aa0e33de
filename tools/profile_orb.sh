#!/bin/bash
# per-kernel times of one ORB batch (32 frames), alone: the 8(d) synthetic image and the chained block's rendered views
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for S in "" 1; do
  tag=orb_synth; [ -n "$S" ] && tag=orb_scenes
  rm -rf $OUT/prof_$tag
  B=32 SCENES=$S rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$tag -- python3 tools/time_orb_batch.py > $OUT/prof_$tag.log 2>&1 || { tail -5 $OUT/prof_$tag.log; exit 1; }
  tail -1 $OUT/prof_$tag.log
  find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r03_${tag}_kernel_stats.csv
  python3 - $OUT/r03_${tag}_kernel_stats.csv <<'PY'
import csv, sys
tot = 0
for row in csv.DictReader(open(sys.argv[1])):
    n = row["Name"].replace("(anonymous namespace)::", "").split("(")[0]
    if "at::" in n or "rocclr" in n: continue
    per = float(row["TotalDurationNs"]) / 23 / 1e3
    tot += per
    print("  %-28s calls %4s avg %8.1f us   per batch %8.1f us" % (n[:28], row["Calls"], float(row["AverageNs"]) / 1e3, per))
print("  sum per batch %.1f us" % tot)
PY
done
