#!/bin/bash
# The round's kernel traces, one rocprofv3 run each (kernel trace + stats only), copied to gpurun_out/r03_*.csv:
#   full       the default bench.py command without extras (the matcher launch average must agree with roofline.launch_ms)
#   chained    the verifier on one 16-frame data-chained batch, twice (tools/chained_ticks.py child 16)
#   l2         the float matcher at the C4 shape (tools/time_l2.py) + its FETCH_SIZE pass
#   pipeline   bench.py's chained block (the data-chained pipeline: whole-block matcher on the trained DB, ORB, the verifier's kernels)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/prof_full $OUT/prof_cv $OUT/prof_l2 $OUT/prof_l2_f $OUT/prof_cp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cp -- python3 bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --stages match --extras chained > $OUT/prof_cp.log 2>&1 || { tail -5 $OUT/prof_cp.log; exit 1; }
find $OUT/prof_cp -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r03_chained_pipeline_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_full -- python3 bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --extras= > $OUT/prof_full.log 2>&1 || { tail -5 $OUT/prof_full.log; exit 1; }
find $OUT/prof_full -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r03_full_kernel_stats.csv
tail -c 400 $OUT/prof_full.log | head -c 400; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cv -- python3 tools/chained_ticks.py child 16 > $OUT/prof_cv.log 2>&1 || { tail -5 $OUT/prof_cv.log; exit 1; }
find $OUT/prof_cv -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r03_chained_verify_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_l2 -- python3 tools/time_l2.py > $OUT/prof_l2.log 2>&1 || { tail -5 $OUT/prof_l2.log; exit 1; }
find $OUT/prof_l2 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/r03_l2_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_l2_f -- python3 tools/time_l2.py > $OUT/prof_l2_f.log 2>&1 || { tail -5 $OUT/prof_l2_f.log; exit 1; }
python3 - <<'PY'
import csv, glob, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
for f in ("r03_full_kernel_stats.csv", "r03_chained_verify_kernel_stats.csv", "r03_l2_kernel_stats.csv"):
    print("==", f)
    for i, row in enumerate(csv.DictReader(open(os.path.join(out, f)))):
        if i < 12: print("  %-70s calls %5s avg %10.1f us  %5s %%" % (row["Name"].replace("(anonymous namespace)::", "")[:70], row["Calls"], float(row["AverageNs"]) / 1e3, row["Percentage"]))
lines = []
for f in glob.glob(os.path.join(out, "prof_l2_f", "**", "*counter_collection.csv"), recursive=True):
    acc, ids = {}, {}
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        if "l2_gemm_kernel" not in k or row["Counter_Name"] != "FETCH_SIZE": continue
        key = (k, row["Grid_Size"])
        acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"]); ids.setdefault(key, set()).add(row["Dispatch_Id"])
    for key in sorted(acc):
        kb = acc[key] / len(ids[key])
        lines.append("%s grid %s: FETCH_SIZE %.1f KB per launch raw = %.1f MB; x2 (vector loads, profiles/r03_fetch_calibration.json) = %.1f MB" % (key[0], key[1], kb, kb * 1024 / 1e6, 2 * kb * 1024 / 1e6))
open(os.path.join(out, "r03_l2_pmc.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
