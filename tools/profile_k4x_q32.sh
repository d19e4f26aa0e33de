#!/bin/bash
# hamming_topk_mfma_q32 in the memory-bound regime (tools/k4x_q32.py): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE each in a
# run of its own (no trace domains beside --pmc); digested into gpurun_out/k4x_q32_pmc.json with ONE correction rule, the one
# tools/fetch_calibration.sh measures for 16 B-per-lane loads.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/prof_q32_trace $OUT/prof_q32_f $OUT/prof_q32_w
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_q32_trace -- python3 tools/k4x_q32.py > $OUT/prof_q32_trace.log 2>&1 || { tail -5 $OUT/prof_q32_trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_q32_f -- python3 tools/k4x_q32.py > $OUT/prof_q32_f.log 2>&1 || { tail -5 $OUT/prof_q32_f.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_q32_w -- python3 tools/k4x_q32.py > $OUT/prof_q32_w.log 2>&1 || { tail -5 $OUT/prof_q32_w.log; exit 1; }
python3 - <<'PY'
import csv, glob, json, os, shutil
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
res = {"kernel": "hamming_topk_mfma_q32<2>", "workload": "16 queries x 40 000 000 DB rows (1.28 GB) per launch, k=2, radius 35 (bench.py hbm_regime)",
       "command": "tools/profile_k4x_q32.sh"}
for f in glob.glob(os.path.join(out, "prof_q32_trace", "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(out, "k4x_q32_kernel_stats.csv"))
    for row in csv.DictReader(open(f)):
        if "hamming_topk_mfma_q32" in row["Name"]:
            res["avg_duration_ns_kernel_trace"] = float(row["AverageNs"]); res["calls"] = int(row["Calls"])
for tag, ctr in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    for f in glob.glob(os.path.join(out, "prof_q32_" + tag, "**", "*counter_collection.csv"), recursive=True):
        acc, ids = 0.0, set()
        for row in csv.DictReader(open(f)):
            if "hamming_topk_mfma_q32" in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                acc += float(row["Counter_Value"]); ids.add(row["Dispatch_Id"])
        if ids: res[ctr + "_KB_per_launch"] = acc / len(ids)
res["live_launch_line"] = [l for l in open(os.path.join(out, "prof_q32_trace.log")).read().splitlines() if l.startswith("q32:")][-1:]
json.dump(res, open(os.path.join(out, "k4x_q32_pmc.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
