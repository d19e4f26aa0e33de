#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the calibration kernels (tools/fetch_calibration.hip), each counter in a run of its own (no trace
# domains beside --pmc), digested into gpurun_out/fetch_calibration.json: counter x 1024 / bytes streamed per kernel.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
BYTES=${1:-2147483648}
rm -rf $OUT/prof_cal_f $OUT/prof_cal_w
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_cal_f -- tools/fetch_calibration $BYTES > $OUT/prof_cal_f.log 2>&1 || { tail -5 $OUT/prof_cal_f.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/prof_cal_w -- tools/fetch_calibration $BYTES > $OUT/prof_cal_w.log 2>&1 || { tail -5 $OUT/prof_cal_w.log; exit 1; }
python3 - $BYTES <<'PY'
import csv, glob, json, os, sys
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out"); nbytes = int(sys.argv[1])
res = {}
for tag, ctr in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    for f in glob.glob(os.path.join(out, "prof_cal_" + tag, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != ctr: continue
            name = row["Kernel_Name"].split("(")[0]
            per.setdefault(name, {}).setdefault(row["Dispatch_Id"], 0.0)
            per[name][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for name, d in per.items():
            vals = list(d.values())
            res.setdefault(name, {})[ctr + "_KB_per_dispatch"] = sum(vals) / len(vals)
for name, r in sorted(res.items()):
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        if ctr + "_KB_per_dispatch" in r: r[ctr + "_bytes_over_streamed"] = r[ctr + "_KB_per_dispatch"] * 1024.0 / nbytes
    print(name, {k: round(v, 4) for k, v in r.items()})
json.dump({"bytes_streamed_per_dispatch": nbytes, "command": "tools/fetch_calibration.sh (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs)",
           "kernels": res, "timings": open(os.path.join(out, "prof_cal_f.log")).read().strip().splitlines()[-5:]},
          open(os.path.join(out, "fetch_calibration.json"), "w"), indent=1)
PY
