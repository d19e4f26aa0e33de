#!/bin/bash
# K4x alone (tools/k4x_one.py): kernel trace + stats, then PMC passes each in its own run (no trace domains beside --pmc).
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
ARGS="tools/k4x_one.py mfma"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_k4x_trace -- python3 $ARGS > $OUT/prof_k4x_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/prof_k4x_pmc1 -- python3 $ARGS > $OUT/prof_k4x_pmc1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_k4x_pmc2 -- python3 $ARGS > $OUT/prof_k4x_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_k4x_pmc3 -- python3 $ARGS > $OUT/prof_k4x_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/prof_k4x_pmc4 -- python3 $ARGS > $OUT/prof_k4x_pmc4.log 2>&1
python3 - <<'PY'
import csv, glob, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
for f in glob.glob(os.path.join(out, "prof_k4x_trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print("%-90s calls %s avg %s ns" % (row["Name"][:90], row["Calls"], row["AverageNs"]))
for d in sorted(glob.glob(os.path.join(out, "prof_k4x_pmc*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc, n = {}, {}
        for row in csv.DictReader(open(f)):
            if "hamming_topk_mfma" not in row["Kernel_Name"]: continue
            c = row["Counter_Name"]; acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"]); n.setdefault(c, set()).add(row["Dispatch_Id"])
        for c in sorted(acc): print("%-32s %.6g per launch" % (c, acc[c] / len(n[c])))
PY
