#!/bin/bash
# extra PMC passes on the matcher: scalar-cache behaviour, instruction-cache, SALU/SMEM cycles, branches
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
ARGS="bench.py --steps 10 --warmup 2 --no-cpu-baseline --stages match"
rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES --output-format csv -d $OUT/prof_k4_s1 -- python3 $ARGS > $OUT/prof_k4_s1.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_INST_LEVEL_SMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/prof_k4_s2 -- python3 $ARGS > $OUT/prof_k4_s2.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES --output-format csv -d $OUT/prof_k4_s3 -- python3 $ARGS > $OUT/prof_k4_s3.log 2>&1
python3 - <<'PY'
import csv, glob, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
for d in sorted(glob.glob(os.path.join(out, "prof_k4_s*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc, n = {}, {}
        for row in csv.DictReader(open(f)):
            if "hamming_topk_tiles" not in row["Kernel_Name"]: continue
            c = row["Counter_Name"]; acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"]); n.setdefault(c, set()).add(row["Dispatch_Id"])
        for c in sorted(acc): print("%-32s %.4g per launch" % (c, acc[c] / len(n[c])))
PY
