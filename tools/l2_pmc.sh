#!/bin/bash
# PMC passes over the float matcher's GEMM (tools/time_l2.py); TODHIP_L2_NO_CANDIDATES=1 in the environment times the bare pass
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/pmc_l2_*
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS --output-format csv -d $OUT/pmc_l2_1 -- python3 tools/time_l2.py > $OUT/pmc_l2_1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_INSTS_SALU --output-format csv -d $OUT/pmc_l2_2 -- python3 tools/time_l2.py > $OUT/pmc_l2_2.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVES FETCH_SIZE --output-format csv -d $OUT/pmc_l2_3 -- python3 tools/time_l2.py > $OUT/pmc_l2_3.log 2>&1
python3 - <<'PY'
import csv, glob, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out")
for kern in ("l2_gemm_kernel<1,", "l2_gemm_kernel<2,"):
    print(kern)
    for d in sorted(glob.glob(os.path.join(out, "pmc_l2_*"))):
        if not os.path.isdir(d): continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc, n = {}, {}
            for row in csv.DictReader(open(f)):
                if kern not in row["Kernel_Name"].replace(" ", ""): continue
                c = row["Counter_Name"]; acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"]); n.setdefault(c, set()).add(row["Dispatch_Id"])
            for c in sorted(acc): print("  %-28s %.4g per launch" % (c, acc[c] / len(n[c])))
PY
