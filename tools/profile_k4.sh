#!/bin/bash
# Profiles bench.py's matcher on the GPU box: kernel trace + stats, then PMC passes (separately, as the
# guide prescribes). Results land in gpurun_out/prof_*; summaries are copied into profiles/ by hand.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
ARGS="bench.py --steps 30 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_trace -- python3 $ARGS > $OUT/prof_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/prof_pmc1 -- python3 $ARGS > $OUT/prof_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_pmc2 -- python3 $ARGS > $OUT/prof_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_pmc3 -- python3 $ARGS > $OUT/prof_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/prof_pmc4 -- python3 $ARGS > $OUT/prof_pmc4.log 2>&1
find $OUT -name "*.csv" | head -40
