#!/bin/bash
# Profiles the matcher of bench.py on the GPU box: kernel trace + stats, then PMC passes each in its own run (no trace domain
# beside --pmc, as the pool requires). Results land in gpurun_out/prof_k4_*; tools/summarize_k4_profile.py digests them into
# profiles/. ENGINE=valu profiles the vector-ALU engine instead.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
ENGINE=${ENGINE:-auto}
ARGS="bench.py --steps 20 --warmup 3 --repeats 1 --no-cpu-baseline --stages match --extras= --engine $ENGINE"
rm -rf $OUT/prof_k4_trace $OUT/prof_k4_pmc1 $OUT/prof_k4_pmc2 $OUT/prof_k4_pmc3 $OUT/prof_k4_pmc4
tools/mfma_fp4_peak > $OUT/mfma_fp4_peak.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_k4_trace -- python3 $ARGS > $OUT/prof_k4_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/prof_k4_pmc1 -- python3 $ARGS > $OUT/prof_k4_pmc1.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d $OUT/prof_k4_pmc2 -- python3 $ARGS > $OUT/prof_k4_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/prof_k4_pmc3 -- python3 $ARGS > $OUT/prof_k4_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/prof_k4_pmc4 -- python3 $ARGS > $OUT/prof_k4_pmc4.log 2>&1
echo done
