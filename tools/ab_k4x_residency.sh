#!/bin/bash
# does the pipeline gain when the matcher's waves leave room (QT = 6: 216 registers, 80 per SIMD stay free) or live shorter (more,
# smaller tiles)? headline + chained (32 frames per step, no CU partition)
cd "$GRAFT_REPO_ROOT"
IFS=","; for CFG in ${CFGS:-8 0,6 0,8 64,6 64,8 128,4 0}; do
  IFS=" "; set -- $CFG
  if [ $1 != 0 ]; then export TODHIP_K4X_QT=$1; else unset TODHIP_K4X_QT; fi; if [ $2 != 0 ]; then export TODHIP_K4X_WAVES_PER_CU=$2; else unset TODHIP_K4X_WAVES_PER_CU; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --extras chained --chained-batch 32 --chained-latency-cus 0 --steps 100 --repeats 3 > gpurun_out/abres.json 2> gpurun_out/abres.err || { tail -3 gpurun_out/abres.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/abres.json').read().strip().splitlines()[-1]); c=d['chained']
print('QT=$1 wpc=$2: headline %.0f frames/s (K4x %.3f ms, stages %s) | chained %.0f (matcher %.2f ms, stages %s)' % (d['value'], d['roofline']['launch_ms'], {k:round(v,2) for k,v in d['config']['stage_ms_per_step'].items()}, c['frames_per_s']['median'], c['matcher_launch_ms'], {k:round(v,2) for k,v in c['stage_ms_per_step'].items()}))"
done | tee gpurun_out/ab_k4x_residency.txt
