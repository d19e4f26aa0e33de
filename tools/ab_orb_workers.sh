#!/bin/bash
# ORB batches in flight (1 / 2): the headline and the chained block (32 frames per step)
cd "$GRAFT_REPO_ROOT"
for OW in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --extras chained --chained-batch 32 --chained-latency-cus 0 --orb-workers $OW --steps 100 --repeats 3 > gpurun_out/abow_$OW.json 2> gpurun_out/abow_$OW.err || { tail -3 gpurun_out/abow_$OW.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/abow_$OW.json').read().strip().splitlines()[-1]); c=d['chained']
print('orb workers $OW: headline %.0f frames/s (K4x %.3f ms, stages %s) | chained %.0f (matcher %.2f ms, stages %s)' % (d['value'], d['roofline']['launch_ms'], {k:round(v,2) for k,v in d['config']['stage_ms_per_step'].items()} if 'stage_ms_per_step' in d['config'] else '', c['frames_per_s']['median'], c['matcher_launch_ms'], {k:round(v,2) for k,v in c['stage_ms_per_step'].items()}))"
done | tee gpurun_out/ab_orb_workers.txt
