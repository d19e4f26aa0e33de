#!/bin/bash
# the chained block's shape: frames per step x verifier batches in flight x compute units kept from the matcher
cd "$GRAFT_REPO_ROOT"
for B in ${BS:-16 32}; do for NV in ${NVS:-2 3 4}; do for LC in ${LCS:-0 96}; do
  timeout -k 10 200 python bench.py --extras chained --stages match --no-cpu-baseline --chained-batch $B --chained-workers $NV --chained-latency-cus $LC --steps 40 --repeats 2 > gpurun_out/cs.json 2> gpurun_out/cs.err || { tail -3 gpurun_out/cs.err; exit 1; }
  python3 -c "
import json; d=json.loads(open('gpurun_out/cs.json').read().strip().splitlines()[-1])['chained']
print('B=%2d NV=%d LC=%3d: %7.0f frames/s  step %.2f ms  matcher %.2f ms  stages %s' % ($B,$NV,$LC,d['frames_per_s']['median'],d['ms_per_step'],d['matcher_launch_ms'],{k:round(v,2) for k,v in d['stage_ms_per_step'].items()}))"
done; done; done | tee gpurun_out/chained_shape_sweep.txt
