# bench `configs` (C1, C2, C5 share, C4) by verifier workers in flight: usage c5_workers.sh "<workers...>"
cd "$GRAFT_REPO_ROOT"
for w in ${1:-2 4 6}; do
  timeout -k 10 400 python bench.py --extras configs --stages match --no-cpu-baseline --verify-workers $w --steps 40 --repeats 2 > gpurun_out/c5w$w.json 2> gpurun_out/c5w$w.err || { tail -5 gpurun_out/c5w$w.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/c5w$w.json'))
c=d['configs']['C5_single_gpu_share']
print('workers $w: C5 share', c['frames_per_s'], c.get('stage_ms_per_step'))
print('   C1', d['configs']['C1']['frames_per_s'] if 'frames_per_s' in d['configs']['C1'] else list(d['configs']['C1'].keys()))
"
done
