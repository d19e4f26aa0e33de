cd "$GRAFT_REPO_ROOT"
for A in "--steps 200" "--steps 100" "--steps 200 --warmup 30" "--steps 100"; do
  python bench.py --no-cpu-baseline --extras= $A > gpurun_out/ov.json 2>/dev/null
  python3 -c "
import json; d=json.loads(open('gpurun_out/ov.json').read().strip().splitlines()[-1])
print('$A: %.0f frames/s, K4x %.3f, stages %s, spread %s' % (d['value'], d['roofline']['launch_ms'], {k:round(v,2) for k,v in d['config']['stage_ms_per_step'].items()}, [round(x) for x in d['repeats']['values']] if 'repeats' in d and 'values' in d['repeats'] else ''))"
done
