cd "$GRAFT_REPO_ROOT"
for w in ${1:-4 6 8}; do
  timeout -k 10 300 python bench.py --extras chained --stages match --no-cpu-baseline --verify-workers $w --steps ${2:-60} --repeats 3 > gpurun_out/vw$w.json 2> gpurun_out/vw$w.err
  python3 -c "
import json
d=json.load(open('gpurun_out/vw$w.json'))
c=d['chained']
print('workers $w: chained', c['frames_per_s']['values'], c['stage_ms_per_step'], c['frames_with_the_rendering_pose'])
"
done
