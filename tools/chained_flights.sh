# bench `chained` by verifier workers x flights (TODHIP_VERIFY_FLIGHTS): usage chained_flights.sh "<workers...>" "<flights...>" [steps]
cd "$GRAFT_REPO_ROOT"
for w in ${1:-3 6}; do
  for f in ${2:-0 3 8}; do
    TODHIP_VERIFY_FLIGHTS=$f timeout -k 10 300 python bench.py --extras chained --stages match --no-cpu-baseline --verify-workers $w --chained-workers $w --steps ${3:-60} --repeats 3 > gpurun_out/vw${w}_f$f.json 2> gpurun_out/vw${w}_f$f.err || { tail -5 gpurun_out/vw${w}_f$f.err; exit 1; }
    python3 -c "
import json
d=json.load(open('gpurun_out/vw${w}_f$f.json'))
c=d['chained']
print('workers $w flights $f: chained', c['frames_per_s']['values'], c['stage_ms_per_step'], c['frames_with_the_rendering_pose'])
"
  done
done
