# bench `chained` (and optionally the headline) with the CU partition: usage cu_partition.sh "<latency cus...>" "<workers...>" [steps]
cd "$GRAFT_REPO_ROOT"
for cus in ${1:-0 8 16 32}; do
  for w in ${2:-2 4}; do
    timeout -k 10 300 python bench.py --extras chained --stages match --no-cpu-baseline --verify-workers $w --chained-workers $w --latency-cus 0 --chained-latency-cus $cus --steps ${3:-60} --repeats 3 > gpurun_out/cu${cus}_w$w.json 2> gpurun_out/cu${cus}_w$w.err || { tail -5 gpurun_out/cu${cus}_w$w.err; exit 1; }
    python3 -c "
import json
d=json.load(open('gpurun_out/cu${cus}_w$w.json'))
c=d['chained']
print('latency cus $cus workers $w: chained', [round(v) for v in c['frames_per_s']['values']], {k: round(v, 2) for k, v in c['stage_ms_per_step'].items()}, 'matcher launch %.3f ms' % c['matcher_launch_ms'])
"
  done
done
