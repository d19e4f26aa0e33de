# headline pipeline with the CU partition (todhip_set_cu_partition): usage cu_partition_headline.sh "<latency cus...>" [steps]
cd "$GRAFT_REPO_ROOT"
for cus in ${1:-0 16 32 64}; do
  timeout -k 10 300 python bench.py --extras "" --no-cpu-baseline --latency-cus $cus --steps ${2:-100} --repeats 3 > gpurun_out/hcu${cus}.json 2> gpurun_out/hcu${cus}.err || { tail -5 gpurun_out/hcu${cus}.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/hcu${cus}.json'))
print('latency cus $cus: headline %.0f frames/s, %.3f ms per step, matcher launch %.3f ms, roofline frac %.3f' % (d['value'], d['ms_per_step'], d['roofline'].get('launch_ms', 0), d['roofline']['frac']), d.get('repeats', {}).get('values'))
"
done
