# A/B on one box: tod_amd/libtodhip.so (new) against tod_amd/libtodhip_head.so (a build of HEAD, not committed): verifier timings
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
  for which in new head; do
    if [ $which = new ]; then unset TODHIP_LIB_PATH; else export TODHIP_LIB_PATH="$PWD/tod_amd/libtodhip_head.so"; fi   # never copied over the product file
    echo "== $which (round $round)"
    timeout -k 10 100 python tools/verify_ticks.py 2>&1 | tail -1
    timeout -k 10 200 python tools/time_verify_batch.py 2>&1 | grep verify_batch
    timeout -k 10 300 python bench.py --extras chained --stages match --no-cpu-baseline --steps 60 --repeats 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('chained', [round(v) for v in d['chained']['frames_per_s']['values']], {k: round(v, 2) for k, v in d['chained']['stage_ms_per_step'].items()})"
  done
done
