cd "$GRAFT_REPO_ROOT"
for b in ${1:-16 32}; do
 for vw in 1 2; do
  TOD_BENCH_FORCE_DIST=1 TOD_BENCH_HEADLINE_VW=$vw timeout -k 10 300 python3 bench.py --extras= --no-cpu-baseline --repeats 3 --batch $b > gpurun_out/bsd_$b.json 2> gpurun_out/bsd_$b.err || { tail -3 gpurun_out/bsd_$b.err; exit 1; }
  python3 - gpurun_out/bsd_$b.json $b $vw <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("dist batch %s vw %s: %.0f frames/s  ms/step %.3f  K4x %.3f ms  %s" % (sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], {k: round(v, 3) for k, v in d["config"]["stage_ms_per_step"].items()}))
PY
 done
done
