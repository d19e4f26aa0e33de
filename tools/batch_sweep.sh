# headline pipeline by frames per step
cd "$GRAFT_REPO_ROOT"
for b in ${1:-16 24 32 48}; do
  timeout -k 10 300 python3 bench.py --extras= --no-cpu-baseline --repeats 3 --batch $b > gpurun_out/bs_$b.json 2> gpurun_out/bs_$b.err || { tail -3 gpurun_out/bs_$b.err; exit 1; }
  python3 - gpurun_out/bs_$b.json $b <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("batch %s: %.0f frames/s  ms/step %.3f  K4x %.3f ms  frac %.3f  %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["roofline"]["frac"], {k: round(v, 3) for k, v in d["config"]["stage_ms_per_step"].items()}))
PY
done
