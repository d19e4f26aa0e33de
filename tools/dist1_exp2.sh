cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
run() { name=$1; shift; env "$@" > $OUT/dx_$name.json 2> $OUT/dx_$name.err || { tail -5 $OUT/dx_$name.err; return; }
  python3 - $OUT/dx_$name.json $name <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("%-28s %.0f frames/s  ms/step %.3f  K4x %.3f  %s" % (sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d["config"].get("stage_ms_per_step")))
PY
}
B="timeout -k 10 300 python3 bench.py --extras= --no-cpu-baseline --repeats 2"
run plain_vw2 X=1 $B
run plain_vw1 TOD_BENCH_HEADLINE_VW=1 $B
run dist_vw2 TOD_BENCH_FORCE_DIST=1 $B
run dist_vw1 TOD_BENCH_FORCE_DIST=1 TOD_BENCH_HEADLINE_VW=1 $B
run dist_serial_vw1 TOD_BENCH_FORCE_DIST=1 TOD_BENCH_HEADLINE_VW=1 $B --serial-exchange
