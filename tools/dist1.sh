# the sharded step with real RCCL collectives on one rank (TOD_BENCH_FORCE_DIST=1) against the plain single-GPU path, same box
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
for mode in plain dist dist_serial; do
  extra=""; env=""
  [ $mode = dist ] && env="TOD_BENCH_FORCE_DIST=1"
  [ $mode = dist_serial ] && env="TOD_BENCH_FORCE_DIST=1" && extra="--serial-exchange"
  env $env timeout -k 10 300 python3 bench.py --extras= --no-cpu-baseline --repeats 3 $extra $1 > $OUT/d1_$mode.json 2> $OUT/d1_$mode.err || { tail -5 $OUT/d1_$mode.err; exit 1; }
  python3 - $OUT/d1_$mode.json $mode <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print(sys.argv[2], "%.0f frames/s" % d["value"], d["repeats"]["frames_per_s"]["values"], "launch_ms %.3f" % d["roofline"]["launch_ms"], d["config"].get("parallelism"), {k: round(v, 3) for k, v in d.get("stage_ms_per_step", {}).items()} if isinstance(d.get("stage_ms_per_step"), dict) else "")
PY
done
