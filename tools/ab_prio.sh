# A/B of wave-priority variants of the library on one box: headline + chained (tod_amd/libtodhip_prio{0,1}.so: builds with
# -DTOD_LATENCY_PRIO_LEVEL=0 / 1, not committed; the product's level is 3). The variants are loaded through TODHIP_LIB_PATH.
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
  for which in base prio0 prio1; do
    if [ $which = base ]; then unset TODHIP_LIB_PATH; else export TODHIP_LIB_PATH="$PWD/tod_amd/libtodhip_$which.so"; fi   # never copied over the product file
    timeout -k 10 300 python3 bench.py --extras chained --no-cpu-baseline --repeats 3 --steps 100 > gpurun_out/abp_$which.json 2> gpurun_out/abp_$which.err || { tail -3 gpurun_out/abp_$which.err; exit 1; }
    python3 - gpurun_out/abp_$which.json $which $round <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("%-6s round %s: %.0f frames/s  K4x %.3f ms  %s  chained %.0f" % (sys.argv[2], sys.argv[3], d["value"], d["roofline"]["launch_ms"], {k: round(v, 3) for k, v in d["config"]["stage_ms_per_step"].items()}, d["chained"]["frames_per_s"]["median"]))
PY
  done
done
