cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for r in 1 2 3 4; do
  rm -rf gpurun_out/prof_l2r$r
  TODHIP_L2_CHUNK_ROUNDS=$r rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l2r$r -- python3 tools/time_l2.py > gpurun_out/prof_l2r$r.log 2>&1
  echo "rounds=$r: $(grep match_l2 gpurun_out/prof_l2r$r.log)"
  python3 - gpurun_out/prof_l2r$r <<'PY'
import csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True))[-1:]:
    for row in csv.DictReader(open(f)):
        if "l2_gemm" in row["Name"] or "rerank" in row["Name"]: print("  %-70s calls %s avg %.1f us" % (row["Name"][:70], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
done
