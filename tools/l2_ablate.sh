# where the float matcher's GEMM pass spends its time: diagnostics builds of l2.hip without the DMA of the next tiles (1), without
# the LDS reads of the next tile's fragments (2), without both (3) (-DTOD_L2_ABLATE=n; results are garbage) against the real kernel,
# each also with TODHIP_L2_NO_CANDIDATES=1 (no epilogue hits)
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for which in base l2abl1 l2abl2 l2abl3; do   # the diagnostics builds are loaded through TODHIP_LIB_PATH: the product file is never touched
  if [ $which = base ]; then unset TODHIP_LIB_PATH; else export TODHIP_LIB_PATH="$PWD/tod_amd/libtodhip_$which.so"; fi
  for nc in 0 1; do
    rm -rf gpurun_out/prof_l2abl
    if [ $nc = 1 ]; then export TODHIP_L2_NO_CANDIDATES=1; else unset TODHIP_L2_NO_CANDIDATES; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l2abl -- python3 tools/time_l2.py > gpurun_out/prof_l2abl.log 2>&1
    python3 - $which $nc <<'PY'
import csv, glob, sys
for f in sorted(glob.glob("gpurun_out/prof_l2abl/**/*_kernel_stats.csv", recursive=True))[-1:]:
    for row in csv.DictReader(open(f)):
        if "l2_gemm_kernel<2" in row["Name"]: print("%-8s no_candidates=%s: pass 2 %.1f us" % (sys.argv[1], sys.argv[2], float(row["AverageNs"]) / 1e3))
PY
  done
done
