# rocprofv3 kernel trace of the verifier on one 16-frame data-chained batch (tools/chained_ticks.py's child, 2 repetitions)
cd /tmp && export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/prof_chained_verify"
rm -rf "$OUT"
TODHIP_VERIFY_FLIGHTS=${1:-0} timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o cv -- python3 "$GRAFT_REPO_ROOT/tools/chained_ticks.py" child 16 > "$OUT.log" 2>&1 || { tail -5 "$OUT.log"; exit 1; }
find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$GRAFT_REPO_ROOT/gpurun_out/chained_verify_kernel_stats.csv"
head -25 "$GRAFT_REPO_ROOT/gpurun_out/chained_verify_kernel_stats.csv" | cut -c1-200
