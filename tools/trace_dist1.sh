# kernel trace (start/end per kernel) of the sharded step with RCCL on one rank: what runs between two DB passes
cd /tmp && export TMPDIR=/tmp
OUT="$GRAFT_REPO_ROOT/gpurun_out/trace_dist1"
rm -rf "$OUT"
[ "${MODE:-dist}" = dist ] && export TOD_BENCH_FORCE_DIST=1
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- python3 "$GRAFT_REPO_ROOT/bench.py" --extras= --no-cpu-baseline --repeats 1 --steps 30 --warmup 5 $1 > "$OUT.log" 2>&1 || { tail -5 "$OUT.log"; exit 1; }
find "$OUT" -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} "$GRAFT_REPO_ROOT/gpurun_out/${MODE:-dist}1_kernel_trace.csv"
ls -la "$GRAFT_REPO_ROOT/gpurun_out/${MODE:-dist}1_kernel_trace.csv"
