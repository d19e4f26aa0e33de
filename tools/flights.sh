# The verifier's batch loop with heavy phases on side streams ("flights", TODHIP_VERIFY_FLIGHTS=n; 0 = pure lock-step):
# parity tests, the headline-shaped batch alone, the chained tick log, the chained bench by verifier workers.
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_verify_gpu.py tests/test_end_to_end_gpu.py tests/test_c5_assembled_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -3 || exit 1
timeout -k 10 300 python tools/eval_phases_chained.py 1 2 2>&1 | grep -v Traceback | tail -12
for n in ${1:-0 4 8 16}; do
  echo "== TODHIP_VERIFY_FLIGHTS=$n"
  TODHIP_VERIFY_FLIGHTS=$n timeout -k 10 200 python tools/time_verify_batch.py 2>&1 | grep verify_batch || exit 1
  TODHIP_VERIFY_FLIGHTS=$n CHAINED_TICKS_RAW=gpurun_out/ticks_f$n timeout -k 10 300 python tools/chained_ticks.py 2>&1 | grep "total" || exit 1
done
