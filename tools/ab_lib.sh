# A/B of two builds of the library on one box: tod_amd/libtodhip.so (new) against tod_amd/libtodhip_head.so (a build of HEAD, not committed)
cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
  for which in new head; do
    if [ $which = new ]; then unset TODHIP_LIB_PATH; else export TODHIP_LIB_PATH="$PWD/tod_amd/libtodhip_head.so"; fi   # never copied over the product file
    echo "== $which (round $round)"
    TODHIP_VERIFY_FLIGHTS=0 timeout -k 10 200 python tools/time_verify_batch.py 2>&1 | grep verify_batch || exit 1
    timeout -k 10 200 python tools/verify_ticks.py 2>&1 | grep -i "ms per\|per call" | head -3
    timeout -k 10 300 python tools/eval_phases_chained.py 2 2>&1 | grep "it 0"
  done
done
