cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
  for which in new head; do
    if [ $which = new ]; then unset TODHIP_LIB_PATH; else export TODHIP_LIB_PATH="$PWD/tod_amd/libtodhip_head.so"; fi   # never copied over the product file
    echo "== $which (round $round)"
    B=32 timeout -k 10 200 python tools/time_verify_batch.py 2>&1 | grep verify_batch
  done
done
