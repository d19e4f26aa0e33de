cd "$GRAFT_REPO_ROOT"
cp tod_amd/libtodhip.so /tmp/new.so
for round in 1 2; do
  for which in new head; do
    if [ $which = new ]; then cp /tmp/new.so tod_amd/libtodhip.so; else cp tod_amd/libtodhip_head.so tod_amd/libtodhip.so; fi
    echo "== $which (round $round)"
    B=32 timeout -k 10 200 python tools/time_verify_batch.py 2>&1 | grep verify_batch
  done
done
cp /tmp/new.so tod_amd/libtodhip.so
