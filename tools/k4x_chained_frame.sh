# the matrix-core matcher on ONE data-chained frame (1000 queries, ~1300 DB rows within the radius per query): tiling knobs
cd "$GRAFT_REPO_ROOT"
for cfg in "0 0" "8 32" "8 16" "8 8" "4 32" "4 16" "4 8" "2 32" "2 16" "2 8"; do
  set -- $cfg
  echo -n "QT=$1 waves_per_cu=$2: "; TODHIP_K4X_QT=$1 TODHIP_K4X_WAVES_PER_CU=$2 timeout -k 10 300 python tools/adapter_chained.py 2>&1 | grep "DB pass kernel"
done
