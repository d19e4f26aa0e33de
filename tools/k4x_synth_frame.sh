cd "$GRAFT_REPO_ROOT"
for cfg in "0 0" "8 8" "4 16" "4 8" "4 4" "2 16" "2 8" "2 4"; do
  set -- $cfg
  echo -n "synthetic 1 frame QT=$1 waves_per_cu=$2: "; B=1 TODHIP_K4X_QT=$1 TODHIP_K4X_WAVES_PER_CU=$2 timeout -k 10 200 python tools/k4x_one.py mfma 2>&1 | tail -1
done
for cfg in "4 4" "2 4" "2 2"; do
  set -- $cfg
  echo -n "chained QT=$1 waves_per_cu=$2: "; TODHIP_K4X_QT=$1 TODHIP_K4X_WAVES_PER_CU=$2 timeout -k 10 300 python tools/adapter_chained.py 2>&1 | grep "DB pass kernel"
done
