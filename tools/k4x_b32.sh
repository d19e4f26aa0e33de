# the matrix-core matcher alone at 32 000 queries x 1M rows: tiling (waves per CU) and query blocks per wave
cd "$GRAFT_REPO_ROOT"
for w in ${1:-0 16 24 32 40 48 64}; do
  echo -n "B=32 waves_per_cu=$w: "; B=32 TODHIP_K4X_WAVES_PER_CU=$w timeout -k 10 200 python tools/k4x_one.py mfma 2>&1 | tail -1
done
for qt in 4 6; do
  echo -n "B=32 QT=$qt: "; B=32 TODHIP_K4X_QT=$qt timeout -k 10 200 python tools/k4x_one.py mfma 2>&1 | tail -1
done
echo -n "B=16 QT=8: "; B=16 TODHIP_K4X_QT=8 timeout -k 10 200 python tools/k4x_one.py mfma 2>&1 | tail -1
echo -n "B=16 default: "; B=16 timeout -k 10 200 python tools/k4x_one.py mfma 2>&1 | tail -1
