#!/bin/bash
# the matcher's block split (whole / after 2 / after 3 MFMAs / adaptive) on the chained block's DB and on independent bits
cd "$GRAFT_REPO_ROOT"
for HALF in 0 2 3 auto; do
  echo "== TODHIP_K4X_HALF=$HALF"
  if [ $HALF = auto ]; then unset TODHIP_K4X_HALF; export TODHIP_K4X_HALF_DEBUG=1; else export TODHIP_K4X_HALF=$HALF; fi
  timeout -k 10 200 python tools/k4x_on_chained_db.py 2>gpurun_out/ksa.err | grep -o '^[a-z_]* \|"ms_per_launch": [0-9.]*\|"matches": [0-9]*' | paste - - - || { tail -3 gpurun_out/ksa.err; exit 1; }
  grep "K4x blocks" gpurun_out/ksa.err | sort | uniq -c | sort -rn | head -6
done 2>&1 | tee gpurun_out/k4x_split_ab.txt
