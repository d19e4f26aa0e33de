#!/bin/bash
# the matcher's launch knobs on the chained block's DB (tools/k4x_on_chained_db.py): half blocks x bound-exchange period x waves per CU
cd "$GRAFT_REPO_ROOT"
for HALF in ${HALFS:-0 1}; do for SH in ${SHARES:-4 16}; do for WPC in ${WPCS:-16 32 64}; do
  echo -n "half=$HALF share=$SH wpc=$WPC: "
  ONLY_CHAINED=1 TODHIP_K4X_HALF=$HALF TODHIP_K4X_SHARE=$SH TODHIP_K4X_WAVES_PER_CU=$WPC timeout -k 10 200 python tools/k4x_on_chained_db.py 2>gpurun_out/kck.err | grep -o '"ms_per_launch": [0-9.]*' || { tail -3 gpurun_out/kck.err; exit 1; }
done; done; done | tee gpurun_out/k4x_chained_knobs.txt
