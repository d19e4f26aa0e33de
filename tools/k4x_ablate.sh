# where the matrix-core matcher's time goes: diagnostics builds of match.hip without DB loads (1), without the block test (2),
# without the fp4 expansion (3) (-DTOD_K4X_ABLATE=n, results are garbage) against the real kernel, alone at 32 000 x 1M
cd "$GRAFT_REPO_ROOT"
for which in base abl1 abl2 abl3 base; do   # the diagnostics builds are loaded through TODHIP_LIB_PATH: the product file is never touched
  if [ $which = base ]; then unset TODHIP_LIB_PATH; else export TODHIP_LIB_PATH="$PWD/tod_amd/libtodhip_$which.so"; fi
  echo -n "$which: "; B=32 timeout -k 10 200 python tools/k4x_one.py mfma 2>&1 | tail -1
done
