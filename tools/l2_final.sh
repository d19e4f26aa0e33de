set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_l2_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu -k "l2 or L2 or sift or float or candidates or gemm or radius_cut or adversarial or integer_valued or c4" 2>&1 | tail -3
bash tools/l2_pmc.sh > gpurun_out/l2_pmc_shipped.txt 2>&1
rm -rf gpurun_out/prof_l2
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_l2 -- python3 tools/time_l2.py > gpurun_out/prof_l2.log 2>&1
grep match_l2 gpurun_out/prof_l2.log
cp $(ls gpurun_out/prof_l2/*/*_kernel_stats.csv | tail -1) gpurun_out/r02_l2_kernel_stats.csv
