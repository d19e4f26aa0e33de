#!/bin/bash
# runs the default bench.py N times, each under its own timeout, and shows where a run that did not finish stopped
cd "$GRAFT_REPO_ROOT"
N=${1:-8}
for i in $(seq 1 $N); do
  timeout -k 5 ${2:-200} python bench.py ${3:-} > gpurun_out/soak_$i.json 2> gpurun_out/soak_$i.err
  rc=$?
  echo "run $i rc=$rc: $(grep 'bench.py \[' gpurun_out/soak_$i.err | tail -1)"
  if [ $rc -ne 0 ]; then echo "  --- stopped here; stderr tail:"; tail -5 gpurun_out/soak_$i.err; break; fi
done
