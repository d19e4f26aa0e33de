#!/bin/bash
# Multi-rank rehearsals of bench.py on a ONE-GPU box: RCCL with one rank (the real collectives, no peer), then gloo with
# 2 and 3 ranks sharing device 0 (the whole choreography with real peers, collectives staged through the host).
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
show() { python3 - "$1" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1]
d = json.loads(line)
print("%s: %.0f frames/s, n_gpus %d, %s | step 0 == unsharded: %s | poses/frame %.2f" % (sys.argv[1], d["value"], d["n_gpus"], d["config"]["parallelism"], d["config"]["sharded_step0_equals_unsharded"], d["config"]["poses_per_frame_rank0"]))
PY
}
TOD_BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_dist1.json 2> $OUT/bench_dist1.err && show $OUT/bench_dist1.json || { tail -5 $OUT/bench_dist1.err; exit 1; }
TOD_BENCH_BACKEND=gloo TOD_BENCH_ONE_DEVICE=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err && show $OUT/bench_gloo2.json || { tail -15 $OUT/bench_gloo2.err; exit 1; }
TOD_BENCH_BACKEND=gloo TOD_BENCH_ONE_DEVICE=1 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 3 --steps 4 --warmup 2 --no-cpu-baseline --exchange all_gather --serial-exchange > $OUT/bench_gloo3.json 2> $OUT/bench_gloo3.err && show $OUT/bench_gloo3.json || { tail -15 $OUT/bench_gloo3.err; exit 1; }
TOD_BENCH_BACKEND=gloo TOD_BENCH_ONE_DEVICE=1 timeout -k 10 400 python3 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --replicas > $OUT/bench_spawn2.json 2> $OUT/bench_spawn2.err && show $OUT/bench_spawn2.json || { tail -15 $OUT/bench_spawn2.err; exit 1; }
# four ranks (the per-rank launch shape of a 4-GPU job: 128 000 queries x a 250k-row shard), gloo, all on device 0
TOD_BENCH_BACKEND=gloo TOD_BENCH_ONE_DEVICE=1 timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 4 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_gloo4.json 2> $OUT/bench_gloo4.err && show $OUT/bench_gloo4.json || { tail -15 $OUT/bench_gloo4.err; exit 1; }
