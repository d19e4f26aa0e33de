#!/usr/bin/env python3
"""bench.py -- frames/s of the detection hot path on MI355X (BASELINE.json metric).

Workload (config C3 of BASELINE.json, also run at N=1 because the 1M-descriptor DB fits one GPU):
one 640x480 synthetic frame = 1000 ORB descriptors matched against the 1M-descriptor object DB
(200 objects x 5000), Hamming brute force k=2, radius 35, then geometric verification.
With N GPUs the descriptor rows are split into N object-aligned shards (strong scaling: the DB is
fixed), every rank matches the frame against its shard, the per-shard candidates are exchanged with
one RCCL all-gather and merged with the order (distance asc, global row asc).

One JSON line on rank 0; see the task contract for the fields. `roofline` is for the dominant kernel
(hamming_topk_tiles); `cpu_baseline` times the CPU oracle on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# 32-bit integer VALU ops (v_xor_b32, v_bcnt_u32_b32) issue at 16 lanes/clk/SIMD on gfx950: 4 cycles per
# wave64 instruction, measured with tools/valu_peak.hip (profiles/r01_valu_peak_microbench.txt: 38-40 T lane-op/s).
# Only f32 FMA is dual-rate, so SURVEY F11's 78.6 T figure does not apply to this kernel.
VALU_PEAK_LANEOPS = 256 * 4 * 16 * 2.4e9            # 39.3 T lane-op/s
LANEOPS_PER_DISTANCE = 16                           # 8 v_xor_b32 + 8 accumulating v_bcnt_u32_b32 per 256-bit pair


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--objects", type=int, default=200, help="objects of 5000 descriptors (200 -> 1M rows)")
    ap.add_argument("--nq", type=int, default=1000)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--radius", type=int, default=35)
    ap.add_argument("--frames", type=int, default=8, help="distinct synthetic frames cycled through")
    ap.add_argument("--stages", default="match,verify", help="comma list of: match,verify")
    ap.add_argument("--iterations", type=int, default=2500, help="n_ransac_iterations (conf/detection.ork:38)")
    ap.add_argument("--min-inliers", type=int, default=8, help="min_inliers (conf/detection.ork:39)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(desc, pts, off, frames, k, radius, budget_s, stages, iterations=2500, min_inliers=8):
    """The CPU oracle (1 thread, the reference has no threads) on as many whole frames as fit the budget."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    spans = O.spans(pts, off)
    t_total, n = 0.0, 0
    for fr in frames:
        t0 = time.perf_counter()
        rc, row_ptr, m, xyz = O.match(desc, off, pts, fr["q_desc"], k, radius)
        if "verify" in stages:
            rng = O.rng_new(1)
            O.verify(fr["kp_xy"], fr["cloud"], row_ptr, m, xyz, spans, min_inliers, iterations, 0.01, rng)
        t_total += time.perf_counter() - t0
        n += 1
        if t_total > budget_s:
            break
    return dict(value=n / t_total, unit="frames/s", cores=1, kind="port",
                sample="%d whole frame(s) of the same workload (%d queries x %d DB rows each), oracle/tod_oracle.cpp, 1 thread"
                       % (n, frames[0]["q_desc"].shape[0], desc.shape[0]))


def main():
    args = parse()
    stages = [s for s in args.stages.split(",") if s]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    from tod_amd import capi, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    desc, pts, off = synth.make_db(args.objects)
    frames = [synth.make_frame(desc, pts, off, args.nq, frame=f, visible_object=(17 * f + 3) % args.objects)
              for f in range(args.frames)]

    stream = torch.cuda.current_stream()
    ctx = capi.Context(local_rank, stream.cuda_stream)
    db_spans = ctx.db_load(desc, pts, off, shard_rank=rank, shard_count=world)
    info = ctx.db_info()

    nq, k = args.nq, args.k
    d_q = [torch.from_numpy(fr["q_desc"]).cuda() for fr in frames]
    d_counts = torch.empty(nq, dtype=torch.int32, device="cuda")
    d_matches = torch.empty((nq * k, 4), dtype=torch.int32, device="cuda")
    d_xyz = torch.empty((nq * k, 3), dtype=torch.float32, device="cuda")
    d_keys = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    d_keys_all = torch.empty((world, nq, k), dtype=torch.int64, device="cuda")
    do_verify = "verify" in stages
    d_kp = [torch.from_numpy(fr["kp_xy"]).cuda() for fr in frames] if do_verify else []
    d_cloud = [torch.from_numpy(fr["cloud"]).cuda() for fr in frames] if do_verify else []
    H, W = frames[0]["cloud"].shape[:2]
    n_pose_total = [0]

    def step(i):
        q = d_q[i % len(d_q)]
        if world == 1:
            ctx.match_device(q.data_ptr(), nq, k, args.radius, d_counts.data_ptr(), d_matches.data_ptr(),
                             d_xyz.data_ptr())
        else:
            ctx.match_shard_device(q.data_ptr(), nq, k, d_keys.data_ptr())
            dist.all_gather_into_tensor(d_keys_all.view(-1), d_keys.view(-1))
            ctx.merge_shards_device(d_keys_all.data_ptr(), world, nq, k, args.radius, d_counts.data_ptr(),
                                    d_matches.data_ptr(), d_xyz.data_ptr())
        if do_verify:
            # every rank verifies the frame it just matched (the merged candidates are identical on all ranks);
            # rand() restarts per frame (decision D4)
            f = i % len(d_q)
            rng = capi.rng_new(1)
            poses = ctx.verify_device(d_kp[f].data_ptr(), nq, d_cloud[f].data_ptr(), H, W, d_counts.data_ptr(),
                                      d_matches.data_ptr(), d_xyz.data_ptr(), k, db_spans, args.min_inliers,
                                      args.iterations, 0.01, rng)
            n_pose_total[0] += len(poses)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    ctx.set_kernel_timing(True)
    c0 = ctx.counters()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    c1 = ctx.counters()
    ctx.set_kernel_timing(False)

    t = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    n_launch = c1.n_match_kernel_launches - c0.n_match_kernel_launches
    k4_ms = (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / max(n_launch, 1)
    alg_bytes = info["shard_rows"] * 32 + nq * 32 + nq * k * 8          # N*32 + Q*32 + Q*k*8
    achieved = alg_bytes / (k4_ms * 1e-3) / 1e9 if k4_ms > 0 else 0.0
    distances = float(nq) * info["shard_rows"]
    valu_frac = LANEOPS_PER_DISTANCE * distances / (k4_ms * 1e-3) / VALU_PEAK_LANEOPS if k4_ms > 0 else 0.0

    if rank == 0:
        out = {
            "metric": "frames/sec @ 640x480, 1M-descriptor DB; achieved HBM GB/s on BF-matcher",
            "value": args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u32 (xor + popcount), f32 in the verifier",
            "data": "synthetic",
            "config": {"workload": "C3: one 640x480 frame = %d ORB descriptors vs %d-descriptor DB (%d objects x 5000), "
                                   "Hamming BF k=%d, radius %d" % (nq, desc.shape[0], args.objects, k, args.radius),
                       "stages": stages, "db_rows_per_gpu": info["shard_rows"],
                       "n_ransac_iterations": args.iterations, "min_inliers": args.min_inliers,
                       "poses_per_frame": n_pose_total[0] / max(args.steps + args.warmup, 1),
                       "parallelism": "db-shard x%d + RCCL all-gather of candidates" % world if world > 1 else "1 GPU"},
            "roofline": {"kernel": "hamming_topk_tiles", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "launch_ms": k4_ms, "algorithmic_bytes": alg_bytes,
                         "note": "at Q=%d queries per DB pass this kernel is bound by integer VALU issue, not HBM "
                                 "(SURVEY F11): see valu_roofline" % nq},
            "valu_roofline": {"bound": "valu", "achieved": LANEOPS_PER_DISTANCE * distances / (k4_ms * 1e-3) / 1e12
                              if k4_ms > 0 else 0.0, "peak": VALU_PEAK_LANEOPS / 1e12, "unit": "T lane-op/s",
                              "frac": valu_frac, "distances_per_launch": distances},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc, pts, off, frames, k, args.radius, args.cpu_seconds, stages,
                                               args.iterations, args.min_inliers)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
