#!/usr/bin/env python3
"""bench.py -- frames/s of the detection hot path on MI355X (BASELINE.json metric).

Workload (config C3 of BASELINE.json, also run at N=1 because the 1M-descriptor DB fits one GPU):
one 640x480 synthetic frame = 1000 ORB descriptors matched against the 1M-descriptor object DB
(200 objects x 5000), Hamming brute force k=2, radius 35, then geometric verification.
With N GPUs (tod_amd/sharded.py) the descriptor rows are split into N object-aligned shards and a step
processes 16 frames per rank: descriptors are all-gathered, every rank matches all N x 16 frames against
its shard, the per-shard candidates are exchanged with one RCCL collective (all-to-all by default: a rank only needs
the candidates of its own frames; --exchange all_gather for the literal all-gather), and every rank merges (order:
distance asc, global row asc) and verifies its own frames. Per-GPU work is constant as N grows. The collectives and
the merge run on their own stream, double buffered, so that they overlap the DB passes of the neighbouring steps
(--serial-exchange puts them back on the matcher's stream in program order).

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

# The HIP runtime maps a process's streams onto 4 hardware queues unless told otherwise, so at most ~4 of the
# per-frame ORB/verifier streams would overlap (tools/stream_concurrency.py: 4 by default, 8 with more queues;
# beyond 8 busy queues the driver time-slices them and every launch stalls, so 8 it is).
# Must be in the environment before the runtime is loaded (i.e. before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0                               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# 32-bit integer VALU ops (v_xor_b32, v_bcnt_u32_b32) issue at 16 lanes/clk/SIMD on gfx950: 4 cycles per
# wave64 instruction, measured with tools/valu_peak.hip (profiles/r01_valu_peak_microbench.txt: 38-40 T lane-op/s).
# Only f32 FMA is dual-rate, so SURVEY F11's 78.6 T figure does not apply to this kernel.
VALU_PEAK_LANEOPS = 39.81e12                        # measured: xor->bcnt mix at 8 waves/SIMD (256 CU x 4 SIMD x 16 lanes x ~2.43 GHz)
LANEOPS_DENSE = 16                                  # 8 v_xor_b32 + 8 accumulating v_bcnt_u32_b32 per full 256-bit pair


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--objects", type=int, default=200, help="objects of 5000 descriptors (200 -> 1M rows)")
    ap.add_argument("--nq", type=int, default=1000)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--radius", type=int, default=35)
    ap.add_argument("--frames", type=int, default=8, help="distinct synthetic frames cycled through")
    ap.add_argument("--stages", default="orb,match,verify", help="comma list of: orb,match,verify")
    ap.add_argument("--batch", type=int, default=16, help="frames per rank per step")
    ap.add_argument("--exchange", choices=("all_to_all", "all_gather"), default="all_to_all",
                    help="several ranks: how the per-shard candidates travel. A rank only merges its own frames, so an "
                         "all-to-all moves 1/world of an all-gather's bytes over the point-to-point xGMI links")
    ap.add_argument("--replicas", action="store_true",
                    help="several ranks: every rank holds the WHOLE DB and matches only its own frames -- no data-path collective "
                         "(SURVEY 8(e)'s comparison point for DBs that fit one GPU; the default shards the DB rows)")
    ap.add_argument("--serial-exchange", action="store_true",
                    help="several ranks: issue the collectives on the matcher's stream, in program order (gather -> match -> "
                         "exchange -> merge), instead of on their own stream where they overlap the neighbouring DB passes")
    ap.add_argument("--matcher-contexts", type=int, default=1,
                    help="single device only: 2 alternates steps between two matcher contexts so that consecutive DB passes "
                         "overlap (+4 %% frames/s); off by default because a launch's own duration then no longer says what "
                         "it costs, and a kernel-trace profile (which serializes launches) no longer agrees with it")
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
        if "orb" in stages:
            O.orb(fr["image"], frames[0]["q_desc"].shape[0], 3, 1.2)
        rc, row_ptr, m, xyz = O.match(desc, off, pts, fr["q_desc"], k, radius)
        if "verify" in stages:
            rng = O.rng_new(1)
            O.verify(fr["kp_xy"], fr["cloud"], row_ptr, m, xyz, spans, min_inliers, iterations, 0.01, rng)
        t_total += time.perf_counter() - t0
        n += 1
        if t_total > budget_s:
            break
    out = dict(value=n / t_total, unit="frames/s", cores=1, kind="port",
               sample="%d whole frame(s) of the same workload, stages %s (%d queries x %d DB rows each), oracle/*.c*, 1 thread"
                      % (n, "+".join(stages), frames[0]["q_desc"].shape[0], desc.shape[0]))
    # the same port over all host cores, frames in parallel (the reference itself is single-threaded; the
    # foreign calls release the GIL). One frame per thread, so the sample costs about one single-thread frame of wall time.
    from concurrent.futures import ThreadPoolExecutor
    n_thr = max(1, min(os.cpu_count() or 1, 16))

    def one(i):
        fr = frames[i % len(frames)]
        if "orb" in stages:
            O.orb(fr["image"], frames[0]["q_desc"].shape[0], 3, 1.2)
        rc, row_ptr, m, xyz = O.match(desc, off, pts, fr["q_desc"], k, radius)
        if "verify" in stages:
            O.verify(fr["kp_xy"], fr["cloud"], row_ptr, m, xyz, spans, min_inliers, iterations, 0.01, O.rng_new(1))

    t0 = time.perf_counter()
    with ThreadPoolExecutor(n_thr) as pool:
        list(pool.map(one, range(n_thr)))
    out["all_cores"] = dict(value=n_thr / (time.perf_counter() - t0), unit="frames/s", cores=n_thr,
                            sample="%d frames, one per thread" % n_thr)
    return out


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
    # rehearsal on a 1-GPU box: TOD_BENCH_BACKEND=gloo TOD_BENCH_ONE_DEVICE=1 puts every rank on device 0
    one_device = os.environ.get("TOD_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("TOD_BENCH_BACKEND", "nccl")
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # TOD_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, collectives, shard + merge) with one rank --
    # the only way to exercise the RCCL calls on a 1-GPU box
    use_dist = world > 1 or os.environ.get("TOD_BENCH_FORCE_DIST") == "1"
    if use_dist:
        for key, val in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(key, val)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    desc, pts, off = synth.make_db(args.objects)
    frames = [synth.make_frame(desc, pts, off, args.nq, frame=f, visible_object=(17 * f + 3) % args.objects)
              for f in range(args.frames)]
    for f, fr in enumerate(frames):
        fr["image"] = synth.make_image(f)

    # one explicit stream for everything: libtodhip kernels, torch copies and the RCCL collectives (which order
    # themselves against torch's current stream). The default stream's handle is 0 == "create your own" for
    # todhip_create, which would put the kernels on a different stream than the collectives.
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    ctx = capi.Context(local_rank, stream.cuda_stream)
    sharded = world > 1 and not args.replicas
    db_spans = ctx.db_load(desc, pts, off, shard_rank=rank if sharded else 0, shard_count=world if sharded else 1)
    info = ctx.db_info()
    # single device: steps alternate between two matcher contexts (own stream, own workspaces, own copy of the 32 MB
    # DB), so the merge/finalize tail and the launch ramp of one step overlap with the DB pass of the next; with
    # several ranks the collectives keep everything on the one stream
    mctx, mstreams = [ctx], [stream]
    if world == 1 and args.matcher_contexts > 1:
        s2 = torch.cuda.Stream()
        c2 = capi.Context(local_rank, s2.cuda_stream)
        c2.db_load(desc, pts, off)
        mctx.append(c2); mstreams.append(s2)

    nq, k = args.nq, args.k
    do_verify = "verify" in stages
    do_orb = "orb" in stages
    B = args.batch                                   # frames per rank per step
    from concurrent.futures import ThreadPoolExecutor
    from math import gcd
    # frame f belongs to rank (f % world); every rank keeps its own frames resident in HBM
    my_ids = [f for f in range(len(frames)) if f % world == rank] or [rank % len(frames)]
    n_my = len(my_ids)
    H, W = frames[0]["cloud"].shape[:2]
    # A step is a batch of B frames: frame (i * B + b) % n_my of this rank's frames, b = 0..B-1. The batches repeat
    # with a short period, so each distinct batch is laid out once as contiguous [B, ...] arrays (what a camera
    # driver handing over B frames would provide).
    period = n_my // gcd(B, n_my)

    def batch_of(v, key, dtype=None):
        arrs = [frames[my_ids[(v * B + b) % n_my]][key] for b in range(B)]
        return torch.from_numpy(np.ascontiguousarray(np.stack(arrs), dtype=dtype)).cuda()

    Q_B = [batch_of(v, "q_desc") for v in range(period)]                            # [B, Q, 32]
    KP_B = [batch_of(v, "kp_xy", np.float32) for v in range(period)]                # [B, Q, 2]
    CLOUD_B = [batch_of(v, "cloud", np.float32) for v in range(period)] if do_verify else []   # [B, H, W, 3]
    # stage A runs on the SURVEY 8(d) synthetic grey image of the frame. Its descriptors are not the matcher's input
    # (random-image ORB descriptors cannot match a synthetic DB; the frame's planted descriptors do that), but its
    # work is part of every frame and the matcher of a step starts only when that step's ORB batch is done.
    IMG_B = [batch_of(v, "image") for v in range(period)] if do_orb else []         # [B, H, W]
    # matcher outputs, ring of D step buffers: the verifier of step s reads set (s % D) while later steps fill the others
    D = 3
    outs = [dict(counts=torch.empty(B * nq, dtype=torch.int32, device="cuda"),
                 matches=torch.empty((B * nq * k, 4), dtype=torch.int32, device="cuda"),
                 xyz=torch.empty((B * nq * k, 3), dtype=torch.float32, device="cuda")) for _ in range(D)]
    d_keys = torch.empty((world * B * nq, k), dtype=torch.int64, device="cuda")
    # Three stages, three streams, three host threads: ORB batch (step s+1..s+2) | matcher (step s, this thread: it owns
    # torch's current stream and the collectives) | verifier batch (step s-1). Every stage call covers the B frames of
    # a step in the launches of one frame (todhip_orb_batch_device, todhip_verify_batch_device).
    ostream, vstream = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)
    octx = capi.Context(local_rank, ostream.cuda_stream) if do_orb else None
    vctx = capi.Context(local_rank, vstream.cuda_stream) if do_verify else None
    orb_out = (torch.empty((B, nq, 2), device="cuda"), torch.empty((B, nq, 4), device="cuda"),
               torch.empty((B, nq, 32), dtype=torch.uint8, device="cuda")) if do_orb else None
    opool = ThreadPoolExecutor(1) if do_orb else None
    vpool = ThreadPoolExecutor(1) if do_verify else None
    stage_s = {"orb": 0.0, "match_issue": 0.0, "verify": 0.0}
    n_kp_total = [0]
    n_pose_total = [0]
    n_steps_done = [0]

    def alloc(shape, dtype_name):
        return torch.empty(shape, dtype=getattr(torch, dtype_name), device="cuda")

    def all_gather(out, inp):
        if backend == "nccl":
            dist.all_gather_into_tensor(out.view(-1), inp.contiguous().view(-1))
        else:                                           # gloo rehearsal: stage through the host
            parts = [torch.empty(inp.numel(), dtype=inp.dtype) for _ in range(world)]
            dist.all_gather(parts, inp.contiguous().view(-1).cpu())
            out.view(-1).copy_(torch.cat(parts).to(out.device))

    def all_to_all(out, inp):
        if backend == "nccl":
            dist.all_to_all_single(out.view(-1), inp.contiguous().view(-1))
        else:                                           # gloo rehearsal: stage through the host
            o = torch.empty(out.numel(), dtype=out.dtype)
            dist.all_to_all_single(o, inp.contiguous().view(-1).cpu())
            out.view(-1).copy_(o.to(out.device))

    def orb_task(i):
        t = time.perf_counter()
        n = octx.orb_batch_device(IMG_B[i % period].data_ptr(), B, H * W, H, W, W, nq, 3, 1.2, orb_out[0].data_ptr(),
                                  orb_out[1].data_ptr(), orb_out[2].data_ptr(), nq)
        stage_s["orb"] += time.perf_counter() - t
        return sum(n)

    def verify_task(i, ev):
        vstream.wait_event(ev)                          # this step's matcher outputs (recorded on the matcher's stream)
        ev.synchronize()                                # (host side too, so that the stage time below is the verifier's own)
        t = time.perf_counter()
        o = outs[i % D]
        rngs = (capi.Rng * B)(*[capi.rng_new(1) for _ in range(B)])   # rand() restarts per frame (decision D4)
        poses = vctx.verify_batch_device(B, KP_B[i % period].data_ptr(), nq, CLOUD_B[i % period].data_ptr(), H, W,
                                         o["counts"].data_ptr(), o["matches"].data_ptr(), o["xyz"].data_ptr(), k, db_spans,
                                         args.min_inliers, args.iterations, 0.01, rngs)
        stage_s["verify"] += time.perf_counter() - t
        return sum(len(p) for p in poses)

    # Several ranks: the collectives and the merge run on their own stream (cstream), so that the DB pass of step i + 1
    # follows that of step i without a gap. Order on cstream, identical on every rank: gather(0), gather(1), exchange(0),
    # merge(0), gather(2), exchange(1), merge(1), ... -- one communicator, one stream, one order. Double-buffered
    # q_all / keys / km; the events below are the only cross-stream edges:
    #   gathered(i) -> match(i);  matched(i) -> exchange(i);  exchanged(i - 2) -> match(i) (keys buffer reuse);
    #   gather(i + 2) overwrites q_all[i % 2] after exchange(i), which itself waited for match(i)  (stream order).
    overlap = use_dist and not args.serial_exchange and not args.replicas
    if overlap:
        cstream = torch.cuda.Stream(priority=-1)
        cctx = capi.Context(local_rank, cstream.cuda_stream)
        cctx.db_load(desc, pts, off, shard_rank=rank, shard_count=world)
        q_all2 = [torch.empty((world, B, nq, 32), dtype=torch.uint8, device="cuda") for _ in range(2)]
        keys2 = [torch.empty((world * B * nq, k), dtype=torch.int64, device="cuda") for _ in range(2)]
        km2 = [torch.empty((world, B * nq, k), dtype=torch.int64, device="cuda") for _ in range(2)]
        keys_all2 = ([torch.empty((world, world, B * nq, k), dtype=torch.int64, device="cuda") for _ in range(2)]
                     if args.exchange == "all_gather" else None)
        ev_gathered, ev_exchanged = {}, {}

    def issue_gather(i):
        with torch.cuda.stream(cstream):
            all_gather(q_all2[i % 2], Q_B[i % period])
            ev_gathered[i] = torch.cuda.Event()
            ev_gathered[i].record(cstream)

    def match_step_overlapped(i, n_steps):
        o = outs[i % D]
        if i == 0:
            ev_gathered.clear(); ev_exchanged.clear()
            issue_gather(0)
        if i + 1 < n_steps:
            issue_gather(i + 1)
        stream.wait_event(ev_gathered.pop(i))
        if i - 2 in ev_exchanged:
            stream.wait_event(ev_exchanged.pop(i - 2))
        ctx.match_shard_device(q_all2[i % 2].data_ptr(), world * B * nq, k, args.radius, keys2[i % 2].data_ptr())
        matched = torch.cuda.Event()
        matched.record(stream)
        with torch.cuda.stream(cstream):
            cstream.wait_event(matched)
            if args.exchange == "all_to_all":
                all_to_all(km2[i % 2], keys2[i % 2])                               # keys is [frame owner][B*Q][k]
                mine = km2[i % 2]
            else:
                all_gather(keys_all2[i % 2], keys2[i % 2])                         # [shard][rank][B*Q][k]
                mine = km2[i % 2]
                mine.copy_(keys_all2[i % 2][:, rank])
            ev_exchanged[i] = torch.cuda.Event()
            ev_exchanged[i].record(cstream)
            cctx.merge_shards_device(mine.data_ptr(), world, B * nq, k, args.radius, o["counts"].data_ptr(),
                                     o["matches"].data_ptr(), o["xyz"].data_ptr())
        return cstream

    def match_step(i, n_steps):
        """Returns the stream on which this step's matcher outputs become complete."""
        if overlap:
            return match_step_overlapped(i, n_steps)
        o = outs[i % D]
        q = Q_B[i % period]
        if not use_dist or args.replicas:
            # single device (or a replica of the whole DB): no key exchange; the B frames' descriptors share one pass over the DB
            mctx[i % len(mctx)].match_device(q.data_ptr(), B * nq, k, args.radius, o["counts"].data_ptr(),
                                             o["matches"].data_ptr(), o["xyz"].data_ptr())
            return mstreams[i % len(mstreams)]
        # tod_amd/sharded.py with B frames per rank, in program order on the matcher's stream: gather descriptors, match
        # all world*B frames against this rank's shard, exchange the candidates, merge this rank's B frames
        q_all = alloc((world, B, nq, 32), "uint8")
        all_gather(q_all, q)
        ctx.match_shard_device(q_all.data_ptr(), world * B * nq, k, args.radius, d_keys.data_ptr())
        if args.exchange == "all_to_all":
            km = alloc((world, B * nq, k), "int64")                             # chunk j <- shard j's keys of MY frames
            all_to_all(km, d_keys)                                              # d_keys is [frame owner][B*Q][k]
        else:
            keys_all = alloc((world, world, B, nq, k), "int64")                 # [shard][rank][b][Q][k]
            all_gather(keys_all, d_keys)
            km = keys_all[:, rank].contiguous()                                 # [shard][B*Q][k]
        ctx.merge_shards_device(km.data_ptr(), world, B * nq, k, args.radius, o["counts"].data_ptr(),
                                o["matches"].data_ptr(), o["xyz"].data_ptr())
        return stream

    def run_steps(n_steps):
        """ORB(i) -> match(i) -> verify(i); ORB runs up to D steps ahead, the matcher up to D - 1 steps ahead of the verifier."""
        ofut, vfut = {}, {}
        for j in range(min(D, n_steps)):
            if do_orb:
                ofut[j] = opool.submit(orb_task, j)
        def orb_done(j):                                              # wait for ORB batch j, keep ORB D batches ahead
            if do_orb and j in ofut:
                n_kp_total[0] += ofut.pop(j).result()
                if j + D < n_steps:
                    ofut[j + D] = opool.submit(orb_task, j + D)

        for i in range(n_steps):
            if do_verify and i - D in vfut:
                n_pose_total[0] += vfut.pop(i - D).result()          # buffer set i % D is free again
            orb_done(i)
            if overlap:
                orb_done(i + 1)                                       # the gather of step i + 1 is issued in step i
            t = time.perf_counter()
            out_stream = match_step(i, n_steps)
            stage_s["match_issue"] += time.perf_counter() - t
            if do_verify:
                ev = torch.cuda.Event()
                ev.record(out_stream)                    # the matcher outputs of this step are complete after this
                vfut[i] = vpool.submit(verify_task, i, ev)
        for i in sorted(vfut):
            n_pose_total[0] += vfut[i].result()
        n_steps_done[0] += n_steps

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    fence()
    for key in stage_s:
        stage_s[key] = 0.0
    for c in mctx:
        c.set_kernel_timing(True)
    c0 = [c.counters() for c in mctx]
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    dt_local = dt
    c1 = [c.counters() for c in mctx]
    # outside the timed region, single device only: the same launch with the radius cut switched off (radius 256) --
    # what the DB pass costs when no lower bound can prune (correlated descriptors); reported beside the roofline
    dense_ms = None
    if world == 1:
        o = outs[0]
        d0 = ctx.counters()
        for _ in range(3):
            ctx.match_device(Q_B[0].data_ptr(), B * nq, k, 256, o["counts"].data_ptr(), o["matches"].data_ptr(), o["xyz"].data_ptr())
        torch.cuda.synchronize()
        d1 = ctx.counters()
        dense_ms = (d1.sum_match_kernel_ms - d0.sum_match_kernel_ms) / max(d1.n_match_kernel_launches - d0.n_match_kernel_launches, 1)
    for c in mctx:
        c.set_kernel_timing(False)

    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

    n_launch = sum(b.n_match_kernel_launches - a.n_match_kernel_launches for a, b in zip(c0, c1))
    k4_ms = sum(b.sum_match_kernel_ms - a.sum_match_kernel_ms for a, b in zip(c0, c1)) / max(n_launch, 1)
    frames_per_launch = B                                          # one launch matches the world*B frames of a step
    if args.replicas:
        world_q = 1                                                # ... or, with replicas, this rank's own B frames
    else:
        world_q = world
    alg_bytes = info["shard_rows"] * 32 + world_q * frames_per_launch * nq * (32 + k * 8)   # N*32 + F*Q*(32 + k*8)
    achieved = alg_bytes / (k4_ms * 1e-3) / 1e9 if k4_ms > 0 else 0.0
    distances = float(nq) * world_q * frames_per_launch * info["shard_rows"]

    # HBM traffic of the dominant kernel from the PMC passes (FETCH_SIZE + WRITE_SIZE, collected in their own
    # rocprofv3 runs by tools/profile_k4.sh and committed as profiles/r01_k4_pmc.json); only quoted for the
    # workload it was measured on
    traffic, traffic_src = None, None
    # VALU instructions the kernel EXECUTES per (row, query) pair: 16 for a full distance, about half of that when the
    # 128-bit lower bound prunes the row (data dependent). Taken from the same PMC passes (SQ_INSTS_VALU) when the
    # workload is the profiled one, else the dense count is used and the figure is an upper bound.
    laneops, laneops_src = float(LANEOPS_DENSE), "dense instruction count (upper bound)"
    pmc_path = os.path.join(ROOT, "profiles", "r01_k4_pmc.json")
    pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
    if (world == 1 and k == 2 and info["shard_rows"] == 1000000 and args.radius == 35 and
            pmc.get("queries_per_launch") == B * nq):
        traffic = pmc["hbm_traffic_bytes_per_launch"]
        traffic_src = "profiles/r01_k4_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        if "valu_insts_per_row_and_wave" in pmc:
            laneops = float(pmc["valu_insts_per_row_and_wave"])
            laneops_src = "profiles/r01_k4_pmc.json (SQ_INSTS_VALU per row and 64-query wave)"
    # Single device: consecutive steps' launches overlap (two matcher contexts), so a launch's own duration overstates
    # its cost; the VALU rate is therefore taken chip-wide over the timed region (all launches' executed lane-ops / wall
    # time), and the per-launch figure is given beside it.
    valu_rate_launch = laneops * distances / (k4_ms * 1e-3) if k4_ms > 0 else 0.0
    valu_rate = laneops * distances * n_launch / dt_local if dt_local > 0 else 0.0
    valu_frac = valu_rate / VALU_PEAK_LANEOPS
    if rank == 0:
        out = {
            "metric": "frames/sec @ 640x480, 1M-descriptor DB; achieved HBM GB/s on BF-matcher",
            "value": args.steps * world * B / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32 (xor + popcount), f32 in the verifier",
            "data": "synthetic",
            "config": {"workload": "C3: one 640x480 frame = %d ORB descriptors vs %d-descriptor DB (%d objects x 5000), "
                                   "Hamming BF k=%d, radius %d" % (nq, desc.shape[0], args.objects, k, args.radius),
                       "stages": stages, "db_rows_per_gpu": info["shard_rows"],
                       "matcher_input": "SURVEY 8(d) synthetic descriptors (iid bits, planted matches at 8 % flips): the "
                                        "partial-distance elimination halves the DB pass on them; it is data dependent -- 0.27 ms per "
                                        "frame at rBRIEF-like correlation, 0.36 ms on strongly biased descriptors, 0.43 ms dense (DESIGN.md 6, "
                                        "tools/k4_on_correlated_descriptors.py, tools/k4_on_orb_descriptors.py)",
                       "n_ransac_iterations": args.iterations, "min_inliers": args.min_inliers,
                       "poses_per_frame_rank0": n_pose_total[0] / max(n_steps_done[0] * B, 1),
                       "frames_per_rank_per_step": B,
                       "pipeline": "3 stages on 3 streams, each one batched call per step: ORB | matcher | verifier" +
                                   ("; collectives + merge on a 4th stream, overlapping the neighbouring DB passes" if overlap else ""),
                       "stage_ms_per_step": {key: 1e3 * v / max(args.steps, 1) for key, v in stage_s.items()},
                       "orb": "ORB-%d, 3 levels, scale 1.2 on the 8(d) synthetic image; %.0f keypoints/frame" %
                              (args.nq, n_kp_total[0] / max(n_steps_done[0] * B, 1)) if do_orb else None,
                       "frames_per_step": world * B,
                       "parallelism": ("DB rows sharded x%d (object aligned), %d frames per rank per step, RCCL all-gather of "
                                       "descriptors, %s of per-shard candidates" % (world, B, args.exchange)) if sharded else
                                      ("%d replicas of the whole DB, %d frames per rank per step, no data-path collective" % (world, B)
                                       if world > 1 else "1 GPU")},
            "roofline": {"kernel": "hamming_topk_tiles", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "launch_ms": k4_ms, "algorithmic_bytes": alg_bytes,
                         "binding_roof": "integer VALU (see valu_roofline)", "valu_frac": valu_frac,
                         "dense_launch_ms": dense_ms,
                         "dense_launch_note": "same launch without the radius bound (no partial-distance elimination possible), "
                                              "measured after the timed region, alone",
                         "queries_per_launch": world_q * B * nq,
                         "concurrent_matcher_contexts": len(mctx),
                         "note": "at %d queries per DB pass this kernel is bound by integer VALU issue, not HBM "
                                 "(SURVEY F11): see valu_roofline; launch_ms is a launch's own duration (%s)"
                                 % (world_q * B * nq, "one matcher context: launches do not overlap" if len(mctx) == 1 else
                                    "%d matcher contexts: consecutive launches overlap" % len(mctx))},
            "valu_roofline": {"bound": "valu", "achieved": valu_rate / 1e12, "peak": VALU_PEAK_LANEOPS / 1e12,
                              "unit": "T lane-op/s", "frac": valu_frac,
                              "basis": "executed lane-ops of all matcher launches of the timed region / its wall time",
                              "per_launch_achieved": valu_rate_launch / 1e12, "launches": n_launch,
                              "concurrent_matcher_contexts": len(mctx),
                              "distances_per_launch": distances, "valu_ops_per_distance": laneops,
                              "valu_ops_source": laneops_src, "dense_ops_per_distance": LANEOPS_DENSE},
        }
        if world > 1:
            out["cpu_baseline"] = None                                      # timed at N = 1 only (rank 0)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc, pts, off, frames, k, args.radius, args.cpu_seconds, stages,
                                               args.iterations, args.min_inliers)
        print(json.dumps(out))
    for c in (octx, vctx, cctx if overlap else None):
        if c is not None:
            c.close()
    for c in mctx:
        c.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
