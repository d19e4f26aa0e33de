#!/usr/bin/env python3
"""bench.py -- frames/s of the detection hot path on MI355X (BASELINE.json metric).

Headline workload (config C3 of BASELINE.json, also run at N=1 because the 1M-descriptor DB fits one GPU): one 640x480
synthetic frame = 1000 ORB descriptors matched against the 1M-descriptor object DB (200 objects x 5000), Hamming brute
force k=2, radius 35, then geometric verification; stages ORB | matcher | verifier, each one batched call per step of 16
frames, on three streams (tod_amd/pipeline.py). The three stages are sized and scheduled as in production but, in the
headline, NOT data-chained: SURVEY 8(d)'s synthetic DB (independent bits) cannot be matched by descriptors of a synthetic
image, so ORB runs on the 8(d) image while the matcher and verifier consume the frame's planted descriptors / keypoints.
The `chained` block (N=1) times the real dataflow: DB trained by todhip_model_* on rendered views, ORB -> match -> verify
consuming each other's device buffers (tod_amd/scenes.py).

With N GPUs (tod_amd/sharded.py::ShardedMatcher) the descriptor rows are split into N object-aligned shards and a step
processes B frames per rank (--batch, 32): descriptors are all-gathered, every rank matches all N x B frames against its shard, the
per-shard candidates are exchanged with one RCCL collective (all-to-all by default: a rank only needs the candidates of
its own frames; --exchange all_gather for the literal all-gather), and every rank merges (order: distance asc, global
row asc) and verifies its own frames. Per-GPU work is constant as N grows (weak scaling). Launch: the driver's
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N`; a bare `python bench.py --gpus N`
starts exactly that as a child process (before anything touches the GPU) and exits with its code.

One JSON line on rank 0. `roofline` is for the dominant kernel (the matcher's DB pass); `cpu_baseline` times the CPU
oracle on a bounded sample of the same workload; `repeats` = the timed region run several times (value = the median);
`chained`, `configs` (C1, C2, C4, C5 single-GPU share), `adapter_path` (one frame at a time through the host-buffer
calls the ecto cells make), `hbm_regime` and `n4` (the 2D-only verifier and the LSH mode at the headline frame shape) are
measured after the timed region, at N=1 only.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
_T0 = time.perf_counter()


def progress(what):
    """One line per phase on stderr (stdout carries the one JSON line only): a run that is still working says so."""
    sys.stderr.write("bench.py [%6.1f s] %s\n" % (time.perf_counter() - _T0, what))
    sys.stderr.flush()

# The HIP runtime maps a process's streams onto 4 hardware queues unless told otherwise (tools/stream_concurrency.py: 4 by
# default, 8 with more queues; beyond 8 busy queues the driver time-slices them and every launch stalls, so 8 it is).
# Must be in the environment before the runtime is loaded (i.e. before torch is imported).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0                               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_FP4_PEAK_TFLOPS = 10000.0                      # MI355X_MICROARCH.md: FP6/FP4 MFMA ~10 PF dense
MFMA_BF16_PEAK_TFLOPS = 2500.0                      # MI355X_MICROARCH.md: BF16 MFMA ~2.5 PF dense
MFMA_BF16_MEASURED_TFLOPS = 1840.0                  # profiles/r02_mfma_bf16_peak_microbench.txt: bare loop, random bf16 operands, sustained
FLOP_PER_PAIR = 512.0                               # a 256-bit Hamming distance on the matrix cores = 256 multiply-adds
# 32-bit integer VALU ops (v_xor_b32, v_bcnt_u32_b32) issue at 16 lanes/clk/SIMD on gfx950 (tools/valu_peak.hip,
# profiles/r01_valu_peak_microbench.txt: 38-40 T lane-op/s): the roof of the vector-ALU engine (--engine valu)
VALU_PEAK_LANEOPS = 39.81e12
LANEOPS_DENSE = 16                                  # 8 v_xor_b32 + 8 accumulating v_bcnt_u32_b32 per full 256-bit pair


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="steps per timed region (200 x 32 frames at ~2.6 ms: 0.5 s per region)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=3, help="the timed region of --steps steps is run this many times; value = median")
    ap.add_argument("--objects", type=int, default=200, help="objects of 5000 descriptors (200 -> 1M rows)")
    ap.add_argument("--nq", type=int, default=1000)
    ap.add_argument("--k", type=int, default=2)
    ap.add_argument("--radius", type=int, default=35)
    ap.add_argument("--frames", type=int, default=64, help="distinct synthetic frames cycled through (64 = two distinct batches of 32 "
                                                           "distinct frames: a step is not a replay of the step before)")
    ap.add_argument("--stages", default="orb,match,verify", help="comma list of: orb,match,verify")
    ap.add_argument("--batch", type=int, default=32,
                    help="frames per rank per step of the headline pipeline (tools/batch_sweep.sh: 16 -> 11.8k frames/s, 24 -> 12.0k, 32 -> 12.4k, "
                         "48 -> 11.9k: the per-launch costs of the DB pass are spread over more queries, and 250 query groups tile the "
                         "matrix-core kernel better than 125 or 375); the other blocks keep their own batch sizes")
    ap.add_argument("--verify-workers", type=int, default=0,
                    help="verifier batches in flight in the chained and C5 pipelines (one context, stream and host thread each); 0 = 4: the "
                         "verifier mostly waits for single-wave kernels, so batches overlap almost freely -- until the process has more busy "
                         "streams than the eight hardware queues (tools/chained_flights.sh: 1 -> 2250, 3 -> 3550, 4 -> 4200, 5 -> 2500 frames/s)")
    ap.add_argument("--engine", choices=("auto", "valu", "mfma"), default="auto",
                    help="the exact Hamming search's engine: vector ALU (K4) or matrix cores (K4x); identical results")
    ap.add_argument("--exchange", choices=("all_to_all", "all_gather"), default="all_to_all",
                    help="several ranks: how the per-shard candidates travel. A rank only merges its own frames, so an "
                         "all-to-all moves 1/world of an all-gather's bytes over the point-to-point xGMI links")
    ap.add_argument("--replicas", action="store_true",
                    help="several ranks: every rank holds the WHOLE DB and matches only its own frames -- no data-path collective "
                         "(SURVEY 8(e)'s comparison point for DBs that fit one GPU; the default shards the DB rows)")
    ap.add_argument("--serial-exchange", action="store_true",
                    help="several ranks: issue the collectives on the matcher's stream, in program order (gather -> match -> "
                         "exchange -> merge), instead of on their own stream where they overlap the neighbouring DB passes")
    ap.add_argument("--chained-workers", type=int, default=2, help="verifier batches in flight in the chained block (0: --verify-workers)")
    ap.add_argument("--orb-workers", type=int, default=1, help="ORB batches in flight (one context, stream and host thread each), headline and chained block")
    ap.add_argument("--chained-batch", type=int, default=32, help="frames per step of the chained block (tools/chained_shape_sweep.sh: 16 frames per "
                    "step leave the matcher's DB pass half as many queries to amortise its rows over: 6.5-7.1k frames/s where 32 give 8.8-10k)")
    ap.add_argument("--chained-latency-cus", type=int, default=0,
                    help="chained block: the matcher's stream keeps off this many compute units (todhip_set_cu_partition), on which the "
                         "verifier's single-wave launch groups (sprints, clique gates, growth: its side streams) then run alone; ORB's and "
                         "the verifier's wide kernels keep the whole chip. Paid while the block was verifier-bound (16 frames per step, the "
                         "matcher's waves filling every register file: 6.4k frames/s without, 7.4k with 96); since the matcher's waves leave "
                         "80 registers per SIMD free (launch_topk_mfma) the block is matcher-bound and the partition only costs (32 frames "
                         "per step: 8.9k without, 7.5k with 96); 0 = no partition")
    ap.add_argument("--latency-cus", type=int, default=0, help="reserve this many compute units for the latency-bound stages' streams "
                    "(todhip_set_cu_partition): ORB and verifier kernels then never share a SIMD with the matcher's DB pass")
    ap.add_argument("--iterations", type=int, default=2500, help="n_ransac_iterations (conf/detection.ork:38)")
    ap.add_argument("--min-inliers", type=int, default=8, help="min_inliers (conf/detection.ork:39)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--extras", default="chained,configs,adapter,hbm,n4", help="N=1 only, after the timed region: comma list of "
                    "chained,configs,adapter,hbm,n4 ('' = none)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks the way the driver does, as a child, and leave with
    its exit code. Nothing has touched the GPU yet (torch is not imported), so this is a plain child process."""
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: launching %s\n" % (args.gpus, " ".join(cmd)))
    raise SystemExit(subprocess.call(cmd))


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


def spread(values):
    v = sorted(values)
    return dict(values=values, median=statistics.median(v), min=v[0], max=v[-1])


class SyntheticPipeline:
    """The three stages on the SURVEY 8(d) synthetic workload with device-resident inputs (the headline's structure, also
    used for the C1 / C2 / C5 blocks): ORB on the 8(d) image, matcher + verifier on the frame's planted descriptors."""

    def __init__(self, torch, capi, device, desc, pts, off, frames, nq, k, radius, B, stages, iterations, min_inliers,
                 engine="auto", H=480, W=640, shard=None, n_levels=3, match_fn=None, main_stream=None, verify_workers=2, orb_workers=1):
        from tod_amd.pipeline import StagePipeline
        self.torch, self.capi = torch, capi
        self.nq, self.k, self.radius, self.B, self.H, self.W = nq, k, radius, B, H, W
        self.iterations, self.min_inliers = iterations, min_inliers
        do_orb, do_verify = "orb" in stages, "verify" in stages
        self.stream = main_stream or pooled_stream(torch, "match")
        self.ctx = capi.Context(device, self.stream.cuda_stream)
        self.ctx.set_matcher_engine(engine)
        self.shard = shard                                               # (rank, count): this device holds one shard of the rows
        self.spans = self.ctx.db_load(desc, pts, off, *(shard or (0, 1)))
        self.info = self.ctx.db_info()
        from math import gcd
        n = len(frames)
        self.period = n // gcd(B, n)

        def batch_of(v, key, dtype=None):
            arrs = [frames[(v * B + b) % n][key] for b in range(B)]
            return torch.from_numpy(np.ascontiguousarray(np.stack(arrs), dtype=dtype)).cuda()

        self.Q_B = [batch_of(v, "q_desc") for v in range(self.period)]                          # [B, Q, 32]
        self.KP_B = [batch_of(v, "kp_xy", np.float32) for v in range(self.period)]              # [B, Q, 2]
        self.CLOUD_B = [batch_of(v, "cloud", np.float32) for v in range(self.period)] if do_verify else []
        self.IMG_B = [batch_of(v, "image") for v in range(self.period)] if do_orb else []       # [B, H, W]
        self.D = max(3, verify_workers + 1)                              # a step's outputs stay untouched until its verifier call is done
        self.outs = [dict(counts=torch.zeros(B * nq, dtype=torch.int32, device="cuda"),
                          matches=torch.zeros((B * nq * k, 4), dtype=torch.int32, device="cuda"),
                          xyz=torch.zeros((B * nq * k, 3), dtype=torch.float32, device="cuda")) for _ in range(self.D)]
        # ORB workers (context + stream + output set each) take alternate steps, as the verifier's do: a batch is ~25 dependent launches
        self.ostreams = [pooled_stream(torch, "orb", j, -1) for j in range(orb_workers)] if do_orb else []
        self.octxs = [capi.Context(device, s.cuda_stream) for s in self.ostreams]
        # two verifier workers (context + stream each) take alternate steps: the verifier is latency bound (host round trips,
        # single-wave clique searches), so two batches in flight fill each other's gaps
        self.vstreams = [pooled_stream(torch, "verify", j, -1) for j in range(verify_workers)] if do_verify else []
        self.vctxs = [capi.Context(device, s.cuda_stream) for s in self.vstreams]
        self.orb_out = [(torch.empty((B, nq, 2), device="cuda"), torch.empty((B, nq, 4), device="cuda"),
                         torch.empty((B, nq, 32), dtype=torch.uint8, device="cuda")) for _ in self.octxs]
        self.n_levels = n_levels
        self.match_fn = match_fn or self._match_local
        self.pipe = StagePipeline(torch, orb=self._orb if do_orb else None, match=lambda i, n_steps: self.match_fn(self, i, n_steps),
                                  verify=self._verify if do_verify else None,
                                  wait_for=lambda i, ev: self.vstreams[i % len(self.vstreams)].wait_event(ev), depth=self.D,
                                  verify_workers=max(verify_workers, 1), orb_workers=max(orb_workers, 1))

    def _orb(self, i):
        o = self.orb_out[i % len(self.octxs)]
        n = self.octxs[i % len(self.octxs)].orb_batch_device(self.IMG_B[i % self.period].data_ptr(), self.B, self.H * self.W, self.H, self.W, self.W,
                                       self.nq, self.n_levels, 1.2, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), self.nq)
        return sum(n)

    @staticmethod
    def _match_local(self, i, n_steps):
        o = self.outs[i % self.D]
        q = self.Q_B[i % self.period]
        self.ctx.match_device(q.data_ptr(), self.B * self.nq, self.k, self.radius, o["counts"].data_ptr(),
                              o["matches"].data_ptr(), o["xyz"].data_ptr())
        return self.stream

    def _verify(self, i):
        o = self.outs[i % self.D]
        rngs = (self.capi.Rng * self.B)(*[self.capi.rng_new(1) for _ in range(self.B)])   # rand() restarts per frame (decision D4)
        poses = self.vctxs[i % len(self.vctxs)].verify_batch_device(self.B, self.KP_B[i % self.period].data_ptr(), self.nq,
                                              self.CLOUD_B[i % self.period].data_ptr(), self.H, self.W, o["counts"].data_ptr(),
                                              o["matches"].data_ptr(), o["xyz"].data_ptr(), self.k, self.spans, self.min_inliers,
                                              self.iterations, 0.01, rngs)
        return sum(len(p) for p in poses)

    def close(self):
        self.pipe.close()
        for c in self.octxs + [self.ctx] + self.vctxs:
            if c is not None:
                c.close()


def timed_regions(torch, run, fence, steps, repeats, reduce_max=None):
    """`repeats` timed regions of exactly `steps` steps, each bracketed by fence() on both sides; seconds per region."""
    out = []
    for _ in range(repeats):
        fence()
        t0 = time.perf_counter()
        run(steps)
        fence()
        dt = time.perf_counter() - t0
        out.append(reduce_max(dt) if reduce_max else dt)
    return out


def launch_ms(c0, c1):
    n = c1.n_match_kernel_launches - c0.n_match_kernel_launches
    return (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / max(n, 1), n


# Streams are a scarce resource: the runtime maps them onto 8 hardware queues (header), and a pipeline whose streams share queues
# with another pipeline's idle ones runs its stages one behind the other (the chained block lost a quarter of its rate to the
# headline pipeline's four idle streams). Pipelines that never run at the same time therefore draw from one pool, per role.
_STREAMS = {}


_LATENCY_CUS = 0                                                          # --latency-cus: the CU partition (todhip_set_cu_partition)


def pooled_stream(torch, role, index=0, priority=0):
    key = (role, index)
    if key not in _STREAMS:
        solo = os.environ.get("TOD_BENCH_PARTITION_SOLO") == "1"              # experiment: only the matcher leaves the reserved CUs; ORB and the
        if _LATENCY_CUS > 0 and (role == "match" or not solo):                 # verifier's wide kernels keep the whole chip, the verifier's
            from tod_amd import capi                                          # single-wave groups (its side streams) get the reserved CUs
            _STREAMS[key] = torch.cuda.ExternalStream(capi.stream_create(torch.cuda.current_device(), latency=(role != "match")))
        else:
            _STREAMS[key] = torch.cuda.Stream(priority=priority)
    return _STREAMS[key]


# ------------------------------------------------------------------------------------------------------------ extras (N = 1)
def run_chained(torch, capi, device, args):
    """Real dataflow: DB trained by todhip_model_* on rendered views of 200 textured planes (this library's own ORB
    descriptors: real rBRIEF statistics), then per step ORB -> matcher -> verifier on 16 rendered detection views, each stage
    consuming the previous one's device buffers (keypoints + descriptors -> matches -> poses from the depth image)."""
    from tod_amd import scenes
    from tod_amd.pipeline import StagePipeline
    t_setup = time.perf_counter()
    B, nq, k, radius = args.chained_batch, args.nq, args.k, args.radius
    n_obj = args.objects
    textures = scenes.make_textures(n_obj)
    tctx = capi.Context(device)
    desc, pts, off = scenes.train_db(tctx, textures, rows_per_object=5000)
    tctx.close()
    batches = scenes.make_detection_batches(textures, 4, B)
    H, W = scenes.H, scenes.W
    lat_cus = 0 if _LATENCY_CUS > 0 else args.chained_latency_cus         # (--latency-cus partitions every stream of the process already)
    if lat_cus > 0:
        capi.set_cu_partition(lat_cus)
        key = ("match-partitioned", lat_cus)
        if key not in _STREAMS:
            _STREAMS[key] = torch.cuda.ExternalStream(capi.stream_create(device, latency=False))
        mstream = _STREAMS[key]
    else:
        mstream = pooled_stream(torch, "match")
    NO = max(args.orb_workers, 1)
    ostreams = [pooled_stream(torch, "orb", j, -1) for j in range(NO)]
    NV = args.chained_workers if args.chained_workers > 0 else (args.verify_workers if args.verify_workers > 0 else 4)
    vstreams = [pooled_stream(torch, "verify", j, -1) for j in range(NV)]
    mctx, octxs = capi.Context(device, mstream.cuda_stream), [capi.Context(device, s.cuda_stream) for s in ostreams]
    vctxs = [capi.Context(device, s.cuda_stream) for s in vstreams]
    mctx.set_matcher_engine(args.engine)
    spans = mctx.db_load(desc, pts, off)
    # ring depth: the matcher of step i waits for verify(i - D); with 2 workers at ~5.4 ms per batch and a 3 ms step, D = 3 leaves no
    # slack for jitter (9.6-9.9k frames/s), D = 4 does (9.9-10.0k on the same box)
    D, P = max(int(os.environ.get("TOD_BENCH_CHAINED_DEPTH", "4")), NV + 1), len(batches)
    R = 2 * D                                                             # ORB output ring: ORB runs D ahead of the matcher, the verifier D behind
    orb_ring = [dict(kp=torch.zeros((B, nq, 2), device="cuda"), aux=torch.zeros((B, nq, 4), device="cuda"),
                     desc=torch.zeros((B, nq, 32), dtype=torch.uint8, device="cuda"), n=[nq] * B) for _ in range(R)]
    outs = [dict(counts=torch.zeros(B * nq, dtype=torch.int32, device="cuda"),
                 matches=torch.zeros((B * nq * k, 4), dtype=torch.int32, device="cuda"),
                 xyz=torch.zeros((B * nq * k, 3), dtype=torch.float32, device="cuda")) for _ in range(D)]
    stats = dict(frames=0, right_object=0, pose_ok=0, poses=0, kp_short=0)
    stats_lock = threading.Lock()

    def orb(i):
        s = orb_ring[i % R]
        s["n"] = octxs[i % NO].orb_batch_device(batches[i % P]["images"].data_ptr(), B, H * W, H, W, W, nq, 3, 1.2, s["kp"].data_ptr(),
                                       s["aux"].data_ptr(), s["desc"].data_ptr(), nq)
        return sum(s["n"])

    def match(i, n_steps):
        s, o = orb_ring[i % R], outs[i % D]
        mctx.match_device(s["desc"].data_ptr(), B * nq, k, radius, o["counts"].data_ptr(), o["matches"].data_ptr(), o["xyz"].data_ptr())
        if min(s["n"]) < nq:                                             # a frame with fewer keypoints pads with counts 0 (todhip.h)
            with torch.cuda.stream(mstream):
                c2 = o["counts"].view(B, nq)
                for b, nb in enumerate(s["n"]):
                    if nb < nq:
                        c2[b, nb:] = 0
            stats["kp_short"] += 1
        return mstream

    def verify(i):
        s, o, bt = orb_ring[i % R], outs[i % D], batches[i % P]
        rngs = (capi.Rng * B)(*[capi.rng_new(1) for _ in range(B)])
        poses = vctxs[i % NV].verify_batch_device(B, s["kp"].data_ptr(), nq, 0, H, W, o["counts"].data_ptr(), o["matches"].data_ptr(),
                                         o["xyz"].data_ptr(), k, spans, args.min_inliers, args.iterations, 0.01, rngs,
                                         depth=(bt["depth"].data_ptr(), False, scenes.K))
        tally = dict(frames=0, right_object=0, pose_ok=0, poses=0)           # per call; the verifier workers run this concurrently
        for f, pl in enumerate(poses):
            tally["frames"] += 1
            tally["poses"] += len(pl)
            hit = [p for p in pl if p["object"] == bt["objects"][f]]
            if hit:
                tally["right_object"] += 1
                Rt, tt = bt["poses"][f]
                if np.abs(hit[0]["R"] - Rt).max() < 0.03 and np.abs(hit[0]["t"] - tt).max() < 0.006:
                    tally["pose_ok"] += 1
        with stats_lock:
            for key, v in tally.items():
                stats[key] += v
        return sum(len(p) for p in poses)

    pipe = StagePipeline(torch, orb=orb, match=match, verify=verify, wait_for=lambda i, ev: vstreams[i % NV].wait_event(ev), depth=D,
                         verify_workers=NV, orb_workers=NO)
    setup_s = time.perf_counter() - t_setup
    pipe.run(max(3, 2 * D))                                               # fill the pipeline: D batches are in flight in steady state
    torch.cuda.synchronize()
    for key in stats:
        stats[key] = 0
    pipe.reset_stats()
    mctx.set_kernel_timing(True)
    c0 = mctx.counters()
    steps = min(max(args.steps, 8 * D), 120)                              # long enough that filling and draining D batches in flight is a few per cent
    secs = timed_regions(torch, pipe.run, torch.cuda.synchronize, steps, args.repeats)
    c1 = mctx.counters()
    k_ms, n_l = launch_ms(c0, c1)
    fps = [steps * B / s for s in secs]
    n_f = max(stats["frames"], 1)
    out = {"what": "data-chained pipeline: DB trained by todhip_model_* on %d rendered views per object (ORB descriptors of this "
                   "library), per step todhip_orb_batch_device -> todhip_match_device -> todhip_verify_batch_device_depth on %d rendered "
                   "detection views, every stage reading the previous stage's device buffers" % (len(scenes.TRAIN_VIEWS), B),
           "db_rows": int(off[-1]), "db_objects": n_obj, "k": k, "radius": radius, "frames_per_step": B, "steps": steps,
           "verifier_batches_in_flight": NV, "latency_cus": lat_cus,
           "frames_per_s": spread(fps), "ms_per_step": statistics.median(secs) / steps * 1e3,
           "matcher_launch_ms": k_ms, "matcher_launches": n_l,
           "stage_ms_per_step": {key: 1e3 * v / (steps * args.repeats) for key, v in pipe.stage_s.items()},
           "keypoints_per_frame": pipe.n_kp / max(pipe.n_steps * B, 1),
           "poses_per_frame": stats["poses"] / n_f, "frames_with_the_right_object": stats["right_object"] / n_f,
           "frames_with_the_rendering_pose": stats["pose_ok"] / n_f,
           "pose_tolerance": "max |dR| < 0.03, max |dt| < 6 mm against the pose the view was rendered from",
           "setup_s": setup_s}
    pipe.close()
    # the matcher's DB pass at the HEADLINE's launch shape (32 frames = 32 000 queries per pass) on this block's DB and descriptors --
    # the data it will see: biased, correlated rBRIEF bits and self-similar textures -- alone on the GPU, on an unpartitioned stream
    if B * nq in (16000, 32000) and R >= 2:
        fctx = capi.Context(device)
        fctx.set_matcher_engine(args.engine)
        fctx.db_load(desc, pts, off)
        de32 = (orb_ring[0]["desc"] if B * nq == 32000 else torch.cat([orb_ring[0]["desc"], orb_ring[1]["desc"]])).contiguous()
        n32 = 32000
        c32 = torch.zeros(n32, dtype=torch.int32, device="cuda"); m32 = torch.zeros((n32 * k, 4), dtype=torch.int32, device="cuda")
        x32 = torch.zeros((n32 * k, 3), dtype=torch.float32, device="cuda")
        call32 = lambda: fctx.match_device(de32.data_ptr(), n32, k, radius, c32.data_ptr(), m32.data_ptr(), x32.data_ptr())
        for _ in range(6):                                                  # (a fresh context walks the block forms from the launches' reports:
            call32(); fctx.synchronize()                                     # settled after three launches whose reports it has seen)
        fctx.set_kernel_timing(True); f0 = fctx.counters()
        for _ in range(6):
            call32()
        fctx.synchronize(); ms32, n32l = launch_ms(f0, fctx.counters()); fctx.set_kernel_timing(False)
        tf32 = float(n32) * int(off[-1]) * FLOP_PER_PAIR / (ms32 * 1e-3) / 1e12 if ms32 > 0 else 0.0
        walks = None
        wp = os.path.join(ROOT, "profiles", "r03_k4x_on_chained_db.json")
        if os.path.exists(wp):
            walks = json.load(open(wp)).get("chained_db", {}).get("walk_fraction")
        out["matcher_at_headline_shape"] = {"what": "32 000 of this block's ORB descriptors x its %d-row trained DB per launch, alone" % int(off[-1]),
                                            "launch_ms": ms32, "launches": n32l, "TFLOPs": tf32, "frac": tf32 / MFMA_FP4_PEAK_TFLOPS,
                                            "block_form": {4: "whole blocks", 3: "split after 3 of 4 MFMAs", 2: "split after 2 of 4 MFMAs"}.get(int(fctx.counters().last_block_split), "?") + " (chosen by the context from its launches' reports)",
                                            "blocks_that_walked_rows": walks,
                                            "blocks_source": "profiles/r03_k4x_on_chained_db.json (diagnostics build, tools/k4x_walks.sh)" if walks is not None else None}
        fctx.close()
    for c in [mctx] + octxs + vctxs:
        c.close()
    if lat_cus > 0:
        capi.set_cu_partition(_LATENCY_CUS)
    return out


def run_adapter_path(torch, capi, device, desc, pts, off, frames, args):
    """What adapter/ecto_cells.hpp can reach: one frame at a time through the host-buffer calls, every call synchronous
    (todhip_orb = FeatureDescriptor, todhip_match = DescriptorMatcher::process, todhip_verify = GuessGenerator::process)."""
    ctx = capi.Context(device)
    ctx.set_matcher_engine(args.engine)
    spans = ctx.db_load(desc, pts, off)
    t_stage = {"orb": [], "match": [], "verify": []}
    n_poses = 0
    reps = 4
    for r in range(reps + 1):
        for fr in frames:
            t0 = time.perf_counter()
            ctx.orb(fr["image"], args.nq, 3, 1.2)
            t1 = time.perf_counter()
            row_ptr, m, xyz = ctx.match(fr["q_desc"], args.k, args.radius)
            t2 = time.perf_counter()
            poses = ctx.verify(fr["kp_xy"], fr["cloud"], row_ptr, m, xyz, spans, args.min_inliers, args.iterations, 0.01, capi.rng_new(1))
            t3 = time.perf_counter()
            if r > 0:                                                     # first pass = warm-up
                t_stage["orb"].append(t1 - t0); t_stage["match"].append(t2 - t1); t_stage["verify"].append(t3 - t2)
                n_poses += len(poses)
    ctx.close()
    med = {key: statistics.median(v) for key, v in t_stage.items()}
    per_frame = [a + b + c for a, b, c in zip(t_stage["orb"], t_stage["match"], t_stage["verify"])]
    return {"what": "one frame at a time through todhip_orb + todhip_match + todhip_verify with host buffers (PCIe inclusive, one "
                    "synchronous call per cell) -- the only forms adapter/ecto_cells.hpp calls; C3 workload",
            "frames": len(per_frame), "frames_per_s": 1.0 / statistics.median(per_frame),
            "frames_per_s_match_verify_only": 1.0 / (med["match"] + med["verify"]),
            "ms_per_call": {key: 1e3 * v for key, v in med.items()},
            "ms_per_frame_spread": spread([1e3 * v for v in sorted(per_frame)[::max(len(per_frame) // 8, 1)]]),
            "poses_per_frame": n_poses / max(len(per_frame), 1)}


def run_n4(torch, capi, device, desc, pts, off, frames, args):
    """SURVEY 8(f) N4 at the headline frame shape, single frame, host buffers: the 2D-only branch (todhip_verify_2d, no cloud) and the
    LSH-approximate mode at the reference configs' parameters, with its recall against the exact search."""
    ctx = capi.Context(device)
    ctx.set_matcher_engine(args.engine)
    spans = ctx.db_load(desc, pts, off)
    K = np.array([[525.0, 0, 320.0], [0, 525.0, 240.0], [0, 0, 1]], np.float32)            # synth.make_frame's camera
    t2d, n2d, same_obj = [], 0, 0
    exact = []
    for r in range(3):
        for fr in frames:
            row_ptr, m, xyz = ctx.match(fr["q_desc"], args.k, args.radius)
            if r == 0:
                exact.append((row_ptr, m))
            t0 = time.perf_counter()
            poses = ctx.verify_2d(fr["kp_xy"], K, row_ptr, m, xyz, spans, args.min_inliers, 1000, 3.0, capi.rng_new(1))
            if r > 0:
                t2d.append(time.perf_counter() - t0); n2d += len(poses)
    out = {"verify_2d": {"what": "todhip_verify_2d on the headline frames' matches: keypoints + camera matrix only (no cloud), 1000 P3P "
                                 "hypotheses per object, reprojection threshold 3 px, host buffers", "ms_per_call": 1e3 * statistics.median(t2d),
                         "poses_per_frame": n2d / max(len(t2d), 1)}}
    lsh = {}
    for name, (tables, ks, lvl) in (("detection.ork", (10, 16, 1)), ("detection.ros.ork", (8, 24, 2))):
        t0 = time.perf_counter()
        ctx.set_lsh(tables, ks, lvl)
        ctx.synchronize()
        t_build = time.perf_counter() - t0
        ts, kept, want = [], 0, 0
        for r in range(3):
            for f, fr in enumerate(frames):
                t0 = time.perf_counter()
                row_ptr, m, _ = ctx.match(fr["q_desc"], args.k, args.radius)
                if r > 0:
                    ts.append(time.perf_counter() - t0)
                else:                                                    # recall: matches of the exact search that the index also returns
                    erp, em = exact[f]
                    e = set(zip(em["queryIdx"].tolist(), em["imgIdx"].tolist(), em["trainIdx"].tolist()))
                    g = set(zip(m["queryIdx"].tolist(), m["imgIdx"].tolist(), m["trainIdx"].tolist()))
                    kept += len(e & g); want += len(e)
        lsh[name] = {"n_tables": tables, "key_size": ks, "multi_probe_level": lvl, "index_build_ms": 1e3 * t_build,
                     "ms_per_call": 1e3 * statistics.median(ts), "recall_of_the_exact_matches_within_radius": kept / max(want, 1)}
    ctx.set_lsh(0)
    ts = []
    for r in range(3):
        for fr in frames:
            t0 = time.perf_counter()
            ctx.match(fr["q_desc"], args.k, args.radius)
            if r > 0:
                ts.append(time.perf_counter() - t0)
    out["lsh_mode"] = {"what": "todhip_match (host buffers, one 1000-descriptor frame) with todhip_set_lsh at the parameters of the reference's "
                               "configs, against the exact search (the default) on the same context", "configs": lsh,
                       "exact_ms_per_call": 1e3 * statistics.median(ts), "db_rows": int(off[-1])}
    ctx.close()
    return out


def run_hbm_regime(torch, capi, device, args):
    """BASELINE.json's "achieved HBM GB/s on BF-matcher" where it can be measured: few queries per pass over a DB far larger than
    the 256 MiB Infinity Cache, so every 32-byte row comes from HBM and is used by only Q queries (at the headline's 16 000
    queries per pass the same kernel family is three orders of magnitude on the compute side of that roof)."""
    rows, Q, k, radius = 40_000_000, 16, args.k, args.radius
    rng = np.random.Generator(np.random.PCG64(77))
    desc = rng.integers(0, 256, size=(rows, 32), dtype=np.uint8)
    pts = np.zeros((rows, 3), np.float32)
    off = (np.arange(rows // 5000 + 1, dtype=np.uint64) * 5000).astype(np.uint32)
    ctx = capi.Context(device)
    ctx.db_load(desc, pts, off)
    qrows = rng.choice(rows, Q, replace=False)
    q = desc[qrows] ^ np.packbits(rng.random((Q, 256)) < 0.08, axis=1, bitorder="little")
    del desc, pts
    d_q = torch.from_numpy(np.ascontiguousarray(q)).cuda()
    cnt = torch.zeros(Q, dtype=torch.int32, device="cuda"); mm = torch.zeros((Q * k, 4), dtype=torch.int32, device="cuda")
    xx = torch.zeros((Q * k, 3), dtype=torch.float32, device="cuda")
    out = {}
    for eng in ("mfma", "valu"):
        ctx.set_matcher_engine(eng)
        call = lambda: ctx.match_device(d_q.data_ptr(), Q, k, radius, cnt.data_ptr(), mm.data_ptr(), xx.data_ptr())
        for _ in range(2):
            call()
        ctx.synchronize(); ctx.set_kernel_timing(True); c0 = ctx.counters()
        for _ in range(6):
            call()
        ctx.synchronize(); ms, n_l = launch_ms(c0, ctx.counters()); ctx.set_kernel_timing(False)
        out[eng] = (ms, int(cnt.sum().item()))
    ctx.close()
    alg_bytes = rows * 32 + Q * (32 + k * 8)
    gbs = alg_bytes / (out["mfma"][0] * 1e-3) / 1e9
    # HBM traffic of this launch from the PMC passes (tools/profile_k4x_q32.sh: FETCH_SIZE and WRITE_SIZE in runs of their own,
    # FETCH_SIZE doubled as profiles/r03_fetch_calibration.json measures for vector loads); only quoted for the shape it was taken on
    traffic, traffic_src = None, None
    pmc_path = os.path.join(ROOT, "profiles", "r03_k4x_q32_pmc.json")
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
        if pmc.get("queries_per_launch") == Q and pmc.get("db_rows") == rows and k == 2 and radius == 35:
            traffic = pmc["hbm_traffic_bytes_per_launch"]
            traffic_src = "profiles/r03_k4x_q32_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; kernel trace %.1f us)" % (
                pmc.get("avg_duration_ns_kernel_trace", 0.0) / 1e3)
    return {"what": "%d queries per pass over a %d-row DB (%.2f GB of descriptors: five times the Infinity Cache), k=%d, radius %d; "
                    "todhip_match_device, kernel hamming_topk_mfma_q32 (one 32-query block per wave: 4 MFMAs per KB of rows)"
                    % (Q, rows, rows * 32 / 1e9, k, radius),
            "roofline": {"kernel": "hamming_topk_mfma_q32", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": gbs / HBM_PEAK_GBS, "launch_ms": out["mfma"][0], "algorithmic_bytes": alg_bytes, "traffic": traffic,
                         "traffic_source": traffic_src},
            "vector_engine_launch_ms": out["valu"][0], "vector_engine_GBs": alg_bytes / (out["valu"][0] * 1e-3) / 1e9,
            "queries_with_a_match": out["mfma"][1], "engines_agree_on_match_count": out["mfma"][1] == out["valu"][1]}


def run_configs(torch, capi, synth, device, args):
    """BASELINE.json configs other than the headline's (C3), each on one GPU: frames/s + the dominant kernel's time."""
    out = {}

    def pipeline_block(name, desc, pts, off, frames, nq, k, radius, B, stages, steps, H=480, W=640, shard=None, match_fn=None, note="",
                       verify_workers=2):
        sp = SyntheticPipeline(torch, capi, device, desc, pts, off, frames, nq, k, radius, B, stages, args.iterations, args.min_inliers,
                               engine=args.engine, H=H, W=W, shard=shard, match_fn=match_fn, verify_workers=verify_workers)
        sp.pipe.run(max(2, 2 * sp.D))
        torch.cuda.synchronize()
        sp.pipe.reset_stats()
        sp.ctx.set_kernel_timing(True)
        c0 = sp.ctx.counters()
        secs = timed_regions(torch, sp.pipe.run, torch.cuda.synchronize, steps, args.repeats)
        k_ms, n_l = launch_ms(c0, sp.ctx.counters())
        blk = {"workload": name, "frames_per_step": B, "steps": steps, "frames_per_s": spread([steps * B / s for s in secs]),
               "ms_per_step": statistics.median(secs) / steps * 1e3, "matcher_launch_ms": k_ms,
               "stage_ms_per_step": {key: 1e3 * v / (steps * args.repeats) for key, v in sp.pipe.stage_s.items()},
               "poses_per_frame": sp.pipe.n_poses / max(sp.pipe.n_steps * B, 1)}
        if note:
            blk["note"] = note
        sp.close()
        progress("  config block done: %s" % name[:40])
        return blk

    # ---- C1: the reference's own case -- one frame, ORB-500, a 1-object DB, k = 5 (DescriptorMatcher.cpp:211), radius 35
    d1, p1, o1 = synth.make_db(1)
    f1 = [synth.make_frame(d1, p1, o1, 500, frame=f, visible_object=0) for f in range(4)]
    for f, fr in enumerate(f1):
        fr["image"] = synth.make_image(f)
    out["C1"] = pipeline_block("C1: single 640x480 frame, ORB-500 vs a 1-object DB (5000 descriptors), k=5, radius 35, full verifier; "
                               "a step = ONE frame (the reference's frame-at-a-time regime)", d1, p1, o1, f1, 500, 5, 35, 1,
                               ["orb", "match", "verify"], 40)
    out["C1"]["batched"] = pipeline_block("the same frames, 16 per step", d1, p1, o1, f1, 500, 5, 35, 16, ["orb", "match", "verify"], 10)
    # ---- C2: ORB-1000 vs 100k descriptors, k = 2
    d2, p2, o2 = synth.make_db(20)
    f2 = [synth.make_frame(d2, p2, o2, 1000, frame=f, visible_object=(17 * f + 3) % 20) for f in range(8)]
    for f, fr in enumerate(f2):
        fr["image"] = synth.make_image(f)
    out["C2"] = pipeline_block("C2: ORB-1000 per frame vs 100k-descriptor DB (20 objects x 5000), Hamming BF k=2, radius 35, 16 frames per step",
                               d2, p2, o2, f2, 1000, 2, 35, 16, ["orb", "match", "verify"], 15)
    del d2, p2, f2
    # ---- C5, one GPU's share of the 8-GPU job: 4 of the 32 1080p frames (ORB-2000 + verifier), all 32 frames' descriptors
    # against this rank's 250k-row shard of the 2M-row DB, merge of its own 4 frames (no peers here: the other ranks' candidate
    # lists are absent, the visible objects are chosen inside this shard)
    d5, p5, o5 = synth.make_db(400)
    B5, nq5, world5 = 4, 2000, 8
    f5 = [synth.make_frame(d5, p5, o5, nq5, frame=f, visible_object=(7 * f + 3) % 50, H=1080, W=1920, f=1400.0) for f in range(B5)]
    for f, fr in enumerate(f5):
        fr["image"] = synth.make_image(f, H=1080, W=1920, n_rect=8000)
    keys5 = {}
    NV5 = args.verify_workers if args.verify_workers > 0 else 4

    def match_c5(sp, i, n_steps):
        if "q_all" not in keys5:
            keys5["q_all"] = sp.Q_B[0].repeat(world5, 1, 1).contiguous()                # what the all-gather delivers: 32 frames
            keys5["keys"] = torch.empty((world5 * B5 * nq5, 2), dtype=torch.int64, device="cuda")
        o = sp.outs[i % sp.D]
        sp.ctx.match_shard_device(keys5["q_all"].data_ptr(), world5 * B5 * nq5, 2, 35, keys5["keys"].data_ptr())
        sp.ctx.merge_shards_device(keys5["keys"].data_ptr(), 1, B5 * nq5, 2, 35, o["counts"].data_ptr(), o["matches"].data_ptr(),
                                   o["xyz"].data_ptr())
        return sp.stream

    out["C5_single_gpu_share"] = pipeline_block(
        "C5, one rank's share of the 8-GPU job: per step ORB-2000 on 4 of the 32 1080p frames, the 32 x 2000 descriptors against this "
        "rank's 250k-row shard of the 2M-row DB (k=2, radius 35), merge + full verifier for its own 4 frames", d5, p5, o5, f5, nq5, 2, 35,
        B5, ["orb", "match", "verify"], 40, H=1080, W=1920, shard=(0, world5), match_fn=match_c5, verify_workers=NV5,
        note="no collectives on one GPU: the candidate exchange (32 x 2000 x 2 keys x 8 B per rank) is missing from this figure; "
             "%d verifier batches (of 4 frames) in flight" % NV5)
    del d5, p5, f5, keys5
    # ---- C4: float descriptors, L2 brute force as a bf16 MFMA GEMM + exact refinement (matcher only: not a reference feature)
    d4, p4, o4 = synth.make_sift_db(100)
    q4, _ = synth.make_sift_queries(d4, 1000, frame=0)
    c4 = capi.Context(device)
    c4.db_load(d4, p4, o4)
    dq = torch.from_numpy(q4).cuda()
    cnt = torch.zeros(1000, dtype=torch.int32, device="cuda"); mm = torch.zeros((2000, 4), dtype=torch.int32, device="cuda")
    xx = torch.zeros((2000, 3), dtype=torch.float32, device="cuda")
    call = lambda: c4.match_l2_device(dq.data_ptr(), 1000, 2, 400.0, cnt.data_ptr(), mm.data_ptr(), xx.data_ptr())
    for _ in range(3):
        call()
    c4.synchronize()
    secs = []
    for _ in range(args.repeats):
        t0 = time.perf_counter()
        for _ in range(20):
            call()
        c4.synchronize()
        secs.append((time.perf_counter() - t0) / 20)
    # the GEMM pass alone, live: HIP events around l2_gemm_kernel<2, 4> on the context's stream (todhip_set_kernel_timing), in
    # further calls of the same shape so that the events do not sit inside the timed region above
    c4.set_kernel_timing(True)
    k0 = c4.counters()
    for _ in range(20):
        call()
    c4.synchronize()
    pass2_ms, pass2_n = launch_ms(k0, c4.counters())
    c4.set_kernel_timing(False)
    pass2_s = max(pass2_ms, 1e-6) * 1e-3
    flops = 2.0 * 1000 * d4.shape[0] * 128
    med = statistics.median(secs)
    # ---- the batch form: 16 frames' queries share one pass over the DB (todhip_match_l2_device with F x Q queries)
    F4 = 16
    q16 = np.concatenate([synth.make_sift_queries(d4, 1000, frame=f)[0] for f in range(F4)])
    dq16 = torch.from_numpy(q16).cuda()
    cnt16 = torch.zeros(F4 * 1000, dtype=torch.int32, device="cuda"); mm16 = torch.zeros((F4 * 2000, 4), dtype=torch.int32, device="cuda")
    xx16 = torch.zeros((F4 * 2000, 3), dtype=torch.float32, device="cuda")
    call16 = lambda: c4.match_l2_device(dq16.data_ptr(), F4 * 1000, 2, 400.0, cnt16.data_ptr(), mm16.data_ptr(), xx16.data_ptr())
    for _ in range(3):
        call16()
    c4.synchronize()
    secs16 = []
    for _ in range(args.repeats):
        t0 = time.perf_counter()
        for _ in range(10):
            call16()
        c4.synchronize()
        secs16.append((time.perf_counter() - t0) / 10)
    c4.set_kernel_timing(True)
    k0 = c4.counters()
    for _ in range(10):
        call16()
    c4.synchronize()
    pass2_ms16, pass2_n16 = launch_ms(k0, c4.counters())
    c4.set_kernel_timing(False)
    med16 = statistics.median(secs16)
    flops16 = flops * F4
    same_as_single = bool(torch.equal(cnt16[:1000], cnt))                  # frame 0 of the batch is the single call's frame
    out["C4"] = {"workload": "C4: 1000 SIFT-128 float descriptors vs 500k-row DB, L2 brute force k=2 as a bf16 MFMA GEMM with exact f32 "
                             "refinement (todhip_match_l2_device; matcher only)",
                 "frames_per_s": spread([1.0 / s for s in secs]), "ms_per_call": med * 1e3,
                 "roofline": {"bound": "mfma", "achieved": flops / med / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": flops / med / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                              "note": "2 Q N 128 flop of the distance table / the whole call (seed pass + GEMM pass + exact re-ranking)",
                              "gemm_pass_kernel": {"name": "l2_gemm_kernel<2,4>", "us": pass2_s * 1e6, "launches_timed": pass2_n,
                                                   "TFLOPs": flops / pass2_s / 1e12,
                                                   "frac_of_nominal": flops / pass2_s / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                                   "frac_of_measured_roof": flops / pass2_s / 1e12 / MFMA_BF16_MEASURED_TFLOPS,
                                                   "source": "live: HIP events around the kernel on the context's stream (todhip_set_kernel_timing); "
                                                             "profiles/r03_l2_kernel_stats.csv is the rocprofv3 kernel trace of the same call"},
                              "measured_mfma_roof": {"TFLOPs_random_operands": MFMA_BF16_MEASURED_TFLOPS, "TFLOPs_zero_operands": 2480.0,
                                                     "source": "profiles/r02_mfma_bf16_peak_microbench.txt (tools/mfma_bf16_peak.hip: bare "
                                                               "v_mfma_f32_32x32x16_bf16 loop, operands in registers; the clock the chip holds "
                                                               "depends on the data)"}},
                 "queries_with_a_match": int((cnt > 0).sum().item()),
                 "batched": {"what": "%d frames per call: %d queries share one pass over the DB (todhip_match_l2_device, queryIdx counts through the call)" % (F4, F4 * 1000),
                             "frames_per_s": spread([F4 / s for s in secs16]), "ms_per_call": med16 * 1e3, "ms_per_frame": med16 * 1e3 / F4,
                             "roofline": {"bound": "mfma", "achieved": flops16 / med16 / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": flops16 / med16 / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                          "note": "2 F Q N 128 flop / the whole call",
                                          "gemm_pass_kernel": {"name": "l2_gemm_kernel<2,4>", "us": pass2_ms16 * 1e3, "launches_timed": pass2_n16,
                                                               "frac_of_nominal": flops16 / max(pass2_ms16 * 1e-3, 1e-9) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                                                               "frac_of_measured_roof": flops16 / max(pass2_ms16 * 1e-3, 1e-9) / 1e12 / MFMA_BF16_MEASURED_TFLOPS}},
                             "frame_0_counts_equal_the_single_call": same_as_single}}
    c4.close()
    return out


# ------------------------------------------------------------------------------------------------------------------- main
def main():
    args = parse()
    stages = [s for s in args.stages.split(",") if s]
    extras = [s for s in args.extras.split(",") if s]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("TOD_BENCH_FORCE_DIST") != "1":
        spawn_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                         "--master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d ..." % (args.gpus, world, args.gpus, args.gpus))

    import torch
    import torch.distributed as dist
    from tod_amd import capi, sharded, synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    # rehearsal on a 1-GPU box: TOD_BENCH_BACKEND=gloo TOD_BENCH_ONE_DEVICE=1 puts every rank on device 0
    one_device = os.environ.get("TOD_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("TOD_BENCH_BACKEND", "nccl")
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if args.latency_cus > 0:
        global _LATENCY_CUS
        _LATENCY_CUS = args.latency_cus
        capi.set_cu_partition(args.latency_cus)
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

    # the library torch.distributed's backend drives on this box, as the bench line names it ("nccl" IS RCCL on ROCm)
    collective_backend = {"nccl": "RCCL", "gloo": "gloo (CPU rehearsal)"}.get(dist.get_backend() if use_dist else backend, backend)
    progress("building the synthetic workload")
    desc, pts, off = synth.make_db(args.objects)
    # frame f belongs to rank (f % world): every rank builds and keeps resident only its own --frames frames
    frame_ids = [rank + world * j for j in range(max(args.frames, 1))]
    frames = [synth.make_frame(desc, pts, off, args.nq, frame=f, visible_object=(17 * f + 3) % args.objects) for f in frame_ids]
    for f, fr in zip(frame_ids, frames):
        fr["image"] = synth.make_image(f)
    my_frames = frames

    # one explicit stream for the matcher: libtodhip kernels, torch copies and (serial form) the RCCL collectives, which order
    # themselves against torch's current stream. The default stream's handle is 0 == "create your own" for todhip_create.
    stream = pooled_stream(torch, "match")
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0
    nq, k, B = args.nq, args.k, args.batch
    sharded_db = use_dist and not args.replicas
    overlap = sharded_db and not args.serial_exchange
    sm_box = {}

    def match_sharded(sp, i, n_steps):
        sm = sm_box["sm"]
        if i == 0:
            sm.begin(n_steps, lambda j: (sp.Q_B[j % sp.period], None))
        return sm.step(i, sp.outs[i % sp.D])

    sp = SyntheticPipeline(torch, capi, local_rank, desc, pts, off, my_frames, nq, k, args.radius, B, stages, args.iterations,
                           args.min_inliers, engine=args.engine, shard=(rank, world) if sharded_db else None,
                           match_fn=match_sharded if sharded_db else None, main_stream=stream,
                           # With collectives in the step the overlapped exchange brings a stream of its own, and with five busy streams
                           # the matcher's short kernels between two DB passes run 5-10x slower (tools/dist1_exp2.sh: RCCL on one rank,
                           # 16 frames per step: 9.8k frames/s with two verifier workers, 10.9k with one; plain path 11.6k). At 32 frames
                           # per step that loss is spread over a step twice as long and one worker no longer keeps up
                           # (tools/batch_sweep_dist.sh: 9.3k with one worker, 11.4k with two; plain path 12.4k)
                           verify_workers=int(os.environ.get("TOD_BENCH_HEADLINE_VW", "1" if sharded_db and not args.serial_exchange and B < 24 else "2")),
                           orb_workers=args.orb_workers)
    sp.pipe.next_orb = overlap
    info = sp.info
    check = None
    if sharded_db:
        cstream = torch.cuda.Stream(priority=int(os.environ.get("TOD_BENCH_COMM_PRIO", "-1"))) if overlap else stream
        ops = sharded.GpuOps(sp.ctx, stream, cstream, backend, k, args.radius)
        sm_box["sm"] = sharded.ShardedMatcher(ops, world, rank, B, nq, k, exchange=args.exchange, overlap=overlap)
        # self-check before anything is timed: step 0's merged matches of THIS rank's frames == the result of a replica that
        # holds the whole DB (every rank checks its own frames; the verdict is reduced over the ranks)
        sm = sm_box["sm"]
        sm.begin(1, lambda j: (sp.Q_B[0], None))
        done_on = sm.step(0, sp.outs[0])
        done_on.synchronize()
        rctx = capi.Context(local_rank, stream.cuda_stream)
        rctx.set_matcher_engine(args.engine)
        rctx.db_load(desc, pts, off)
        ref = dict(counts=torch.zeros_like(sp.outs[0]["counts"]), matches=torch.zeros_like(sp.outs[0]["matches"]),
                   xyz=torch.zeros_like(sp.outs[0]["xyz"]))
        rctx.match_device(sp.Q_B[0].data_ptr(), B * nq, k, args.radius, ref["counts"].data_ptr(), ref["matches"].data_ptr(),
                          ref["xyz"].data_ptr())
        stream.synchronize()
        keep = (torch.arange(k, device="cuda").view(1, k) < ref["counts"].view(-1, 1)).view(-1)
        same = (torch.equal(ref["counts"], sp.outs[0]["counts"]) and torch.equal(ref["matches"][keep], sp.outs[0]["matches"][keep]) and
                torch.equal(ref["xyz"][keep], sp.outs[0]["xyz"][keep]))
        rctx.close()
        flag = torch.tensor([1 if same else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        check = bool(flag.item())
        if not check:
            raise SystemExit("sharded step 0 differs from the unsharded result on some rank")

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(dt):
        if not use_dist:
            return dt
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    sp.pipe.run(args.warmup)
    fence()
    sp.pipe.reset_stats()
    sp.ctx.set_kernel_timing(True)
    c0 = sp.ctx.counters()
    kp0, poses0, steps0 = sp.pipe.n_kp, sp.pipe.n_poses, sp.pipe.n_steps
    t_local0 = time.perf_counter()
    secs = timed_regions(torch, sp.pipe.run, fence, args.steps, args.repeats, reduce_max)
    dt_local = time.perf_counter() - t_local0
    c1 = sp.ctx.counters()
    k4_ms, n_launch = launch_ms(c0, c1)
    n_timed_steps = sp.pipe.n_steps - steps0
    dt = statistics.median(secs)
    engine_used = args.engine
    if engine_used == "auto":
        world_q0 = 1 if (args.replicas or not use_dist) else world
        engine_used = "mfma" if (world_q0 * B * nq >= 64 and world_q0 * B * nq * info["shard_rows"] >= (1 << 24)) else "valu"

    # outside the timed region, single device only: the other engine on the same launch, and (vector engine) the dense schedule
    other = {}
    if world == 1 and not use_dist:
        o = sp.outs[0]
        for eng, radius in (("valu", args.radius), ("valu_dense", 256), ("mfma", args.radius)):
            sp.ctx.set_matcher_engine(eng.split("_")[0])
            d0 = sp.ctx.counters()
            for _ in range(3):
                sp.ctx.match_device(sp.Q_B[0].data_ptr(), B * nq, k, radius, o["counts"].data_ptr(), o["matches"].data_ptr(), o["xyz"].data_ptr())
            torch.cuda.synchronize()
            other[eng], _ = launch_ms(d0, sp.ctx.counters())
        sp.ctx.set_matcher_engine(args.engine)
    sp.ctx.set_kernel_timing(False)

    world_q = 1 if (args.replicas or not use_dist) else world          # frames' worth of queries one launch matches: world * B or B
    q_launch = world_q * B * nq
    alg_bytes = info["shard_rows"] * 32 + q_launch * (32 + k * 8)      # N*32 + F*Q*(32 + k*8)
    pairs = float(q_launch) * info["shard_rows"]
    hbm_gbs = alg_bytes / (k4_ms * 1e-3) / 1e9 if k4_ms > 0 else 0.0
    # HBM traffic of the dominant kernel from the PMC passes (FETCH_SIZE + WRITE_SIZE, each in its own rocprofv3 run,
    # tools/profile_k4.sh -> profiles/r03_k4x_pmc.json); only quoted for the workload it was measured on
    traffic, traffic_src, pmc = None, None, {}
    pmc_path = os.path.join(ROOT, "profiles", "r03_k4x_pmc.json" if engine_used == "mfma" else "r01_k4_pmc.json")
    if os.path.exists(pmc_path):
        pmc = json.load(open(pmc_path))
    if (world == 1 and k == 2 and info["shard_rows"] == 1000000 and args.radius == 35 and pmc.get("queries_per_launch") == q_launch):
        traffic = pmc["hbm_traffic_bytes_per_launch"]
        traffic_src = os.path.relpath(pmc_path, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"

    if rank == 0:
        if engine_used == "mfma":
            tflops = pairs * FLOP_PER_PAIR / (k4_ms * 1e-3) / 1e12 if k4_ms > 0 else 0.0
            roofline = {"kernel": "hamming_topk_mfma", "bound": "mfma", "achieved": tflops, "peak": MFMA_FP4_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": tflops / MFMA_FP4_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                        "launch_ms": k4_ms, "launches": n_launch, "queries_per_launch": q_launch, "db_rows": info["shard_rows"],
                        "algorithmic_flops": pairs * FLOP_PER_PAIR, "algorithmic_bytes": alg_bytes,
                        "pairs_per_s": pairs / (k4_ms * 1e-3) if k4_ms > 0 else 0.0,
                        "hbm": {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                                "note": "BASELINE.json's 'achieved HBM GB/s on BF-matcher': every 32-byte row is reused by all %d queries "
                                        "of the pass, so the pass is three orders of magnitude on the compute side of the HBM roof; "
                                        "`hbm_regime` (and tools/k4_small_q.py) measure the memory-bound regime: a few queries per "
                                        "pass over a DB beyond the Infinity Cache" % q_launch},
                        "measured_mfma_roof": pmc.get("measured_mfma_roof"),
                        "note": "exact 256-bit Hamming distances as fp4 (+-1) dot products on the matrix cores: 512 flop per (query, row) "
                                "pair, v_mfma_f32_32x32x64_f8f6f4. `achieved` counts those ALGORITHMIC flops (SURVEY 8(d)); since round 3 a "
                                "32 x 32 block only executes the rest of its 256 bit positions when its first 128 (or 192) leave a pair inside the "
                                "limit (partial-distance elimination, exact for any data, the form adapts to it; `half_blocks`), so on this workload about half "
                                "of them are executed and `frac` measures the pass against the roof of the arithmetic it replaces, not "
                                "matrix-pipe occupancy. launch_ms = HIP events around the launch on its own stream, averaged over the "
                                "timed regions"}
            hc = sp.ctx.counters()
            if hc.k4x_half_blocks:
                done = hc.k4x_half_blocks_completed / hc.k4x_half_blocks
                first = {2: 0.5, 3: 0.75}.get(int(hc.last_block_split), 1.0)   # the block form the context has settled on (todhip_set_matcher_block_split: adaptive)
                executed = first + (1.0 - first) * done if first < 1.0 else 1.0
                roofline["half_blocks"] = {"split_after_mfmas_of_4": int(hc.last_block_split), "blocks_started_as_parts": int(hc.k4x_half_blocks),
                                           "fraction_completed": done, "executed_over_algorithmic_flops": executed,
                                           "executed_TFLOPs": tflops * executed, "executed_frac_of_peak": tflops * executed / MFMA_FP4_PEAK_TFLOPS}
        else:
            laneops = float(pmc.get("valu_insts_per_row_and_wave", LANEOPS_DENSE)) if traffic is not None else float(LANEOPS_DENSE)
            valu_rate = laneops * pairs / (k4_ms * 1e-3) if k4_ms > 0 else 0.0
            roofline = {"kernel": "hamming_topk_tiles", "bound": "hbm", "achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": hbm_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "launch_ms": k4_ms,
                        "launches": n_launch, "queries_per_launch": q_launch, "db_rows": info["shard_rows"], "algorithmic_bytes": alg_bytes,
                        "binding_roof": "integer VALU", "valu": {"achieved": valu_rate / 1e12, "peak": VALU_PEAK_LANEOPS / 1e12,
                                                                 "unit": "T lane-op/s", "frac": valu_rate / VALU_PEAK_LANEOPS,
                                                                 "valu_ops_per_distance": laneops},
                        "note": "vector-ALU engine (xor + popcount with partial-distance elimination): bound by integer VALU issue, data dependent"}
        if other:
            roofline["same_launch_other_engines_ms"] = {
                "vector ALU (K4), radius %d" % args.radius: other["valu"], "vector ALU, no radius bound (dense schedule)": other["valu_dense"],
                "matrix cores (K4x)": other["mfma"],
                "note": "measured after the timed region, alone on the GPU; the vector engine's elimination is data dependent (independent "
                        "bits here: its best case), the matrix engine's time is not"}
        n_fr = max(n_timed_steps * B, 1)
        out = {
            "metric": "frames/sec @ 640x480, 1M-descriptor DB; achieved HBM GB/s on BF-matcher",
            "value": args.steps * world * B / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("fp4 (+-1) x fp4 -> f32 on the matrix cores (exact integers)" if engine_used == "mfma" else "u32 (xor + popcount)") +
                     ", f32 in the verifier",
            "data": "synthetic",
            "repeats": {"timed_regions": args.repeats, "steps_each": args.steps, "value_is": "median region",
                        "frames_per_s": spread([args.steps * world * B / s for s in secs])},
            "config": {"workload": "C3: one 640x480 frame = %d ORB descriptors vs %d-descriptor DB (%d objects x 5000), "
                                   "Hamming BF k=%d, radius %d" % (nq, desc.shape[0], args.objects, k, args.radius),
                       "stages": stages, "db_rows_per_gpu": info["shard_rows"], "matcher_engine": engine_used,
                       "dataflow": "stages run concurrently on three streams and are sized as in production but are NOT data-chained in this "
                                   "figure (8(d)'s synthetic DB cannot be matched by descriptors of a synthetic image): ORB runs on the 8(d) "
                                   "image, matcher and verifier consume the frame's planted descriptors; `chained` times the real dataflow",
                       "n_ransac_iterations": args.iterations, "min_inliers": args.min_inliers,
                       "poses_per_frame_rank0": (sp.pipe.n_poses - poses0) / n_fr,
                       "frames_per_rank_per_step": B, "distinct_frames_per_rank": len(my_frames), "distinct_batches_cycled": sp.period,
                       "pipeline": "3 stages, each one batched call per step: ORB | matcher | verifier (%s)" % ("two verifier workers on alternate steps" if len(sp.vstreams) == 2 else "%d verifier worker(s)" % len(sp.vstreams)) +
                                   ("; collectives + merge on a 4th stream, overlapping the neighbouring DB passes" if overlap else ""),
                       "stage_ms_per_step": {key: 1e3 * v / max(n_timed_steps, 1) for key, v in sp.pipe.stage_s.items()},
                       "orb": "ORB-%d, 3 levels, scale 1.2 on the 8(d) synthetic image; %.0f keypoints/frame" %
                              (args.nq, (sp.pipe.n_kp - kp0) / n_fr) if "orb" in stages else None,
                       "frames_per_step": world * B,
                       "parallelism": ("DB rows sharded x%d (object aligned), %d frames per rank per step, {be} all_gather of descriptors, "
                                       "{be} %s of per-shard candidates, %s".format(be=collective_backend) % (world, B, args.exchange, "collectives + merge overlapped "
                                       "on their own stream" if overlap else "collectives in program order on the matcher's stream")) if sharded_db else
                                      ("%d replicas of the whole DB, %d frames per rank per step, no data-path collective" % (world, B)
                                       if world > 1 else "1 GPU"),
                       "sharded_step0_equals_unsharded": check},
            "roofline": roofline,
        }
        if world > 1 or use_dist:
            out["cpu_baseline"] = None                                      # timed at N = 1 only (rank 0)
        else:
            progress("headline done: %.0f frames/s" % out["value"])
            if "chained" in extras:
                out["chained"] = run_chained(torch, capi, local_rank, args)
                if "matcher_at_headline_shape" in out["chained"]:           # the dominant kernel on the data it will see
                    out["roofline"]["on_the_chained_db"] = out["chained"]["matcher_at_headline_shape"]
                progress("chained done")
            if "configs" in extras:
                out["configs"] = run_configs(torch, capi, synth, local_rank, args)
                progress("configs done")
            if "adapter" in extras:
                out["adapter_path"] = run_adapter_path(torch, capi, local_rank, desc, pts, off, frames[:8], args)
                progress("adapter path done")
            if "hbm" in extras:
                out["hbm_regime"] = run_hbm_regime(torch, capi, local_rank, args)
                progress("hbm regime done")
            if "n4" in extras:
                out["n4"] = run_n4(torch, capi, local_rank, desc, pts, off, frames[:8], args)
                progress("n4 done")
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(desc, pts, off, frames, k, args.radius, args.cpu_seconds, stages,
                                                   args.iterations, args.min_inliers)
        print(json.dumps(out))
    sp.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
