"""The matcher's per-rank launch shape at world = 1, 2, 4, 8 on ONE GPU: the rank holds 1/world of the 1M-row DB and
matches the world x 16 frames of a step against it (tod_amd/sharded.py), then merges world key lists for its own 16
frames. Per-rank work is constant by construction; this shows whether the kernel time is, too."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
desc, pts, off = synth.make_db(200)
B, nq, k, radius = int(os.environ.get("B", "32")), 1000, 2, 35
frames = [synth.make_frame(desc, pts, off, nq, frame=f % 8, visible_object=(17 * (f % 8) + 3) % 200)["q_desc"] for f in range(8)]
for world in (1, 2, 4, 8):
    ctx = capi.Context(0)
    ctx.db_load(desc, pts, off, shard_rank=0, shard_count=world)
    n = world * B * nq
    q = np.concatenate([frames[f % 8] for f in range(world * B)])
    d_q = torch.from_numpy(q).cuda()
    d_keys = torch.empty((n, k), dtype=torch.int64, device='cuda')
    d_c = torch.empty(B * nq, dtype=torch.int32, device='cuda'); d_m = torch.empty((B * nq * k, 4), dtype=torch.int32, device='cuda')
    d_x = torch.empty((B * nq * k, 3), device='cuda')
    km = torch.empty((world, B * nq, k), dtype=torch.int64, device='cuda')
    def run():
        ctx.match_shard_device(d_q.data_ptr(), n, k, radius, d_keys.data_ptr())
        km[:] = d_keys.view(world, B * nq, k)[0]        # stand-in for the exchange: world lists for this rank's frames
        ctx.merge_shards_device(km.data_ptr(), world, B * nq, k, radius, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
    for _ in range(2): run()
    ctx.synchronize(); torch.cuda.synchronize(); ctx.set_kernel_timing(True); c0 = ctx.counters()
    t = time.perf_counter()
    for _ in range(10): run()
    ctx.synchronize(); torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    c1 = ctx.counters()
    k4 = (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / max(1, c1.n_match_kernel_launches - c0.n_match_kernel_launches)
    info = ctx.db_info()
    print("world=%d: shard rows %d, %d queries: step %.3f ms, K4 %.3f ms" % (world, info['shard_rows'], n, dt * 1e3, k4), flush=True)
    ctx.close()
