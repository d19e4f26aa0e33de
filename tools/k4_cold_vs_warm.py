"""The matcher's DB pass with the DB warm in L2 / Infinity Cache (back-to-back launches) and cold (a 1 GB device fill before
every launch evicts it): SURVEY 8(d) asks for both. The kernel is VALU-bound, so the difference is what HBM adds."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
desc, pts, off = synth.make_db(200)
B, nq, k = 16, 1000, 2
q = np.concatenate([synth.make_frame(desc, pts, off, nq, frame=f, visible_object=(17 * f + 3) % 200)["q_desc"] for f in range(B)])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = capi.Context(0, stream.cuda_stream)
ctx.db_load(desc, pts, off)
d_q = torch.from_numpy(q).cuda(); n = B * nq
d_c = torch.empty(n, dtype=torch.int32, device='cuda'); d_m = torch.empty((n * k, 4), dtype=torch.int32, device='cuda'); d_x = torch.empty((n * k, 3), device='cuda')
junk = torch.empty(1 << 28, dtype=torch.float32, device='cuda')          # 1 GB: four times the Infinity Cache
def run(): ctx.match_device(d_q.data_ptr(), n, k, 35, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
acc = torch.zeros((), device='cuda')
for mb in (0, 16, 64, 256, 1024, -1024):                  # negative: read the region (no dirty lines) instead of filling it
    for _ in range(2): run()
    ctx.synchronize(); ctx.set_kernel_timing(True); c0 = ctx.counters()
    for _ in range(10):
        if mb > 0: junk[: mb << 18].fill_(1.0)
        if mb < 0: acc += junk[: (-mb) << 18].sum()
        run()
    ctx.synchronize(); c1 = ctx.counters(); ctx.set_kernel_timing(False)
    print("%5d MB device %s before every launch: K4 %.3f ms per launch" % (abs(mb), "read" if mb < 0 else "fill",
          (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / (c1.n_match_kernel_launches - c0.n_match_kernel_launches)))
