"""The matcher alone on the bench's launch shape (16 frames x 1000 descriptors vs the 1M-row DB), one engine, for profilers.
  python tools/k4x_one.py [mfma|valu] [orb]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
eng = sys.argv[1] if len(sys.argv) > 1 else "mfma"
desc, pts, off = synth.make_db(200)
ctx = capi.Context(0); ctx.db_load(desc, pts, off); ctx.set_matcher_engine(eng)
K, R, F = 2, 35, int(os.environ.get("B", "16"))
q = np.concatenate([synth.make_frame(desc, pts, off, 1000, frame=f, visible_object=(17 * f + 3) % 200)["q_desc"] for f in range(F)])
n = len(q); d_q = torch.from_numpy(q).cuda()
d_c = torch.zeros(n, dtype=torch.int32, device='cuda'); d_m = torch.zeros((n * K, 4), dtype=torch.int32, device='cuda'); d_x = torch.zeros((n * K, 3), device='cuda')
call = lambda: ctx.match_device(d_q.data_ptr(), n, K, R, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
for _ in range(3): call()
ctx.synchronize(); ctx.set_kernel_timing(True); c0 = ctx.counters()
for _ in range(20): call()
ctx.synchronize(); c1 = ctx.counters()
print("%s: %.4f ms per launch over 20 launches" % (eng, (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / 20))
