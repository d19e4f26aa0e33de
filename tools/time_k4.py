"""K4 alone on the bench's launch shape: a step's 16 frames x 1000 descriptors against the 1M-row DB."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
desc, pts, off = synth.make_db(int(os.environ.get("OBJECTS", "200")))
B, nq, k = int(os.environ.get("B", "16")), 1000, 2
q = np.concatenate([synth.make_frame(desc, pts, off, nq, frame=f, visible_object=(17 * f + 3) % 200 % int(os.environ.get("OBJECTS", "200")))["q_desc"] for f in range(B)])
ctx = capi.Context(0)
ctx.db_load(desc, pts, off)
d_q = torch.from_numpy(q).cuda(); n = B * nq
d_c = torch.empty(n, dtype=torch.int32, device='cuda'); d_m = torch.empty((n * k, 4), dtype=torch.int32, device='cuda'); d_x = torch.empty((n * k, 3), device='cuda')
RADIUS = int(os.environ.get("RADIUS", "35"))
def run(): ctx.match_device(d_q.data_ptr(), n, k, RADIUS, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
for _ in range(2): run()
ctx.synchronize(); ctx.set_kernel_timing(True); c0 = ctx.counters()
t = time.perf_counter()
for _ in range(10): run()
ctx.synchronize(); dt = (time.perf_counter() - t) / 10
c1 = ctx.counters()
k4 = (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / (c1.n_match_kernel_launches - c0.n_match_kernel_launches)
print("waves/CU=%s: match_device %d queries: %.3f ms per call, K4 %.3f ms (%.4f ms per frame)" % (os.environ.get("TODHIP_K4_WAVES_PER_CU", "24"), n, dt * 1e3, k4, k4 / B))
