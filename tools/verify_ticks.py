"""todhip_verify_device on one bench frame: time per call; with TODHIP_DEBUG=1 the engine prints every tick (launch lists + wall time)."""
import sys, os, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tod_amd import capi, synth
desc, pts, off = synth.make_db(200)
nq, k = 1000, 2
fr = synth.make_frame(desc, pts, off, nq, frame=0, visible_object=3)
ctx = capi.Context(0)
spans = ctx.db_load(desc, pts, off)
d_q = torch.from_numpy(fr["q_desc"]).cuda()
d_c = torch.empty(nq, dtype=torch.int32, device='cuda'); d_m = torch.empty((nq * k, 4), dtype=torch.int32, device='cuda'); d_x = torch.empty((nq * k, 3), device='cuda')
ctx.match_device(d_q.data_ptr(), nq, k, 35, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr()); ctx.synchronize()
d_kp = torch.from_numpy(fr["kp_xy"]).cuda(); d_cl = torch.from_numpy(fr["cloud"]).cuda()
def run():
    return ctx.verify_device(d_kp.data_ptr(), nq, d_cl.data_ptr(), 480, 640, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr(), k, spans, 8, 2500, 0.01, capi.rng_new(1))
for _ in range(3): run()
t=time.perf_counter()
for _ in range(20): p = run()
print("verify_device: %.3f ms, %d poses" % ((time.perf_counter()-t)/20*1e3, len(p)))
