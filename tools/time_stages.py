"""Per-stage wall times on the GPU box (host-form and device-form calls), for DESIGN.md."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
desc, pts, off = synth.make_db(200)
fr = synth.make_frame(desc, pts, off, 1000, frame=1, visible_object=20)
img = synth.make_image(1)
ctx = capi.Context(0)
spans = ctx.db_load(desc, pts, off)
def timeit(f, n=30, warm=3):
    for _ in range(warm): f()
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(n): f()
    ctx.synchronize(); return (time.perf_counter() - t) / n * 1e3
print("orb host-form 640x480 ORB-1000 3 levels: %.3f ms" % timeit(lambda: ctx.orb(img, 1000, 3, 1.2)))
d_img = torch.from_numpy(img).cuda(); d_kp = torch.empty((1000, 2), device='cuda'); d_aux = torch.empty((1000, 4), device='cuda'); d_desc = torch.empty((1000, 32), dtype=torch.uint8, device='cuda')
print("orb device-form: %.3f ms" % timeit(lambda: ctx.orb_device(d_img.data_ptr(), 480, 640, 640, 1000, 3, 1.2, d_kp.data_ptr(), d_aux.data_ptr(), d_desc.data_ptr(), 1000)))
print("match host-form (PCIe inclusive) Q=1000 N=1M k=2: %.3f ms" % timeit(lambda: ctx.match(fr["q_desc"], 2, 35)))
d_q = torch.from_numpy(fr["q_desc"]).cuda(); d_c = torch.empty(1000, dtype=torch.int32, device='cuda'); d_m = torch.empty((2000, 4), dtype=torch.int32, device='cuda'); d_x = torch.empty((2000, 3), device='cuda')
print("match device-form: %.3f ms" % timeit(lambda: ctx.match_device(d_q.data_ptr(), 1000, 2, 35, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())))
row_ptr, m, xyz = ctx.match(fr["q_desc"], 2, 35)
def vh():
    rng = capi.rng_new(1); ctx.verify(fr["kp_xy"], fr["cloud"], row_ptr, m, xyz, spans, 8, 2500, 0.01, rng)
print("verify host-form (uploads clustered matches): %.3f ms" % timeit(vh, 20))
d_kpx = torch.from_numpy(fr["kp_xy"]).cuda(); d_cl = torch.from_numpy(fr["cloud"]).cuda()
def vd():
    rng = capi.rng_new(1); ctx.verify_device(d_kpx.data_ptr(), 1000, d_cl.data_ptr(), 480, 640, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr(), 2, spans, 8, 2500, 0.01, rng)
print("verify device-form: %.3f ms" % timeit(vd, 20))
