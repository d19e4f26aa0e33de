"""todhip_verify_batch_device on the bench's 16-frame step, alone and with the matcher running beside it."""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
desc, pts, off = synth.make_db(200)
B, nq, k = int(os.environ.get("B", "16")), 1000, 2
frames = [synth.make_frame(desc, pts, off, nq, frame=f % 8, visible_object=(17 * (f % 8) + 3) % 200) for f in range(B)]
main = capi.Context(0)
spans = main.db_load(desc, pts, off)
d_q = torch.from_numpy(np.stack([f["q_desc"] for f in frames])).cuda()
d_c = torch.empty(B * nq, dtype=torch.int32, device='cuda'); d_m = torch.empty((B * nq * k, 4), dtype=torch.int32, device='cuda'); d_x = torch.empty((B * nq * k, 3), device='cuda')
main.match_device(d_q.data_ptr(), B * nq, k, 35, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr()); main.synchronize()
d_kp = torch.from_numpy(np.stack([f["kp_xy"] for f in frames])).cuda(); d_cl = torch.from_numpy(np.stack([f["cloud"] for f in frames])).cuda()
vs = torch.cuda.Stream(priority=-1)
v = capi.Context(0, vs.cuda_stream)
v.db_load(desc[:5000], pts[:5000], off[:2])
def run():
    rngs = (capi.Rng * B)(*[capi.rng_new(1) for _ in range(B)])
    return v.verify_batch_device(B, d_kp.data_ptr(), nq, d_cl.data_ptr(), 480, 640, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr(), k, spans, 8, 2500, 0.01, rngs)
stop = [False]
def matcher():
    d_c2 = torch.empty_like(d_c); d_m2 = torch.empty_like(d_m); d_x2 = torch.empty_like(d_x)
    while not stop[0]:
        main.match_device(d_q.data_ptr(), B * nq, k, 35, d_c2.data_ptr(), d_m2.data_ptr(), d_x2.data_ptr()); main.synchronize()
for with_matcher in (False, True):
    for _ in range(2): run()
    stop[0] = False
    t = threading.Thread(target=matcher)
    if with_matcher: t.start(); time.sleep(0.05)
    t0 = time.perf_counter(); n = 10
    for _ in range(n): p = run()
    dt = (time.perf_counter() - t0) / n
    stop[0] = True
    if with_matcher: t.join()
    print("verify_batch_device, %d frames, matcher running=%s: %.3f ms per batch (%d poses)" % (B, with_matcher, dt * 1e3, sum(len(x) for x in p)), flush=True)
