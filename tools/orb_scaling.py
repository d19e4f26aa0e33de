"""ORB / verify throughput vs number of concurrent contexts, with and without the matcher running beside them."""
import os, sys, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
desc, pts, off = synth.make_db(200)
fr = synth.make_frame(desc, pts, off, 1000, frame=1, visible_object=20)
img = synth.make_image(1)
main = capi.Context(0)
spans = main.db_load(desc, pts, off)
d_q = torch.from_numpy(fr["q_desc"]).cuda(); d_c = torch.empty(1000, dtype=torch.int32, device='cuda'); d_m = torch.empty((2000, 4), dtype=torch.int32, device='cuda'); d_x = torch.empty((2000, 3), device='cuda')
main.match_device(d_q.data_ptr(), 1000, 2, 35, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr()); main.synchronize()
d_img = torch.from_numpy(img).cuda()
d_kpx = torch.from_numpy(fr["kp_xy"]).cuda(); d_cl = torch.from_numpy(fr["cloud"]).cuda()
what = sys.argv[1] if len(sys.argv) > 1 else "orb"
stop = [False]
def matcher():
    while not stop[0]:
        for _ in range(16): main.match_device(d_q.data_ptr(), 1000, 2, 35, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
        main.synchronize()
for with_matcher in (False, True):
    for n in (1, 2, 4, 6):
        ctxs = [capi.Context(0) for _ in range(n)]
        for c in ctxs: c.db_load(desc[:5000], pts[:5000], off[:2])
        outs = [(torch.empty((1000, 2), device='cuda'), torch.empty((1000, 4), device='cuda'), torch.empty((1000, 32), dtype=torch.uint8, device='cuda')) for _ in range(n)]
        iters = 60
        def work(i):
            c = ctxs[i]; o = outs[i]
            for _ in range(iters):
                if what == "orb":
                    c.orb_device(d_img.data_ptr(), 480, 640, 640, 1000, 3, 1.2, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), 1000)
                else:
                    rng = capi.rng_new(1)
                    c.verify_device(d_kpx.data_ptr(), 1000, d_cl.data_ptr(), 480, 640, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr(), 2, spans, 8, 2500, 0.01, rng)
        iters = 5
        for i in range(n): work(i)
        iters = 60
        stop[0] = False
        mt = threading.Thread(target=matcher)
        if with_matcher: mt.start()
        torch.cuda.synchronize() if not with_matcher else None
        t = time.perf_counter()
        th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        for x in th: x.start()
        for x in th: x.join()
        dt = time.perf_counter() - t
        stop[0] = True
        if with_matcher: mt.join()
        print("%s contexts=%d matcher=%s: %.0f frames/s, %.3f ms per call" % (what, n, with_matcher, n * iters / dt, dt / iters * 1e3), flush=True)
        for c in ctxs: c.close()
