"""K4x launch-shape sweep: waves per CU (tile count) x bound-exchange period, on the bench's launch shape (and a single frame)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
desc, pts, off = synth.make_db(200)
ctx = capi.Context(0); ctx.db_load(desc, pts, off); ctx.set_matcher_engine("mfma")
K, R = 2, 35
for F in (16, 1):
    q = np.concatenate([synth.make_frame(desc, pts, off, 1000, frame=f, visible_object=(17 * f + 3) % 200)["q_desc"] for f in range(F)])
    n = len(q); d_q = torch.from_numpy(q).cuda()
    d_c = torch.zeros(n, dtype=torch.int32, device='cuda'); d_m = torch.zeros((n * K, 4), dtype=torch.int32, device='cuda'); d_x = torch.zeros((n * K, 3), device='cuda')
    for qt, wpc, share in [(a, b, c) for a in os.environ.get("QTS", "6").split(",") for b in os.environ.get("WPCS", "12,16,20,24,32,48,64,96,128").split(",")
                           for c in os.environ.get("SHARES", "4,16,64").split(",")]:
        if True:
            os.environ["TODHIP_K4X_WAVES_PER_CU"] = wpc; os.environ["TODHIP_K4X_SHARE"] = share; os.environ["TODHIP_K4X_QT"] = qt
            call = lambda: ctx.match_device(d_q.data_ptr(), n, K, R, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
            for _ in range(2): call()
            ctx.synchronize(); ctx.set_kernel_timing(True); c0 = ctx.counters()
            for _ in range(6): call()
            ctx.synchronize(); c1 = ctx.counters(); ctx.set_kernel_timing(False)
            kern = (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / (c1.n_match_kernel_launches - c0.n_match_kernel_launches)
            print("F=%2d qt=%s wpc=%3s share=%3s: K4x %.3f ms  (sum counts %d)" % (F, qt, wpc, share, kern, int(d_c.sum())), flush=True)
