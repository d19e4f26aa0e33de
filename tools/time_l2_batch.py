"""The float matcher over F frames per call (F x 1000 SIFT-like queries vs the 500k-row float DB, k = 2): one DB pass for the batch.
Prints ms per call and the whole-call fraction of the bf16 MFMA roof, and checks every frame's result against its single-frame call."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
desc, pts, off = synth.make_sift_db(100)
ctx = capi.Context(0); ctx.db_load(desc, pts, off)
k = 2
for F in [int(a) for a in (sys.argv[1:] or ["1", "4", "16", "32"])]:
    q = np.concatenate([synth.make_sift_queries(desc, 1000, frame=f)[0] for f in range(F)])
    nq = len(q)
    d_q = torch.from_numpy(q).cuda(); d_c = torch.zeros(nq, dtype=torch.int32, device='cuda')
    d_m = torch.zeros((nq * k, 4), dtype=torch.int32, device='cuda'); d_x = torch.zeros((nq * k, 3), device='cuda')
    run = lambda: ctx.match_l2_device(d_q.data_ptr(), nq, k, 400.0, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
    for _ in range(3): run()
    ctx.synchronize(); t = time.perf_counter()
    for _ in range(10): run()
    ctx.synchronize(); dt = (time.perf_counter() - t) / 10
    ctx.set_kernel_timing(True); c0 = ctx.counters()
    for _ in range(5): run()
    ctx.synchronize(); c1 = ctx.counters(); ctx.set_kernel_timing(False)
    gemm_ms = (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / max(c1.n_match_kernel_launches - c0.n_match_kernel_launches, 1)
    flop = 2.0 * nq * desc.shape[0] * 128
    print("F=%2d: %.3f ms per call (%.3f ms per frame), whole call %.2f PFLOP/s = %.3f of 2.5 PF; GEMM pass %.3f ms = %.3f of 2.5 PF" %
          (F, dt * 1e3, dt * 1e3 / F, flop / dt / 1e15, flop / dt / 2.5e15, gemm_ms, flop / (gemm_ms * 1e-3) / 2.5e15), flush=True)
    if F > 1:
        cb, mb, xb = d_c.cpu().numpy().copy(), d_m.cpu().numpy().copy().reshape(F, 1000 * k, 4), d_x.cpu().numpy().copy().reshape(F, 1000 * k, 3)
        ok = True
        for f in (0, F // 2, F - 1):
            d_q1 = d_q[f * 1000:(f + 1) * 1000].contiguous(); c1_ = torch.zeros(1000, dtype=torch.int32, device='cuda')
            m1 = torch.zeros((1000 * k, 4), dtype=torch.int32, device='cuda'); x1 = torch.zeros((1000 * k, 3), device='cuda')
            ctx.match_l2_device(d_q1.data_ptr(), 1000, k, 400.0, c1_.data_ptr(), m1.data_ptr(), x1.data_ptr()); ctx.synchronize()
            cc = c1_.cpu().numpy()
            ok = ok and np.array_equal(cc, cb[f * 1000:(f + 1) * 1000])
            for qi in range(1000):
                n_ = int(cc[qi])
                ok = ok and np.array_equal(m1.cpu().numpy()[qi * k:qi * k + n_], mb[f][qi * k:qi * k + n_]) if n_ and qi % 37 == 0 else ok
        print("      frames 0, %d, %d equal their single-frame calls: %s" % (F // 2, F - 1, ok), flush=True)
