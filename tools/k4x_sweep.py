"""K4x launch-shape sweep: waves per CU (tile count) x bound-exchange period, on the bench's launch shape (and a single frame).
The library reads its tuning knobs once per process, so every setting runs in a child process (k4x_one.py's shape, F frames)."""
import os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    import numpy as np, torch
    from tod_amd import capi, synth
    F = int(sys.argv[2])
    desc, pts, off = synth.make_db(200)
    ctx = capi.Context(0); ctx.db_load(desc, pts, off); ctx.set_matcher_engine("mfma")
    K, R = 2, 35
    q = np.concatenate([synth.make_frame(desc, pts, off, 1000, frame=f, visible_object=(17 * f + 3) % 200)["q_desc"] for f in range(F)])
    n = len(q); d_q = torch.from_numpy(q).cuda()
    d_c = torch.zeros(n, dtype=torch.int32, device='cuda'); d_m = torch.zeros((n * K, 4), dtype=torch.int32, device='cuda'); d_x = torch.zeros((n * K, 3), device='cuda')
    call = lambda: ctx.match_device(d_q.data_ptr(), n, K, R, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
    for _ in range(2): call()
    ctx.synchronize(); ctx.set_kernel_timing(True); c0 = ctx.counters()
    for _ in range(6): call()
    ctx.synchronize(); c1 = ctx.counters()
    kern = (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / (c1.n_match_kernel_launches - c0.n_match_kernel_launches)
    print("K4x %.3f ms  (sum counts %d)" % (kern, int(d_c.sum())))
    sys.exit(0)
for F in (16, 1):
    for qt in os.environ.get("QTS", "6").split(","):
        for wpc in os.environ.get("WPCS", "12,16,20,24,32,48,64,96,128").split(","):
            for share in os.environ.get("SHARES", "4,16,64").split(","):
                env = dict(os.environ, TODHIP_K4X_WAVES_PER_CU=wpc, TODHIP_K4X_SHARE=share, TODHIP_K4X_QT=qt)
                p = subprocess.run([sys.executable, __file__, "child", str(F)], env=env, capture_output=True, text=True, timeout=300)
                print("F=%2d qt=%s wpc=%3s share=%3s: %s" % (F, qt, wpc, share, (p.stdout.strip().splitlines() or [p.stderr[-200:]])[-1]), flush=True)
