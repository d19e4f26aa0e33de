"""TODHIP_DEBUG tick log of the verifier on a data-chained frame / batch, summarized (count and time of ticks by content)."""
import os, sys, time, re, subprocess, collections
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    import numpy as np, torch
    from tod_amd import capi, scenes
    n_obj, B, nq, k, radius = 200, int(sys.argv[2]), 1000, 2, 35
    tex = scenes.make_textures(n_obj)
    ctx = capi.Context(0)
    desc, pts, off = scenes.train_db(ctx, tex, rows_per_object=5000)
    spans = ctx.db_load(desc, pts, off)
    bt = scenes.make_detection_batches(tex, 1, 16)[0]
    kp = torch.zeros((16, nq, 2), device='cuda'); aux = torch.zeros((16, nq, 4), device='cuda'); de = torch.zeros((16, nq, 32), dtype=torch.uint8, device='cuda')
    ctx.orb_batch_device(bt["images"].data_ptr(), 16, 480 * 640, 480, 640, 640, nq, 3, 1.2, kp.data_ptr(), aux.data_ptr(), de.data_ptr(), nq)
    cnt = torch.zeros(16 * nq, dtype=torch.int32, device='cuda'); mm = torch.zeros((16 * nq * k, 4), dtype=torch.int32, device='cuda'); xx = torch.zeros((16 * nq * k, 3), device='cuda')
    ctx.match_device(de.data_ptr(), 16 * nq, k, radius, cnt.data_ptr(), mm.data_ptr(), xx.data_ptr()); ctx.synchronize()
    for rep in range(2):
        rngs = (capi.Rng * B)(*[capi.rng_new(1) for _ in range(B)])
        sys.stderr.write("=== REP %d\n" % rep); sys.stderr.flush()
        t = time.perf_counter()
        ctx.verify_batch_device(B, kp.data_ptr(), nq, 0, 480, 640, cnt.data_ptr(), mm.data_ptr(), xx.data_ptr(), k, spans, 8, 2500, 0.01, rngs, depth=(bt["depth"].data_ptr(), False, scenes.K))
        sys.stderr.write("=== TOTAL %.1f ms\n" % ((time.perf_counter() - t) * 1e3))
    sys.exit(0)
for B in (1, 16):
    env = dict(os.environ, TODHIP_DEBUG=os.environ.get("CHAINED_TICKS_LEVEL", "1"))   # 2: also rounds and draw windows (slows the host down)
    p = subprocess.run([sys.executable, __file__, "child", str(B)], env=env, capture_output=True, text=True)
    log = p.stderr.split("=== REP 1")[-1]
    if os.environ.get("CHAINED_TICKS_RAW"):
        open(os.environ["CHAINED_TICKS_RAW"] + ".B%d" % B, "w").write(log)
    ticks = [(us, re.sub(r" \(lane.*", "", what)) for us, what in re.findall(r"tick ([0-9.]+) us: (.*)", log)]
    total = re.findall(r"=== TOTAL ([0-9.]+) ms", log)
    by = collections.defaultdict(lambda: [0, 0.0])
    for us, what in ticks:
        key = " ".join(w for w, v in zip(what.split()[0::2], what.split()[1::2]) if v not in ("0", "0+0"))
        by[key][0] += 1; by[key][1] += float(us)
    print("B=%d: total %s ms, %d ticks, %.1f ms in ticks" % (B, total, len(ticks), sum(float(u) for u, _ in ticks) / 1e3))
    for key, (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
        print("   %-50s %5d ticks %9.1f ms  (%.0f us each)" % (key, n, us / 1e3, us / n))
    wins = re.findall(r"draw window: S=(\d+) len=(\d+) -> done=(\d+) pos_end=(\d+) attempts=(\d+) flag=(\d+)", log)
    print("   draw windows: %d; S histogram: %s" % (len(wins), sorted(collections.Counter(int(w[0]) for w in wins).items())[:12]))
