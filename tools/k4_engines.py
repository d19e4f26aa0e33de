"""The two engines of the exact Hamming search side by side: K4 (vector ALU, partial-distance elimination) and K4x
(matrix cores, bits as +-1 MX-fp4). For each DB kind (the 8(d) synthetic bits; optionally this repo's ORB descriptors)
and launch shape: results must be identical (counts, matches, 3D points), then the dominant kernel's time per launch.
  python tools/k4_engines.py [orb]        env: SHAPES="16000x1000000,1000x1000000,..." K=2 RADIUS=35"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import torch
from tod_amd import capi, synth

K = int(os.environ.get("K", "2"))
RADIUS = int(os.environ.get("RADIUS", "35"))
shapes = [tuple(int(v) for v in s.split("x")) for s in
          os.environ.get("SHAPES", "16000x1000000,1000x1000000,1000x100000,500x5000").split(",")]
ctx = capi.Context(0)


def orb_descs(first, count, B=50):
    out = []
    for i0 in range(first, first + count, B):
        imgs = np.stack([synth.make_image(1000 + i) for i in range(i0, i0 + B)])
        d = torch.from_numpy(imgs).cuda()
        kp = torch.empty((B, 1000, 2), device='cuda'); aux = torch.empty((B, 1000, 4), device='cuda')
        de = torch.zeros((B, 1000, 32), dtype=torch.uint8, device='cuda')
        n = ctx.orb_batch_device(d.data_ptr(), B, 480 * 640, 480, 640, 640, 1000, 3, 1.2, kp.data_ptr(), aux.data_ptr(),
                                 de.data_ptr(), 1000)
        assert min(n) == 1000
        out.append(de.cpu().numpy().reshape(-1, 32))
    return np.concatenate(out)


def run_shape(name, db, pts, off, q):
    n = len(q)
    ctx.db_load(db, pts, off)
    d_q = torch.from_numpy(q).cuda()
    res = {}
    for eng in ("valu", "mfma"):
        ctx.set_matcher_engine(eng)
        d_c = torch.zeros(n, dtype=torch.int32, device='cuda'); d_m = torch.zeros((n * K, 4), dtype=torch.int32, device='cuda')
        d_x = torch.zeros((n * K, 3), device='cuda')
        call = lambda: ctx.match_device(d_q.data_ptr(), n, K, RADIUS, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
        for _ in range(2): call()
        ctx.synchronize(); ctx.set_kernel_timing(True); c0 = ctx.counters()
        t = time.perf_counter()
        for _ in range(8): call()
        ctx.synchronize(); wall = (time.perf_counter() - t) / 8
        c1 = ctx.counters(); ctx.set_kernel_timing(False)
        kern = (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / (c1.n_match_kernel_launches - c0.n_match_kernel_launches)
        cnt = d_c.cpu().numpy(); m = d_m.cpu().numpy().reshape(n, K, 4); x = d_x.cpu().numpy().reshape(n, K, 3)
        valid = np.arange(K)[None, :] < cnt[:, None]
        res[eng] = (cnt, m[valid], x[valid], kern, wall * 1e3)
    same = all(np.array_equal(res["valu"][i], res["mfma"][i]) for i in range(3))
    pairs = n * len(db)
    print("%-28s %6d x %8d k=%d r=%d: identical=%s matches=%d | K4 %.3f ms (call %.3f) | K4x %.3f ms (call %.3f) = %.2f T pairs/s, "
          "%.2f of the 10 PF fp4 peak" % (name, n, len(db), K, RADIUS, same, int(res["valu"][0].sum()), res["valu"][3], res["valu"][4],
                                        res["mfma"][3], res["mfma"][4], pairs / res["mfma"][3] / 1e9,
                                        pairs * 512 / (res["mfma"][3] * 1e-3) / 10e15), flush=True)
    return same


ok = True
big = max(s[1] for s in shapes)
desc, pts, off = synth.make_db(max(1, big // 5000))
for nq, nrows in shapes:
    n_obj = max(1, nrows // 5000); per = min(5000, nrows)
    d, p, o = (desc[:n_obj * per], pts[:n_obj * per], off[:n_obj + 1]) if per == 5000 else synth.make_db(1, per_object=per)
    F = max(1, nq // 1000); per_f = nq // F
    q = np.concatenate([synth.make_frame(d, p, o, per_f, frame=f, visible_object=(17 * f + 3) % n_obj)["q_desc"] for f in range(F)])
    ok &= run_shape("synthetic bits", d, p, o, q)
if "orb" in sys.argv[1:]:
    t = time.time()
    db = orb_descs(0, 1000); q = orb_descs(5000, 50)[:16000]
    print("1M ORB descriptors from 1000 images in %.1f s" % (time.time() - t), flush=True)
    pts_o = np.zeros((len(db), 3), np.float32); off_o = (np.arange(201) * 5000).astype(np.uint32)
    ok &= run_shape("ORB descriptors", db, pts_o, off_o, q)
    ok &= run_shape("ORB descriptors", db, pts_o, off_o, q[:1000])
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0)
