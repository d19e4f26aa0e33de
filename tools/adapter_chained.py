"""The single-frame host-buffer path (what the ecto cells call: todhip_orb + todhip_match + todhip_verify) on DATA-CHAINED frames:
DB trained from rendered views, detection views rendered, every stage consuming the previous one's host buffers."""
import os, sys, time, statistics
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, scenes
n_obj, nq, k, radius = 200, 1000, 2, 35
tex = scenes.make_textures(n_obj)
ctx = capi.Context(0)
desc, pts, off = scenes.train_db(ctx, tex, rows_per_object=5000)
spans = ctx.db_load(desc, pts, off)
bt = scenes.make_detection_batches(tex, 1, 16)[0]
imgs = bt["images"].cpu().numpy().reshape(16, 480, 640)
depth = bt["depth"].cpu().numpy().reshape(16, 480, 640).astype(np.float32)
fx, fy, cx, cy = scenes.K[0, 0], scenes.K[1, 1], scenes.K[0, 2], scenes.K[1, 2]
u, v = np.meshgrid(np.arange(640, dtype=np.float32), np.arange(480, dtype=np.float32))
clouds = []
for f in range(16):
    z = depth[f]
    c = np.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], -1).astype(np.float32)
    c[~(z > 0)] = np.nan
    clouds.append(np.ascontiguousarray(c))
t_orb, t_match, t_verify, n_pose, n_objs = [], [], [], 0, []
for rep in range(3):
    for f in range(16):
        t0 = time.perf_counter()
        kp, aux, de = ctx.orb(imgs[f], nq, 3, 1.2)
        t1 = time.perf_counter()
        row_ptr, m, xyz = ctx.match(de, k, radius)
        t2 = time.perf_counter()
        poses = ctx.verify(kp, clouds[f], row_ptr, m, xyz, spans, 8, 2500, 0.01, capi.rng_new(1))
        t3 = time.perf_counter()
        if rep:
            t_orb.append(t1 - t0); t_match.append(t2 - t1); t_verify.append(t3 - t2); n_pose += len(poses)
            n_objs.append(len(set(m["imgIdx"].tolist())))
med = lambda x: 1e3 * statistics.median(x)
tot = [a + b + c for a, b, c in zip(t_orb, t_match, t_verify)]
print("data-chained frames through host buffers: orb %.3f ms, match %.3f ms, verify %.3f ms (median; verify %.2f .. %.2f), %.0f frames/s; "
      "%.2f poses per frame, %d objects with matches per frame" % (med(t_orb), med(t_match), med(t_verify), 1e3 * min(t_verify), 1e3 * max(t_verify),
                                                                 len(tot) / sum(tot), n_pose / len(tot), statistics.median(n_objs)))
# the matcher alone on one of these frames, host and device form
kp, aux, de = ctx.orb(imgs[0], nq, 3, 1.2)
d_q = torch.from_numpy(de).cuda(); n = len(de)
d_c = torch.zeros(n, dtype=torch.int32, device='cuda'); d_m = torch.zeros((n * k, 4), dtype=torch.int32, device='cuda'); d_x = torch.zeros((n * k, 3), device='cuda')
for name, call in (("host form", lambda: ctx.match(de, k, radius)), ("device form", lambda: (ctx.match_device(d_q.data_ptr(), n, k, radius, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr()), ctx.synchronize()))):
    for _ in range(3): call()
    t0 = time.perf_counter()
    for _ in range(50): call()
    print("match %s on a chained frame (%d queries): %.3f ms" % (name, n, (time.perf_counter() - t0) / 50 * 1e3))
ctx.set_kernel_timing(True); c0 = ctx.counters()
for _ in range(20): ctx.match_device(d_q.data_ptr(), n, k, radius, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
ctx.synchronize(); c1 = ctx.counters()
print("DB pass kernel alone: %.3f ms" % ((c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / 20))
