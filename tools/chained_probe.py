"""What a data-chained frame looks like to the verifier: matches per frame, objects with >= 3 matches, rounds, hypotheses,
time per frame (single and batch of 16), for the matcher radii of conf/detection.ork (35) and conf/detection.ros.ork (55)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, scenes
n_obj = int(os.environ.get("OBJECTS", "200")); B, nq, k = 16, 1000, 2  # frames: 30 % object window + clutter
tex = scenes.make_textures(n_obj)
ctx = capi.Context(0)
t = time.time(); desc, pts, off = scenes.train_db(ctx, tex, rows_per_object=5000); print("trained %d rows in %.1f s" % (off[-1], time.time() - t), flush=True)
spans = ctx.db_load(desc, pts, off)
bt = scenes.make_detection_batches(tex, 1, B)[0]
kp = torch.zeros((B, nq, 2), device='cuda'); aux = torch.zeros((B, nq, 4), device='cuda'); de = torch.zeros((B, nq, 32), dtype=torch.uint8, device='cuda')
n = ctx.orb_batch_device(bt["images"].data_ptr(), B, 480 * 640, 480, 640, 640, nq, 3, 1.2, kp.data_ptr(), aux.data_ptr(), de.data_ptr(), nq)
print("keypoints per frame:", n)
for radius in (35, 45, 55):
    cnt = torch.zeros(B * nq, dtype=torch.int32, device='cuda'); mm = torch.zeros((B * nq * k, 4), dtype=torch.int32, device='cuda'); xx = torch.zeros((B * nq * k, 3), device='cuda')
    ctx.match_device(de.data_ptr(), B * nq, k, radius, cnt.data_ptr(), mm.data_ptr(), xx.data_ptr()); ctx.synchronize()
    c = cnt.cpu().numpy().reshape(B, nq); m = mm.cpu().numpy().reshape(B, nq, k, 4)
    objs_ge3 = []
    for f in range(B):
        valid = np.arange(k)[None, :] < c[f][:, None]
        o = m[f][valid][:, 2]
        hist = np.bincount(o, minlength=n_obj)
        objs_ge3.append(int((hist >= 3).sum()))
    print("radius %d: matches/frame %.0f, objects with >= 3 matches per frame: mean %.1f max %d; matches on the true object: %.0f" %
          (radius, c.sum() / B, np.mean(objs_ge3), max(objs_ge3), np.mean([(m[f][np.arange(k)[None, :] < c[f][:, None]][:, 2] == bt["objects"][f]).sum() for f in range(B)])), flush=True)
    # single frame
    for f in (0, 1):
        rng = capi.rng_new(1)
        t = time.perf_counter()
        poses = ctx.verify_device_depth(kp[f].data_ptr(), nq, bt["depth"][f].data_ptr(), False, 480, 640, scenes.K, cnt[f * nq:].data_ptr(), mm[f * nq * k:].data_ptr(),
                                        xx[f * nq * k:].data_ptr(), k, spans, 8, 2500, 0.01, rng)
        dt = time.perf_counter() - t
        cc = ctx.counters()
        print("  frame %d alone: %.1f ms, poses %d (objects %s, true %d), objects verified %d rounds %d hypotheses %d gate calls %d draws %d" %
              (f, dt * 1e3, len(poses), [p["object"] for p in poses], bt["objects"][f], cc.last_objects_verified, cc.last_rounds, cc.last_hypotheses, cc.last_gate_calls, rng.draws), flush=True)
    rngs = (capi.Rng * B)(*[capi.rng_new(1) for _ in range(B)])
    t = time.perf_counter()
    poses = ctx.verify_batch_device(B, kp.data_ptr(), nq, 0, 480, 640, cnt.data_ptr(), mm.data_ptr(), xx.data_ptr(), k, spans, 8, 2500, 0.01, rngs, depth=(bt["depth"].data_ptr(), False, scenes.K))
    print("  batch of 16: %.1f ms; poses per frame %s" % ((time.perf_counter() - t) * 1e3, [len(p) for p in poses]), flush=True)
