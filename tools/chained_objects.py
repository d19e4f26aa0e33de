"""What the objects of data-chained frames look like to the verifier: per (frame, object) the number of matches, and per RANSAC
round the iterations, best count and rand() draws consumed (todhip_verify_trace). Also saves the verifier inputs of the 16 frames
(keypoints, match lists, spans; the depth is the constant plane Z) to gpurun_out/chained_frames.npz so that the CPU oracle can
replay them without a GPU (tests/golden fixtures are cut from it by tools/make_chained_fixture.py)."""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, scenes

out_dir = os.path.join(os.path.dirname(__file__), '..', 'gpurun_out')
os.makedirs(out_dir, exist_ok=True)
n_obj, B, nq, k, radius = int(os.environ.get("OBJECTS", "200")), 16, 1000, 2, 35
tex = scenes.make_textures(n_obj)
ctx = capi.Context(0)
desc, pts, off = scenes.train_db(ctx, tex, rows_per_object=5000)
spans = ctx.db_load(desc, pts, off)
bt = scenes.make_detection_batches(tex, 1, B)[0]
kp = torch.zeros((B, nq, 2), device='cuda'); aux = torch.zeros((B, nq, 4), device='cuda'); de = torch.zeros((B, nq, 32), dtype=torch.uint8, device='cuda')
n_kp = ctx.orb_batch_device(bt["images"].data_ptr(), B, 480 * 640, 480, 640, 640, nq, 3, 1.2, kp.data_ptr(), aux.data_ptr(), de.data_ptr(), nq)
cnt = torch.zeros(B * nq, dtype=torch.int32, device='cuda'); mm = torch.zeros((B * nq * k, 4), dtype=torch.int32, device='cuda'); xx = torch.zeros((B * nq * k, 3), device='cuda')
ctx.match_device(de.data_ptr(), B * nq, k, radius, cnt.data_ptr(), mm.data_ptr(), xx.data_ptr()); ctx.synchronize()
c = cnt.cpu().numpy().reshape(B, nq); m = mm.cpu().numpy().reshape(B, nq, k, 4); x = xx.cpu().numpy().reshape(B, nq, k, 3)
np.savez_compressed(os.path.join(out_dir, "chained_frames.npz"), kp=kp.cpu().numpy(), counts=c, matches=m, xyz=x, spans=np.asarray(spans, np.float32),
                    n_kp=np.asarray(n_kp), K=scenes.K, Z=np.float32(scenes.Z), objects=np.asarray(bt["objects"]))
rows = []
for f in range(B):
    valid = np.arange(k)[None, :] < c[f][:, None]
    hist = np.bincount(m[f][valid][:, 2], minlength=n_obj)
    rng = capi.rng_new(1)
    t = time.perf_counter()
    poses = ctx.verify_device_depth(kp[f].data_ptr(), nq, bt["depth"][f].data_ptr(), False, 480, 640, scenes.K, cnt[f * nq:].data_ptr(),
                                    mm[f * nq * k:].data_ptr(), xx[f * nq * k:].data_ptr(), k, spans, 8, 2500, 0.01, rng)
    ms = (time.perf_counter() - t) * 1e3
    cc = ctx.counters()
    tr = ctx.verify_trace()
    rounds = [dict(object=int(r.object), n=int(hist[r.object]), iterations=int(r.iterations), best_it=int(r.best_iteration), best=int(r.best_count),
                   draws=int(r.draws_after - r.draws_before), kp=int(r.n_inlier_kp), accepted=int(r.accepted)) for r in tr]
    rows.append(dict(frame=f, ms=ms, true=int(bt["objects"][f]), poses=[int(p["object"]) for p in poses], hist=[int(h) for h in hist if h],
                     hyp=int(cc.last_hypotheses), gate=int(cc.last_gate_calls), draws=int(rng.draws), rounds=rounds))
    small = [r for r in rounds if r["n"] <= 64]
    live = [r for r in small if r["iterations"] > 0]
    print("frame %2d: %.1f ms, %d objects (>=3: %d, >64: %d), rounds %d; small rounds with iterations: %d, their iterations sum %d max %d, draws sum %d; "
          "triangle-free rounds %d draws %d; big rounds: %s" %
          (f, ms, (hist > 0).sum(), (hist >= 3).sum(), (hist > 64).sum(), len(rounds), len(live), sum(r["iterations"] for r in live),
           max([r["iterations"] for r in live] or [0]), sum(r["draws"] for r in live), len(small) - len(live),
           sum(r["draws"] for r in small if r["iterations"] == 0),
           [(r["object"], r["n"], r["iterations"], r["best"], r["kp"]) for r in rounds if r["n"] > 64]), flush=True)
json.dump(rows, open(os.path.join(out_dir, "chained_objects.json"), "w"))
