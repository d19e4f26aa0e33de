"""The 44-frame batch of tests/test_verify_gpu.py::test_batch_of_44_heavy_and_light_frames... N times in one process, against the
frame-by-frame results: prints every frame whose generator position, pose count, inliers or rotation differ (a schedule-dependent
result is a bug: nothing in run_ticks may change what a slot computes)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np, torch
from tod_amd import capi, synth
from test_verify_gpu import _pack_scene
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
k, nq = 3, 400
vis = [((1, 0.45),), ((6, 0.40), (2, 0.04)), (), ((3, 0.30), (5, 0.30)), ((7, 0.5),), ((0, 0.03), (4, 0.03), (6, 0.03)), ((2, 0.35),),
       ((1, 0.2), (3, 0.2), (5, 0.2)), ((4, 0.6),), ((0, 0.05),)]
base = [synth.make_verify_scene(nq, visible=v, seed=640 + i, matches_per_kp=3, n_objects=8) for i, v in enumerate(vis)]
packed = [_pack_scene(s, k) for s in base]
F = 44
idx = [(7 * f) % len(base) for f in range(F)]
d_kp = torch.from_numpy(np.stack([base[i]["kp_xy"] for i in idx]).astype(np.float32)).cuda()
d_cloud = torch.from_numpy(np.stack([base[i]["cloud"] for i in idx]).astype(np.float32)).cuda()
d_counts = torch.from_numpy(np.stack([packed[i][0] for i in idx])).cuda()
d_m = torch.from_numpy(np.stack([packed[i][1] for i in idx])).cuda()
d_xyz = torch.from_numpy(np.stack([packed[i][2] for i in idx])).cuda()
torch.cuda.synchronize()
spans = base[0]["spans"]
ctx = capi.Context(0)
want = []
for f in range(F):
    r = capi.rng_new(3 + f % 5)
    want.append((ctx.verify_device(d_kp[f].data_ptr(), nq, d_cloud[f].data_ptr(), 480, 640, d_counts[f].data_ptr(), d_m[f].data_ptr(),
                                   d_xyz[f].data_ptr(), k, spans, 8, 600, 0.01, r), r))
bad = 0
for rep in range(N):
    rngs = (capi.Rng * F)(*[capi.rng_new(3 + f % 5) for f in range(F)])
    got = ctx.verify_batch_device(F, d_kp.data_ptr(), nq, d_cloud.data_ptr(), 480, 640, d_counts.data_ptr(), d_m.data_ptr(), d_xyz.data_ptr(),
                                  k, spans, 8, 600, 0.01, rngs)
    for f in range(F):
        w, wr = want[f]
        why = []
        if rngs[f].draws != wr.draws: why.append("draws %d != %d" % (rngs[f].draws, wr.draws))
        if len(got[f]) != len(w): why.append("poses %d != %d" % (len(got[f]), len(w)))
        for j, (a, b) in enumerate(zip(got[f], w)):
            if a["object"] != b["object"]: why.append("pose %d object %d != %d" % (j, a["object"], b["object"]))
            elif not np.array_equal(a["inliers"], b["inliers"]): why.append("pose %d (object %d) inliers %d != %d" % (j, a["object"], len(a["inliers"]), len(b["inliers"])))
            elif not np.array_equal(a["R"], b["R"]): why.append("pose %d R differs by %.3g" % (j, np.abs(a["R"] - b["R"]).max()))
        if why:
            bad += 1
            print("rep %d frame %d (scene %d, visible %s): %s" % (rep, f, idx[f], vis[idx[f]], "; ".join(why)), flush=True)
print("%d repetitions, %d frame results differed" % (N, bad))
