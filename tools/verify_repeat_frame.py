"""One frame through todhip_verify_device N times: every run must end where the oracle ends; prints the first RANSAC round of a run
whose trace (object, iterations, best iteration, best count, rand() positions) differs from the oracle's."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np, torch
import oracle_lib as O
from tod_amd import capi, synth
from test_verify_gpu import _pack_scene
k, nq = 3, 400
scene_i = int(sys.argv[1]) if len(sys.argv) > 1 else 2
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
vis = [((1, 0.45),), ((6, 0.40), (2, 0.04)), (), ((3, 0.30), (5, 0.30)), ((7, 0.5),), ((0, 0.03), (4, 0.03), (6, 0.03)), ((2, 0.35),),
       ((1, 0.2), (3, 0.2), (5, 0.2)), ((4, 0.6),), ((0, 0.05),)]
sc = synth.make_verify_scene(nq, visible=vis[scene_i], seed=640 + scene_i, matches_per_kp=3, n_objects=8)
c, m, x = _pack_scene(sc, k)
d_kp = torch.from_numpy(sc["kp_xy"].astype(np.float32)).cuda(); d_cloud = torch.from_numpy(sc["cloud"].astype(np.float32)).cuda()
d_c = torch.from_numpy(c).cuda(); d_m = torch.from_numpy(m).cuda(); d_x = torch.from_numpy(x).cuda()
ctx = capi.Context(0)
key = lambda r: (r.iterations, r.best_iteration, r.best_count, r.draws_before, r.draws_after)
for seed in (4, 5, 6, 7):
    ro = O.rng_new(seed)
    rc, want, o_rounds = O.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], 8, 600, 0.01, ro)
    o_rounds = [key(r) for r in o_rounds if not (r.iterations == 0 and r.draws_after == r.draws_before and r.best_count == 0)]
    n_bad = 0
    for rep in range(N):
        r = capi.rng_new(seed)
        got = ctx.verify_device(d_kp.data_ptr(), nq, d_cloud.data_ptr(), 480, 640, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr(), k, sc["spans"], 8, 600, 0.01, r)
        tr = [t for t in ctx.verify_trace() if not (t.iterations == 0 and t.draws_after == t.draws_before)]
        g_rounds = [key(t) for t in tr]; g_obj = [t.object for t in tr]
        if r.draws != ro.draws or g_rounds != o_rounds:
            n_bad += 1
            for j, (g, o) in enumerate(zip(g_rounds, o_rounds)):
                if g != o:
                    print("  seed %d rep %d: round %d of %d: object %d got (iterations, best iteration, best count, draws before, after) %s, oracle %s" % (seed, rep, j, len(o_rounds), g_obj[j], g, o), flush=True)
                    break
            else:
                print("  seed %d rep %d: %d rounds, oracle %d; draws %d, oracle %d" % (seed, rep, len(g_rounds), len(o_rounds), r.draws, ro.draws), flush=True)
    cn = ctx.counters()
    print("scene %d seed %d: %d of %d runs differ from the oracle (%d rounds; sprint launches %d)" % (scene_i, seed, n_bad, N, len(o_rounds), cn.last_sprint_launches), flush=True)
