import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
from tod_amd import capi, synth
sc = synth.make_verify_scene(300, visible=((1, 0.30),), seed=300)
ctx = capi.Context(0)
rng = capi.rng_new(1)
poses = ctx.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], 8, 500, 0.01, rng)
print("poses", [(p['object'], len(p['inliers'])) for p in poses], "draws", rng.draws)
for r in ctx.verify_trace(): print(r.object, r.iterations, r.best_iteration, r.best_count, r.draws_before, r.draws_after, r.n_inlier_kp)
