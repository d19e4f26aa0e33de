"""Verifier on scenes where MANY objects collect a few distractor matches each (what a large DB with correlated
descriptors produces): every such object costs host round trips before its RANSAC gives up."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
from test_verify_gpu import _pack_scene
ctx = capi.Context(0)
for n_obj in (6, 40, 200):
    sc = synth.make_verify_scene(600, n_objects=n_obj, visible=((1, 0.30),), matches_per_kp=5, seed=77)
    k, nq = 5, 600
    counts, m, xyz = _pack_scene(sc, k)
    d_kp = torch.from_numpy(sc["kp_xy"]).cuda(); d_cl = torch.from_numpy(sc["cloud"]).cuda()
    d_c = torch.from_numpy(counts).cuda(); d_m = torch.from_numpy(m).cuda(); d_x = torch.from_numpy(xyz).cuda()
    def run():
        rng = capi.rng_new(1)
        return ctx.verify_device(d_kp.data_ptr(), nq, d_cl.data_ptr(), 480, 640, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr(), k, sc["spans"], 8, 2500, 0.01, rng)
    for _ in range(2): p = run()
    t = time.perf_counter(); n = 5
    for _ in range(n): p = run()
    dt = (time.perf_counter() - t) / n
    c = ctx.counters()
    print("%3d objects: verify_device %.2f ms per frame; objects verified %d, rounds %d, hypotheses %d, poses %d" %
          (n_obj, dt * 1e3, c.last_objects_verified, c.last_rounds, c.last_hypotheses, len(p)), flush=True)

# the same scenes as a batch of 16 frames (different generator seeds): the ticks are shared
for n_obj in (40, 200):
    sc = synth.make_verify_scene(600, n_objects=n_obj, visible=((1, 0.30),), matches_per_kp=5, seed=77)
    k, nq, F = 5, 600, 16
    counts, m, xyz = _pack_scene(sc, k)
    d_kp = torch.from_numpy(np.stack([sc["kp_xy"]] * F)).cuda(); d_cl = torch.from_numpy(np.stack([sc["cloud"]] * F)).cuda()
    d_c = torch.from_numpy(np.stack([counts] * F)).cuda(); d_m = torch.from_numpy(np.stack([m] * F)).cuda(); d_x = torch.from_numpy(np.stack([xyz] * F)).cuda()
    def runb():
        rngs = (capi.Rng * F)(*[capi.rng_new(1 + f) for f in range(F)])
        return ctx.verify_batch_device(F, d_kp.data_ptr(), nq, d_cl.data_ptr(), 480, 640, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr(), k, sc["spans"], 8, 2500, 0.01, rngs)
    p = runb()
    t = time.perf_counter(); n = 3
    for _ in range(n): p = runb()
    dt = (time.perf_counter() - t) / n
    print("%3d objects, batch of %d frames: %.1f ms per batch = %.2f ms per frame; poses %d" % (n_obj, F, dt * 1e3, dt * 1e3 / F, sum(len(x) for x in p)), flush=True)
