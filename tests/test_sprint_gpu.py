"""The verifier's single-wave path for small objects (tod_amd/csrc/verify_sprint.h): consecutive live objects of at most 64
matches run their RANSAC rounds, growth and invalidation on the device without a host round trip. Everything must equal the CPU
oracle draw for draw (reference: GuessGenerator.cpp:170-235, adjacency_ransac.cpp:234-309, sac_model_registration_graph.h:102-269,
ransac.h:80-143), and equal the tick-by-tick path (TODHIP_VERIFY_SPRINT=0, a child process) by construction of both."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as O
from tod_amd import capi, synth

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-3
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chained_frames.npz")


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def chained_scene(d, f, H=480, W=640):
    """Frame f of tests/golden/chained_frames.npz (this library's ORB keypoints + matcher output on rendered views against a DB
    of 200 trained objects, tools/chained_objects.py) as the host-buffer inputs of todhip_verify / the oracle."""
    K, Z = d["K"], np.float32(d["Z"])
    v, u = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    cloud = np.stack([(u - K[0, 2]) * Z / K[0, 0], (v - K[1, 2]) * Z / K[1, 1], np.full((H, W), Z, np.float32)], -1).astype(np.float32)
    cnt = d["counts"][f].astype(np.int64)
    k = d["matches"].shape[2]
    row_ptr = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
    sel = np.arange(k)[None, :] < cnt[:, None]
    m = d["matches"][f][sel]
    mm = np.zeros(len(m), capi.DMATCH_DTYPE)
    mm["queryIdx"], mm["trainIdx"], mm["imgIdx"] = m[:, 0], m[:, 1], m[:, 2]
    mm["distance"] = np.ascontiguousarray(m[:, 3]).view(np.float32)
    return dict(kp_xy=np.ascontiguousarray(d["kp"][f]), cloud=cloud, row_ptr=row_ptr, matches=mm,
                matches_xyz=np.ascontiguousarray(d["xyz"][f][sel], np.float32), spans=d["spans"])


def compare_frame(ctx, sc, min_inliers=8, n_iter=2500, err=0.01, seed=1):
    rng_o, rng_g = O.rng_new(seed), capi.rng_new(seed)
    rc, o_poses, o_rounds = O.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"],
                                     min_inliers, n_iter, err, rng_o)
    assert rc == 0
    g_poses = ctx.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], min_inliers, n_iter,
                         err, rng_g)
    g_rounds = ctx.verify_trace()
    cnt = ctx.counters()
    o_r = [r for r in o_rounds if not (r.iterations == 0 and r.draws_after == r.draws_before and r.best_count == 0)]
    g_r = [r for r in g_rounds if not (r.iterations == 0 and r.draws_after == r.draws_before)]
    got = [(g.iterations, g.best_iteration, g.best_count, g.draws_before, g.draws_after) for g in g_r]
    want = [(o.iterations, o.best_iteration, o.best_count, o.draws_before, o.draws_after) for o in o_r]
    first_bad = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), None)
    assert first_bad is None, (first_bad, got[first_bad], want[first_bad], g_r[first_bad].object)
    assert len(got) == len(want)
    assert rng_g.draws == rng_o.draws and list(rng_g.s) == list(rng_o.s) and (rng_g.f, rng_g.b) == (rng_o.f, rng_o.b)
    assert [p["object"] for p in g_poses] == [p["object"] for p in o_poses]
    for g, o in zip(g_poses, o_poses):
        assert np.array_equal(g["inliers"], o["inliers"])
        assert np.abs(g["R"] - o["R"]).max() < POSE_TOL and np.abs(g["t"] - o["t"]).max() < POSE_TOL
    return g_poses, g_r, cnt


@pytest.mark.parametrize("f", [0, 1, 2])
def test_data_chained_frames_of_200_objects(ctx, f):
    """~190 objects with >= 3 matches per frame, nearly all decided by arithmetic, 4-8 live small RANSAC problems and the object that
    is really there (340-600 matches): rounds' iterations, best iteration, best count and rand() positions equal the oracle's,
    and the small rounds ran on the device (no host round trip each)."""
    d = np.load(GOLDEN)
    sc = chained_scene(d, f)
    poses, rounds, cnt = compare_frame(ctx, sc)
    assert [p["object"] for p in poses] == [int(d["objects"][f])]
    live_small = [r for r in rounds if r.iterations > 0 and r.best_count < 64]
    assert len(live_small) >= 3 and cnt.last_sprint_rounds >= len(live_small) - 1 and cnt.last_sprint_launches <= 4
    assert cnt.last_verify_ticks <= 14


def small_objects_scene(seed, n_kp=700, n_objects=120, small=((5, 0.05), (33, 0.045), (34, 0.06), (90, 0.035)), big=(60, 0.25),
                        matches_per_kp=2):
    """Several objects that are really there with 20-45 true matches each (poses accepted by the device path: growth, invalidation,
    a second round), one of ~175 matches in between (the tick-by-tick path), and ~115 objects of random matches."""
    vis = tuple(sorted(small + (big,)))
    place = [(0.3 + 0.5 * i, (-0.36 + 0.18 * i, -0.2 + 0.1 * (i % 3), 0.9 + 0.05 * i)) for i in range(len(vis))]   # all in view
    return synth.make_verify_scene(n_kp, n_objects=n_objects, per_object=300, visible=vis, matches_per_kp=matches_per_kp, seed=seed,
                                   nan_frac=0.05, placements=place)


@pytest.mark.parametrize("seed", range(6))
def test_small_objects_with_accepted_poses(ctx, seed):
    sc = small_objects_scene(400 + seed)
    poses, rounds, cnt = compare_frame(ctx, sc, min_inliers=8, n_iter=[2500, 300, 1000][seed % 3])
    assert len(poses) >= 4 and cnt.last_sprint_rounds >= 6
    small_poses = [p for p in poses if len(p["inliers"]) <= 64]
    assert len(small_poses) >= 3


@pytest.mark.parametrize("seed,mpk,n_obj", [(0, 5, 60), (1, 3, 40), (2, 5, 200), (3, 3, 30)])
def test_hopeless_small_objects_burn_their_budget_on_the_device(ctx, seed, mpk, n_obj):
    """Objects of 15-64 random matches: sample graphs with a few triangles, thousands of iterations or 1000 failing attempts each."""
    sc = synth.make_verify_scene(500, n_objects=n_obj, per_object=200, visible=((1, 0.2),), matches_per_kp=mpk, seed=900 + seed)
    poses, rounds, cnt = compare_frame(ctx, sc, n_iter=[2500, 400][seed % 2])
    assert cnt.last_sprint_rounds >= 3


def test_short_stream_copy_is_extended_and_the_round_restarts(ctx):
    """A fresh generator state (nothing cached on the device) and objects whose rounds consume far more than the first margin."""
    sc = synth.make_verify_scene(420, n_objects=30, per_object=200, visible=((2, 0.15),), matches_per_kp=4, seed=977)
    compare_frame(ctx, sc, seed=12345)
    compare_frame(ctx, sc, seed=54321, n_iter=2500)


def test_batch_of_frames_with_small_objects_equals_frame_by_frame(ctx):
    """The batch form (device-resident inputs): frames reach their sprints and their big objects together; each frame's poses,
    traces and generator state equal the single-frame call's."""
    import torch
    d = np.load(GOLDEN)
    scs = [chained_scene(d, f) for f in range(3)] + [small_objects_scene(410), small_objects_scene(411)]
    nq = max(len(sc["kp_xy"]) for sc in scs)
    k = 2
    F, H, W = len(scs), 480, 640
    kp = np.zeros((F, nq, 2), np.float32); cloud = np.full((F, H, W, 3), np.nan, np.float32)
    counts = np.zeros((F, nq), np.uint32); mm = np.zeros((F, nq, k), capi.DMATCH_DTYPE); xyz = np.zeros((F, nq, k, 3), np.float32)
    n_objs = max(len(sc["spans"]) for sc in scs)
    spans = np.zeros(n_objs, np.float32)
    singles = []
    for f, sc in enumerate(scs):
        n = len(sc["kp_xy"])
        kp[f, :n] = sc["kp_xy"]; cloud[f] = sc["cloud"]
        rp = sc["row_ptr"].astype(np.int64)
        for q in range(n):
            c = rp[q + 1] - rp[q]
            assert c <= k
            counts[f, q] = c
            mm[f, q, :c] = sc["matches"][rp[q]:rp[q + 1]]
            xyz[f, q, :c] = sc["matches_xyz"][rp[q]:rp[q + 1]]
    # one spans table per call: the chained frames and the synthetic scenes have their own, so run them as two batches
    for group in ([0, 1, 2], [3, 4]):
        sp = scs[group[0]]["spans"]
        g = len(group)
        t_kp = torch.from_numpy(kp[group]).cuda(); t_cloud = torch.from_numpy(cloud[group]).cuda()
        t_cnt = torch.from_numpy(counts[group].astype(np.int32)).cuda()
        t_mm = torch.from_numpy(np.ascontiguousarray(mm[group]).view(np.int32).reshape(g, nq, k, 4)).cuda()
        t_xyz = torch.from_numpy(xyz[group]).cuda()
        rngs = (capi.Rng * g)(*[capi.rng_new(1) for _ in range(g)])
        poses = ctx.verify_batch_device(g, t_kp.data_ptr(), nq, t_cloud.data_ptr(), H, W, t_cnt.data_ptr(), t_mm.data_ptr(),
                                        t_xyz.data_ptr(), k, sp, 8, 2500, 0.01, rngs)
        batch_tr = [(r.object, r.iterations, r.best_iteration, r.best_count, r.draws_before, r.draws_after) for r in ctx.verify_trace(16384)]
        cnt = ctx.counters()
        assert cnt.last_sprint_rounds >= 3 * g
        single_tr = []
        for i, f in enumerate(group):
            rng1 = capi.rng_new(1)
            p1 = ctx.verify(scs[f]["kp_xy"], scs[f]["cloud"], scs[f]["row_ptr"], scs[f]["matches"], scs[f]["matches_xyz"], sp, 8, 2500,
                            0.01, rng1)
            single_tr += [(r.object, r.iterations, r.best_iteration, r.best_count, r.draws_before, r.draws_after) for r in ctx.verify_trace(16384)]
            assert [p["object"] for p in p1] == [p["object"] for p in poses[i]]
            for a, b in zip(p1, poses[i]):
                assert np.array_equal(a["inliers"], b["inliers"]) and np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"])
            assert rngs[i].draws == rng1.draws and list(rngs[i].s) == list(rng1.s)
        keep = lambda tr: [t for t in tr if not (t[1] == 0 and t[4] == t[5])]
        assert keep(batch_tr) == keep(single_tr)


def test_tick_by_tick_path_gives_the_same_results():
    """TODHIP_VERIFY_SPRINT=0 (read once per process, hence a child): the same frames through the host-driven ticks."""
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_sprint_gpu as T
from tod_amd import capi
ctx = capi.Context(0)
d = np.load(T.GOLDEN)
_, _, cnt = T.compare_frame(ctx, T.chained_scene(d, 0))
assert cnt.last_sprint_launches == 0 and cnt.last_verify_ticks > 14, (cnt.last_sprint_launches, cnt.last_verify_ticks)
_, _, cnt = T.compare_frame(ctx, T.small_objects_scene(400))
assert cnt.last_sprint_launches == 0
print("ok")
''' % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, TODHIP_VERIFY_SPRINT="0")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]
