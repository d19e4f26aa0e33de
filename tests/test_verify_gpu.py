"""GPU parity of stage C (GuessGenerator + src/common verifier) through the C ABI, against the CPU oracle.
Integer results (adjacency rows, clique sizes, per-round iteration counts, consensus sizes, rand() stream
position, inlier keypoint lists) must be bit-exact; poses within 1e-3 (BASELINE.json north_star)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from tod_amd import capi, synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
POSE_TOL = 1e-3


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


# ------------------------------------------------------------------------------------------ clique search
@pytest.mark.parametrize("name", ["Graph1", "Graph2"])
def test_clique_reference_gtests_on_gpu(ctx, name):
    """reference test/test_maximum_clique.cpp:7-53 run on the HIP clique search"""
    spec = json.load(open(os.path.join(GOLD, "clique_reference_tests.json")))[name]
    n = spec["n"]
    add = spec["add_edges"]
    if add == "complete":
        add = [(i, j) for i in range(n) for j in range(i + 1, n)]
    dele = {tuple(sorted(e)) for e in spec["delete_edges"]}
    edges = [e for e in add if tuple(sorted(e)) not in dele]
    size, err, steps = ctx.test_clique(n, edges)
    assert err == 0 and size == spec["expected_maximum_clique_size"]


@pytest.mark.parametrize("minimal", [7, 0xFFFFFFFF])
def test_clique_random_graphs_equal_oracle(ctx, minimal):
    """sizes and step counts of FindClique on random graphs (sparse to near complete, with and without
    colour-stack under-runs) must equal the CPU restatement"""
    bad = []
    for seed in range(120):
        n = [8, 12, 20, 33, 64, 65, 100, 130, 200, 257][seed % 10]
        p = [0.1, 0.3, 0.5, 0.7, 0.9, 0.97][seed % 6]
        edges = synth.random_graph_edges(n, p, 100 + seed)
        if len(edges) == 0:
            continue
        o_size, _, _, o_steps = O.clique(n, edges, minimal_size=minimal)
        size, err, steps = ctx.test_clique(n, edges, minimal)
        if (size, steps, err) != (o_size, o_steps, 0):
            bad.append((seed, n, p, size, o_size, steps, o_steps, err))
    assert not bad, bad[:10]


def test_gate_form_of_the_search_decides_as_the_full_search_does(ctx):
    """The verifier's gate runs the search only until `clique larger than minimal_size` is decided. On random graphs whose first
    leaf of >= 7 vertices has 7, 8 or many more (densities 0.3 .. 0.97), and with every minimal size from 3 to 12: the decision
    equals the full FindClique(minimal_size)'s, the size is exact when it is <= minimal_size, and the gate walks fewer steps."""
    bad, decided_no, decided_yes, saved = [], 0, 0, 0
    for seed in range(160):
        n = [12, 20, 33, 64, 65, 100, 130, 200, 257, 400][seed % 10]
        p = [0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.97][seed % 8]
        minimal = [7, 7, 3, 5, 9, 12][seed % 6]
        edges = synth.random_graph_edges(n, p, 900 + seed)
        o_size, _, _, o_steps = O.clique(n, edges, minimal_size=minimal)
        size, err, steps = ctx.test_clique(n, edges, minimal, gate=True)
        ok = err == 0 and (size > minimal) == (o_size > minimal) and steps <= o_steps and (size == o_size if o_size <= minimal else size <= o_size)
        if not ok:
            bad.append((seed, n, p, minimal, size, o_size, steps, o_steps, err))
        decided_yes += o_size > minimal; decided_no += o_size <= minimal; saved += o_steps - steps
    assert not bad, bad[:10]
    assert decided_yes >= 40 and decided_no >= 20 and saved > 0, (decided_yes, decided_no, saved)


# ------------------------------------------------------------------------------------------ FillAdjacency
def _clusters_of(sc):
    """ClusterPerObject on the host (plain gather) -> {obj: (train, query, qidx)}"""
    out = {}
    kp, cloud = sc["kp_xy"], sc["cloud"]
    for qi in range(len(kp)):
        p = cloud[int(kp[qi, 1]), int(kp[qi, 0])]
        if np.isnan(p[0]):
            continue
        for m in range(sc["row_ptr"][qi], sc["row_ptr"][qi + 1]):
            o = int(sc["matches"][m]["imgIdx"])
            t, q, i = out.setdefault(o, ([], [], []))
            t.append(sc["matches_xyz"][m]); q.append(p); i.append(qi)
    return {o: (np.array(t, np.float32), np.array(q, np.float32), np.array(i, np.uint32)) for o, (t, q, i) in out.items()}


@pytest.mark.parametrize("seed,n_kp", [(1, 120), (2, 400), (3, 900)])
def test_adjacency_rows_bit_exact(ctx, seed, n_kp):
    sc = synth.make_verify_scene(n_kp, visible=((1, 0.35), (3, 0.15)), seed=seed)
    for obj, (t, q, qi) in _clusters_of(sc).items():
        if len(qi) < 2:
            continue
        cl = O.Cluster(t, q, qi)
        cl.fill(sc["kp_xy"], float(sc["spans"][obj]), 0.01)
        phys, samp = ctx.test_adjacency(t, q, sc["kp_xy"][qi], float(sc["spans"][obj]), 0.01)
        assert np.array_equal(phys, cl.bits(0)), ("physical", obj)
        assert np.array_equal(samp, cl.bits(1)), ("sample", obj)


def test_adjacency_nan_points_follow_reference_comparisons(ctx):
    """NaN y/z coordinates pass the reference's `>` rejections (only .x is tested in ClusterPerObject)"""
    sc = synth.make_verify_scene(200, seed=9)
    t, q, qi = _clusters_of(sc)[1]
    q = q.copy(); t = t.copy()
    q[3, 1] = np.nan; q[10, 2] = np.inf; t[20, 0] = np.nan
    cl = O.Cluster(t, q, qi)
    cl.fill(sc["kp_xy"], float(sc["spans"][1]), 0.01)
    phys, samp = ctx.test_adjacency(t, q, sc["kp_xy"][qi], float(sc["spans"][1]), 0.01)
    assert np.array_equal(phys, cl.bits(0)) and np.array_equal(samp, cl.bits(1))


# ------------------------------------------------------------------------------------------ whole frames
def _compare_frame(ctx, sc, min_inliers, n_iter, err=0.01, seed=1, max_poses=64):
    rng_o = O.rng_new(seed)
    rc, o_poses, o_rounds = O.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"],
                                     sc["spans"], min_inliers, n_iter, err, rng_o, max_poses=max_poses)
    assert rc == 0
    rng_g = capi.rng_new(seed)
    g_poses = ctx.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"],
                         min_inliers, n_iter, err, rng_g, max_poses=max_poses)
    g_rounds = ctx.verify_trace()
    # the oracle also traces rounds of objects with < 3 matches (no draws, no iterations); the GPU path skips them
    o_rounds = [r for r in o_rounds if not (r.iterations == 0 and r.draws_after == r.draws_before and r.best_count == 0)]
    g_rounds = [r for r in g_rounds if not (r.iterations == 0 and r.draws_after == r.draws_before)]
    assert len(g_rounds) == len(o_rounds)
    for g, o in zip(g_rounds, o_rounds):
        assert (g.iterations, g.best_iteration, g.best_count, g.draws_before, g.draws_after) == \
               (o.iterations, o.best_iteration, o.best_count, o.draws_before, o.draws_after)
    assert rng_g.draws == rng_o.draws and list(rng_g.s) == list(rng_o.s) and (rng_g.f, rng_g.b) == (rng_o.f, rng_o.b)
    assert len(g_poses) == len(o_poses)
    for g, o in zip(g_poses, o_poses):
        assert g["object"] == o["object"]
        assert np.array_equal(g["inliers"], o["inliers"])
        assert np.abs(g["R"] - o["R"]).max() < POSE_TOL and np.abs(g["t"] - o["t"]).max() < POSE_TOL
    return g_poses, g_rounds


def test_frame_single_object(ctx):
    sc = synth.make_verify_scene(300, visible=((1, 0.30),), seed=300)
    poses, rounds = _compare_frame(ctx, sc, 8, 500)
    assert len(poses) == 1 and poses[0]["object"] == 1
    R, t = sc["poses"][1]
    assert np.abs(poses[0]["R"] - R).max() < 0.05 and np.abs(poses[0]["t"] - t).max() < 0.02


def test_frame_two_objects_many_rounds(ctx):
    sc = synth.make_verify_scene(500, visible=((1, 0.30), (4, 0.20)), seed=500)
    poses, rounds = _compare_frame(ctx, sc, 8, 500)
    assert sorted(p["object"] for p in poses) == [1, 4] and len(rounds) >= 6


@pytest.mark.parametrize("seed", range(6))
def test_frame_variants(ctx, seed):
    cfgs = [dict(n_kp=150, visible=((0, 0.5),), matches_per_kp=2),
            dict(n_kp=250, visible=((2, 0.2), (5, 0.2), (3, 0.2)), matches_per_kp=3),
            dict(n_kp=400, visible=((1, 0.4),), matches_per_kp=5, true_match_rank=4),
            dict(n_kp=80, visible=(), matches_per_kp=5),                 # clutter only: no pose, many failed draws
            dict(n_kp=600, visible=((1, 0.25),), matches_per_kp=1, n_objects=2),
            dict(n_kp=350, visible=((1, 0.3),), matches_per_kp=5, noise=0.006)]
    sc = synth.make_verify_scene(seed=40 + seed, **cfgs[seed])
    _compare_frame(ctx, sc, [8, 6, 15, 8, 8, 8][seed], [300, 1000, 200, 100, 400, 300][seed])


def test_c1_shape_frame_via_matcher(ctx):
    """BASELINE configs[0] end to end: ORB-500 frame, 1-object DB, k=5, radius 35, iterations 2500, min_inliers 8"""
    desc, pts, off = synth.make_db(1)
    fr = synth.make_frame(desc, pts, off, 500)
    spans = ctx.db_load(desc, pts, off)
    row_ptr, m, xyz = ctx.match(fr["q_desc"], 5, 35)
    sc = dict(kp_xy=fr["kp_xy"], cloud=fr["cloud"], row_ptr=row_ptr, matches=m, matches_xyz=xyz, spans=spans)
    poses, rounds = _compare_frame(ctx, sc, 8, 2500)
    assert len(poses) == 1
    assert np.abs(poses[0]["R"] - synth.pose_R()).max() < 0.03 and np.abs(poses[0]["t"] - synth.POSE_T).max() < 0.01


def test_empty_and_degenerate_inputs(ctx):
    sc = synth.make_verify_scene(50, seed=1)
    rng = capi.rng_new(1)
    # no point cloud: the 2D-only branch is an empty TODO in the reference (GuessGenerator.cpp:147-152)
    assert ctx.verify(sc["kp_xy"], np.zeros((0, 0, 3), np.float32), sc["row_ptr"], sc["matches"], sc["matches_xyz"],
                      sc["spans"], 8, 100, 0.01, rng) == []
    assert rng.draws == 0
    # no matches at all
    rp = np.zeros(51, np.uint32)
    assert ctx.verify(sc["kp_xy"], sc["cloud"], rp, sc["matches"][:0], sc["matches_xyz"][:0], sc["spans"], 8, 100,
                      0.01, rng) == []
    # keypoint outside the cloud
    kp = sc["kp_xy"].copy(); kp[0, 0] = 10000
    with pytest.raises(capi.TodError) as e:
        ctx.verify(kp, sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], 8, 100, 0.01, rng)
    assert e.value.status == capi.ERANGE


# ------------------------------------------------------------------------------------------ device-resident form
def _verify_device_from_scene(ctx, sc, k, min_inliers, n_iter, seed=1):
    import torch
    nq = len(sc["kp_xy"])
    counts = np.diff(sc["row_ptr"].astype(np.int64)).astype(np.int32)
    m = np.zeros((nq, k), capi.DMATCH_DTYPE)
    xyz = np.zeros((nq, k, 3), np.float32)
    for q in range(nq):
        lo, hi = int(sc["row_ptr"][q]), int(sc["row_ptr"][q + 1])
        m[q, :hi - lo] = sc["matches"][lo:hi]
        xyz[q, :hi - lo] = sc["matches_xyz"][lo:hi]
    d_kp = torch.from_numpy(np.ascontiguousarray(sc["kp_xy"], np.float32)).cuda()
    d_cloud = torch.from_numpy(np.ascontiguousarray(sc["cloud"], np.float32)).cuda()
    d_counts = torch.from_numpy(counts).cuda()
    d_m = torch.from_numpy(m.view(np.int32).reshape(nq * k, 4).copy()).cuda()
    d_xyz = torch.from_numpy(xyz.reshape(nq * k, 3)).cuda()
    torch.cuda.synchronize()
    rng = capi.rng_new(seed)
    H, W = sc["cloud"].shape[:2]
    poses = ctx.verify_device(d_kp.data_ptr(), nq, d_cloud.data_ptr(), H, W, d_counts.data_ptr(), d_m.data_ptr(),
                              d_xyz.data_ptr(), k, sc["spans"], min_inliers, n_iter, 0.01, rng)
    return poses, rng, ctx.verify_trace()


@pytest.mark.parametrize("cfg", [dict(n_kp=300, visible=((1, 0.30),), seed=300, k=5),
                                 dict(n_kp=500, visible=((1, 0.30), (4, 0.20)), seed=500, k=5),
                                 dict(n_kp=250, visible=((2, 0.2), (5, 0.2), (3, 0.2)), seed=41, k=3, matches_per_kp=3)])
def test_device_form_equals_host_form(ctx, cfg):
    cfg = dict(cfg)
    k = cfg.pop("k")
    sc = synth.make_verify_scene(**cfg)
    rng_h = capi.rng_new(1)
    host = ctx.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], 8, 400,
                      0.01, rng_h)
    host_tr = [(r.object, r.iterations, r.best_iteration, r.best_count, r.draws_after) for r in ctx.verify_trace()]
    dev, rng_d, dev_tr = _verify_device_from_scene(ctx, sc, k, 8, 400)
    assert [(r.object, r.iterations, r.best_iteration, r.best_count, r.draws_after) for r in dev_tr] == host_tr
    assert rng_d.draws == rng_h.draws and list(rng_d.s) == list(rng_h.s)
    assert len(dev) == len(host) and len(host) >= 1
    for a, b in zip(dev, host):
        assert a["object"] == b["object"] and np.array_equal(a["inliers"], b["inliers"])
        assert np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"])      # same kernels, same inputs


def test_device_pipeline_match_then_verify(ctx):
    """match_device -> verify_device without leaving HBM == oracle on the same frame (C1 shape)"""
    import torch
    desc, pts, off = synth.make_db(2, per_object=3000)
    fr = synth.make_frame(desc, pts, off, 500, frame=4, visible_object=1)
    spans = ctx.db_load(desc, pts, off)
    nq, k = 500, 5
    d_q = torch.from_numpy(fr["q_desc"]).cuda()
    d_counts = torch.empty(nq, dtype=torch.int32, device="cuda")
    d_m = torch.empty((nq * k, 4), dtype=torch.int32, device="cuda")
    d_xyz = torch.empty((nq * k, 3), dtype=torch.float32, device="cuda")
    d_kp = torch.from_numpy(fr["kp_xy"]).cuda()
    d_cloud = torch.from_numpy(fr["cloud"]).cuda()
    torch.cuda.synchronize()
    ctx.match_device(d_q.data_ptr(), nq, k, 35, d_counts.data_ptr(), d_m.data_ptr(), d_xyz.data_ptr())
    rng = capi.rng_new(1)
    poses = ctx.verify_device(d_kp.data_ptr(), nq, d_cloud.data_ptr(), 480, 640, d_counts.data_ptr(), d_m.data_ptr(),
                              d_xyz.data_ptr(), k, spans, 8, 2500, 0.01, rng)
    rc, row_ptr, m, xyz = O.match(desc, off, pts, fr["q_desc"], k, 35)
    rng_o = O.rng_new(1)
    rc, o_poses, _ = O.verify(fr["kp_xy"], fr["cloud"], row_ptr, m, xyz, O.spans(pts, off), 8, 2500, 0.01, rng_o)
    assert rng.draws == rng_o.draws and len(poses) == len(o_poses) == 1
    assert poses[0]["object"] == o_poses[0]["object"] == 1 and np.array_equal(poses[0]["inliers"], o_poses[0]["inliers"])
    assert np.abs(poses[0]["R"] - o_poses[0]["R"]).max() < POSE_TOL and np.abs(poses[0]["t"] - o_poses[0]["t"]).max() < POSE_TOL


def test_two_rank_step_in_process(ctx):
    """tod_amd/sharded.py driven with two contexts on one GPU and in-process 'collectives': every rank's frame
    must come out exactly as on a single device."""
    import torch
    from tod_amd import sharded
    world, nq, k, radius = 2, 400, 2, 35
    desc, pts, off = synth.make_db(6, per_object=2000)
    frames = [synth.make_frame(desc, pts, off, nq, frame=20 + r, visible_object=(1, 4)[r]) for r in range(world)]
    ctxs = [capi.Context(0) for _ in range(world)]
    spans = [c.db_load(desc, pts, off, shard_rank=r, shard_count=world) for r, c in enumerate(ctxs)][0]
    d_q = [torch.from_numpy(f["q_desc"]).cuda() for f in frames]
    q_all = torch.stack(d_q)                                            # what all_gather of descriptors yields
    keys = []
    for r in range(world):
        kk = torch.empty((world * nq, k), dtype=torch.int64, device="cuda")
        ctxs[r].match_shard_device(q_all.data_ptr(), world * nq, k, radius, kk.data_ptr())
        ctxs[r].synchronize()
        keys.append(kk.reshape(world, nq, k))
    keys_all = torch.stack(keys)                                        # [shard][frame][Q][k]
    single = capi.Context(0)
    single.db_load(desc, pts, off)
    for r in range(world):
        mine = keys_all[:, r].contiguous()
        d_counts = torch.empty(nq, dtype=torch.int32, device="cuda")
        d_m = torch.empty((nq * k, 4), dtype=torch.int32, device="cuda")
        d_xyz = torch.empty((nq * k, 3), dtype=torch.float32, device="cuda")
        ctxs[r].merge_shards_device(mine.data_ptr(), world, nq, k, radius, d_counts.data_ptr(), d_m.data_ptr(),
                                    d_xyz.data_ptr())
        d_kp = torch.from_numpy(frames[r]["kp_xy"]).cuda()
        d_cloud = torch.from_numpy(frames[r]["cloud"]).cuda()
        torch.cuda.synchronize()
        rng = capi.rng_new(1)
        poses = ctxs[r].verify_device(d_kp.data_ptr(), nq, d_cloud.data_ptr(), 480, 640, d_counts.data_ptr(),
                                      d_m.data_ptr(), d_xyz.data_ptr(), k, spans, 8, 2500, 0.01, rng)
        row_ptr, m, xyz = single.match(frames[r]["q_desc"], k, radius)
        assert np.array_equal(d_counts.cpu().numpy(), np.diff(row_ptr.astype(np.int64)))
        rng1 = capi.rng_new(1)
        want = single.verify(frames[r]["kp_xy"], frames[r]["cloud"], row_ptr, m, xyz, spans, 8, 2500, 0.01, rng1)
        assert len(poses) == len(want) == 1 and poses[0]["object"] == (1, 4)[r]
        assert np.array_equal(poses[0]["inliers"], want[0]["inliers"]) and np.array_equal(poses[0]["R"], want[0]["R"])
    for c in ctxs + [single]:
        c.close()


def test_large_dense_object_uses_big_lds_pass(ctx):
    """> 330 mutually consistent matches on one object: the induced graph no longer fits the 48 KB LDS carve of the
    first evaluation pass and goes through the deferred 160 KB pass; results must still equal the oracle"""
    sc = synth.make_verify_scene(1300, n_objects=2, per_object=1000, visible=((1, 0.55),), matches_per_kp=1, seed=77,
                                 nan_frac=0.0)
    poses, rounds = _compare_frame(ctx, sc, 8, 60)
    assert len(poses) == 1 and len(poses[0]["inliers"]) > 500


@pytest.mark.parametrize("n_kp,per_object,frac,seed", [(1100, 900, 0.55, 81), (1500, 1200, 0.60, 82), (1800, 1400, 0.55, 83)])
def test_objects_of_513_to_1024_matches_take_the_wide_register_paths(ctx, n_kp, per_object, frac, seed):
    """600 / 900 / ~1000 consistent matches on one object (9 .. 16 words of 64 vertices): eval_kernel<true> -- the object's vertex
    numbers kept, 32-register class tuples in the colouring, DegreeSort in registers -- must equal the oracle like every other size"""
    sc = synth.make_verify_scene(n_kp, n_objects=2, per_object=per_object, visible=((1, frac),), matches_per_kp=1, seed=seed, nan_frac=0.0)
    poses, rounds = _compare_frame(ctx, sc, 8, 40)
    assert len(poses) == 1 and 512 < len(poses[0]["inliers"]) <= 1024, len(poses[0]["inliers"])


def test_very_large_object_falls_back_to_global_adjacency(ctx):
    """~1500 consistent matches: the induced graph exceeds one CU's LDS, so the gate keeps its adjacency matrix in
    global scratch (third tier). Slow, but the result must still equal the oracle."""
    sc = synth.make_verify_scene(2600, n_objects=2, per_object=2000, visible=((1, 0.6),), matches_per_kp=1, seed=78,
                                 nan_frac=0.0)
    poses, rounds = _compare_frame(ctx, sc, 8, 12)
    assert len(poses) == 1 and len(poses[0]["inliers"]) > 1400


@pytest.mark.parametrize("u16", [False, True])
def test_depth_image_form_equals_cloud_form(ctx, u16):
    """N3: back-projecting only the keypoints from the depth image == looking them up in the full cloud that the
    same back-projection produces (cv::depthTo3d convention), for float metres and uint16 millimetres"""
    import torch
    desc, pts, off = synth.make_db(2, per_object=3000)
    fr = synth.make_frame(desc, pts, off, 500, frame=6, visible_object=0)
    H, W, f = 480, 640, 525.0
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float32)
    z = fr["cloud"][:, :, 2].copy()
    if u16:
        d16 = np.where(np.isnan(z), 0, np.rint(z * 1000.0)).astype(np.uint16)
        z = np.where(d16 == 0, np.nan, d16.astype(np.float32) * np.float32(0.001)).astype(np.float32)
        depth = d16
    else:
        depth = z
    u, v = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    cloud = np.stack([(u - K[0, 2]) * z / K[0, 0], (v - K[1, 2]) * z / K[1, 1], z], axis=2).astype(np.float32)
    spans = ctx.db_load(desc, pts, off)
    nq, k = 500, 5
    d_q = torch.from_numpy(fr["q_desc"]).cuda()
    d_counts = torch.empty(nq, dtype=torch.int32, device="cuda")
    d_m = torch.empty((nq * k, 4), dtype=torch.int32, device="cuda")
    d_xyz = torch.empty((nq * k, 3), dtype=torch.float32, device="cuda")
    d_kp = torch.from_numpy(fr["kp_xy"]).cuda()
    d_cloud = torch.from_numpy(cloud).cuda()
    d_depth = torch.from_numpy(depth.view(np.int16) if u16 else depth).cuda()
    torch.cuda.synchronize()
    ctx.match_device(d_q.data_ptr(), nq, k, 35, d_counts.data_ptr(), d_m.data_ptr(), d_xyz.data_ptr())
    r1, r2 = capi.rng_new(1), capi.rng_new(1)
    a = ctx.verify_device(d_kp.data_ptr(), nq, d_cloud.data_ptr(), H, W, d_counts.data_ptr(), d_m.data_ptr(),
                          d_xyz.data_ptr(), k, spans, 8, 1000, 0.01, r1)
    b = ctx.verify_device_depth(d_kp.data_ptr(), nq, d_depth.data_ptr(), u16, H, W, K, d_counts.data_ptr(),
                                d_m.data_ptr(), d_xyz.data_ptr(), k, spans, 8, 1000, 0.01, r2)
    assert len(a) == len(b) == 1 and r1.draws == r2.draws
    assert np.array_equal(a[0]["inliers"], b[0]["inliers"]) and np.array_equal(a[0]["R"], b[0]["R"]) and np.array_equal(a[0]["t"], b[0]["t"])


def _pack_scene(sc, k):
    nq = len(sc["kp_xy"])
    counts = np.diff(sc["row_ptr"].astype(np.int64)).astype(np.int32)
    m = np.zeros((nq, k), capi.DMATCH_DTYPE)
    xyz = np.zeros((nq, k, 3), np.float32)
    for q in range(nq):
        lo, hi = int(sc["row_ptr"][q]), int(sc["row_ptr"][q + 1])
        m[q, :hi - lo] = sc["matches"][lo:hi]
        xyz[q, :hi - lo] = sc["matches_xyz"][lo:hi]
    return counts, m.view(np.int32).reshape(nq * k, 4).copy(), xyz.reshape(nq * k, 3)


def test_batch_of_frames_equals_frame_by_frame(ctx):
    """todhip_verify_batch_device: frames at different points of their RANSAC state machines share launches; every
    frame's poses, inliers, traces and generator state must equal the single-frame call's. Frames: one object, two
    objects (several rounds each), three objects, and one frame without any match."""
    import torch
    k, nq, seed = 5, 400, 500
    vis = [((1, 0.30),), ((1, 0.30), (4, 0.20)), ((2, 0.2), (5, 0.2), (3, 0.2)), ((0, 0.5),)]
    scenes = [synth.make_verify_scene(nq, visible=v, seed=seed) for v in vis]
    assert all(np.array_equal(s["spans"], scenes[0]["spans"]) for s in scenes)
    packed = [_pack_scene(s, k) for s in scenes]
    packed.append((np.zeros(nq, np.int32), packed[0][1], packed[0][2]))            # a frame whose queries matched nothing
    scenes.append(scenes[0])
    F = len(scenes)
    d_kp = torch.from_numpy(np.stack([s["kp_xy"] for s in scenes]).astype(np.float32)).cuda()
    d_cloud = torch.from_numpy(np.stack([s["cloud"] for s in scenes]).astype(np.float32)).cuda()
    d_counts = torch.from_numpy(np.stack([p[0] for p in packed])).cuda()
    d_m = torch.from_numpy(np.stack([p[1] for p in packed])).cuda()
    d_xyz = torch.from_numpy(np.stack([p[2] for p in packed])).cuda()
    torch.cuda.synchronize()
    spans = scenes[0]["spans"]
    want, want_rng, want_tr = [], [], []
    for f in range(F):
        r = capi.rng_new(1 + f)
        want.append(ctx.verify_device(d_kp[f].data_ptr(), nq, d_cloud[f].data_ptr(), 480, 640, d_counts[f].data_ptr(),
                                      d_m[f].data_ptr(), d_xyz[f].data_ptr(), k, spans, 8, 400, 0.01, r))
        want_rng.append((r.draws, list(r.s)))
        want_tr += [(t.object, t.iterations, t.best_iteration, t.best_count, t.draws_after) for t in ctx.verify_trace()]
    rngs = (capi.Rng * F)(*[capi.rng_new(1 + f) for f in range(F)])
    got = ctx.verify_batch_device(F, d_kp.data_ptr(), nq, d_cloud.data_ptr(), 480, 640, d_counts.data_ptr(), d_m.data_ptr(),
                                  d_xyz.data_ptr(), k, spans, 8, 400, 0.01, rngs)
    got_tr = [(t.object, t.iterations, t.best_iteration, t.best_count, t.draws_after) for t in ctx.verify_trace()]
    assert got_tr == want_tr
    assert [len(p) for p in got] == [len(p) for p in want] and min(len(p) for p in got[:4]) >= 1 and got[4] == []
    for f in range(F):
        assert (rngs[f].draws, list(rngs[f].s)) == want_rng[f]
        for a, b in zip(got[f], want[f]):
            assert a["object"] == b["object"] and np.array_equal(a["inliers"], b["inliers"])
            assert np.array_equal(a["R"], b["R"]) and np.array_equal(a["t"], b["t"])


def test_batch_depth_form_after_batched_matching(ctx):
    """The bench path: one match_device call for F x Q queries, then verify_batch_device_depth on its outputs ==
    the oracle frame by frame."""
    import torch
    desc, pts, off = synth.make_db(6, per_object=2000)
    F, nq, k, H, W, f = 3, 500, 2, 480, 640, 525.0
    frames = [synth.make_frame(desc, pts, off, nq, frame=30 + i, visible_object=(1, 4, 2)[i]) for i in range(F)]
    K = np.array([[f, 0, W / 2.0], [0, f, H / 2.0], [0, 0, 1]], np.float32)
    depth = np.stack([fr["cloud"][:, :, 2] for fr in frames]).astype(np.float32)
    u, v = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    spans = ctx.db_load(desc, pts, off)
    d_q = torch.from_numpy(np.stack([fr["q_desc"] for fr in frames])).cuda()
    d_counts = torch.empty(F * nq, dtype=torch.int32, device="cuda")
    d_m = torch.empty((F * nq * k, 4), dtype=torch.int32, device="cuda")
    d_xyz = torch.empty((F * nq * k, 3), dtype=torch.float32, device="cuda")
    d_kp = torch.from_numpy(np.stack([fr["kp_xy"] for fr in frames])).cuda()
    d_depth = torch.from_numpy(depth).cuda()
    torch.cuda.synchronize()
    ctx.match_device(d_q.data_ptr(), F * nq, k, 35, d_counts.data_ptr(), d_m.data_ptr(), d_xyz.data_ptr())
    rngs = (capi.Rng * F)(*[capi.rng_new(1) for _ in range(F)])
    got = ctx.verify_batch_device(F, d_kp.data_ptr(), nq, 0, H, W, d_counts.data_ptr(), d_m.data_ptr(), d_xyz.data_ptr(), k,
                                  spans, 8, 2500, 0.01, rngs, depth=(d_depth.data_ptr(), False, K))
    for i, fr in enumerate(frames):
        z = depth[i]
        cloud = np.stack([(u - K[0, 2]) * z / K[0, 0], (v - K[1, 2]) * z / K[1, 1], z], axis=2).astype(np.float32)
        rc, row_ptr, m, xyz = O.match(desc, off, pts, fr["q_desc"], k, 35)
        rng_o = O.rng_new(1)
        rc, o_poses, _ = O.verify(fr["kp_xy"], cloud, row_ptr, m, xyz, O.spans(pts, off), 8, 2500, 0.01, rng_o)
        assert rngs[i].draws == rng_o.draws and len(got[i]) == len(o_poses) == 1
        assert got[i][0]["object"] == o_poses[0]["object"] == (1, 4, 2)[i]
        assert np.array_equal(got[i][0]["inliers"], o_poses[0]["inliers"])
        assert np.abs(got[i][0]["R"] - o_poses[0]["R"]).max() < POSE_TOL and np.abs(got[i][0]["t"] - o_poses[0]["t"]).max() < POSE_TOL


def test_batch_larger_than_one_launch_group(ctx):
    """More frames than argument sets fit one launch (16): the engine splits every kernel list into several launches.
    18 frames (the same 3 scenes, different generators) == frame by frame."""
    import torch
    k, nq, seed = 3, 200, 41
    vis = [((2, 0.3),), ((5, 0.3), (3, 0.25)), ((1, 0.4),)]
    base = [synth.make_verify_scene(nq, visible=v, seed=seed, matches_per_kp=3) for v in vis]
    packed = [_pack_scene(s, k) for s in base]
    F = 18
    idx = [f % 3 for f in range(F)]
    d_kp = torch.from_numpy(np.stack([base[i]["kp_xy"] for i in idx]).astype(np.float32)).cuda()
    d_cloud = torch.from_numpy(np.stack([base[i]["cloud"] for i in idx]).astype(np.float32)).cuda()
    d_counts = torch.from_numpy(np.stack([packed[i][0] for i in idx])).cuda()
    d_m = torch.from_numpy(np.stack([packed[i][1] for i in idx])).cuda()
    d_xyz = torch.from_numpy(np.stack([packed[i][2] for i in idx])).cuda()
    torch.cuda.synchronize()
    spans = base[0]["spans"]
    want = []
    for f in range(F):
        r = capi.rng_new(1 + f // 3)
        want.append((ctx.verify_device(d_kp[f].data_ptr(), nq, d_cloud[f].data_ptr(), 480, 640, d_counts[f].data_ptr(),
                                       d_m[f].data_ptr(), d_xyz[f].data_ptr(), k, spans, 8, 300, 0.01, r), r.draws))
    rngs = (capi.Rng * F)(*[capi.rng_new(1 + f // 3) for f in range(F)])
    got = ctx.verify_batch_device(F, d_kp.data_ptr(), nq, d_cloud.data_ptr(), 480, 640, d_counts.data_ptr(), d_m.data_ptr(),
                                  d_xyz.data_ptr(), k, spans, 8, 300, 0.01, rngs)
    assert sum(len(p) for p in got) >= F
    for f in range(F):
        assert rngs[f].draws == want[f][1] and len(got[f]) == len(want[f][0])
        for a, b in zip(got[f], want[f][0]):
            assert a["object"] == b["object"] and np.array_equal(a["inliers"], b["inliers"]) and np.array_equal(a["R"], b["R"])


def test_batch_of_heavy_and_light_frames_equals_frame_by_frame(ctx):
    """Frames that reach their big object at different ticks, frames with nothing but small objects and an empty frame in one
    batch: the heavy phases leave the lock-step for side streams (Engine::run_ticks) while the others keep ticking. Whatever the
    schedule, every frame's poses, consensus sets and generator must equal the single-frame call's."""
    import torch
    k, nq = 3, 400
    vis = [((1, 0.45),), ((6, 0.40), (2, 0.04)), (), ((3, 0.30), (5, 0.30)), ((7, 0.5),), ((0, 0.03), (4, 0.03), (6, 0.03)), ((2, 0.35),), ()]
    base = [synth.make_verify_scene(nq, visible=v, seed=600 + i, matches_per_kp=3, n_objects=8) for i, v in enumerate(vis)]
    packed = [_pack_scene(s, k) for s in base]
    F = 12
    idx = [(5 * f) % len(base) for f in range(F)]
    d_kp = torch.from_numpy(np.stack([base[i]["kp_xy"] for i in idx]).astype(np.float32)).cuda()
    d_cloud = torch.from_numpy(np.stack([base[i]["cloud"] for i in idx]).astype(np.float32)).cuda()
    d_counts = torch.from_numpy(np.stack([packed[i][0] for i in idx])).cuda()
    d_m = torch.from_numpy(np.stack([packed[i][1] for i in idx])).cuda()
    d_xyz = torch.from_numpy(np.stack([packed[i][2] for i in idx])).cuda()
    d_counts[F - 1] = 0                                                           # a frame without a single match
    torch.cuda.synchronize()
    spans = base[0]["spans"]
    want = []
    for f in range(F):
        r = capi.rng_new(7 + f)
        want.append((ctx.verify_device(d_kp[f].data_ptr(), nq, d_cloud[f].data_ptr(), 480, 640, d_counts[f].data_ptr(),
                                       d_m[f].data_ptr(), d_xyz[f].data_ptr(), k, spans, 8, 400, 0.01, r), r))
    for rep in range(2):                                                          # (the second call finds every buffer warm)
        rngs = (capi.Rng * F)(*[capi.rng_new(7 + f) for f in range(F)])
        got = ctx.verify_batch_device(F, d_kp.data_ptr(), nq, d_cloud.data_ptr(), 480, 640, d_counts.data_ptr(), d_m.data_ptr(),
                                      d_xyz.data_ptr(), k, spans, 8, 400, 0.01, rngs)
        assert sum(len(p) for p in got) >= 6 and got[F - 1] == []
        for f in range(F):
            assert rngs[f].draws == want[f][1].draws and list(rngs[f].s) == list(want[f][1].s) and len(got[f]) == len(want[f][0]), f
            for a, b in zip(got[f], want[f][0]):
                assert a["object"] == b["object"] and np.array_equal(a["inliers"], b["inliers"]) and np.array_equal(a["R"], b["R"])
    assert max(len(p["inliers"]) for f in range(F) for p in got[f]) >= 96         # some object was heavy enough to fly



def _frame_diffs(got, rngs, want):
    """why frame results of a batch differ from the frame-by-frame calls' (want[f] = (poses, generator)); [] when they do not"""
    out = []
    for f, (w, wr) in enumerate(want):
        why = []
        if rngs[f].draws != wr.draws or list(rngs[f].s) != list(wr.s):
            why.append("generator at draw %d, frame by frame %d" % (rngs[f].draws, wr.draws))
        if len(got[f]) != len(w):
            why.append("%d poses, frame by frame %d" % (len(got[f]), len(w)))
        for j, (a, b) in enumerate(zip(got[f], w)):
            if a["object"] != b["object"]:
                why.append("pose %d: object %d, frame by frame %d" % (j, a["object"], b["object"]))
            elif not np.array_equal(a["inliers"], b["inliers"]):
                why.append("pose %d (object %d): %d inliers, frame by frame %d" % (j, a["object"], len(a["inliers"]), len(b["inliers"])))
            elif not np.array_equal(a["R"], b["R"]):
                why.append("pose %d (object %d): R differs by %.3g" % (j, a["object"], np.abs(a["R"] - b["R"]).max()))
        if why:
            out.append("frame %d: %s" % (f, "; ".join(why)))
    return out


def test_batch_of_44_heavy_and_light_frames_with_several_groups_in_the_air(ctx):
    """More than 32 frames: a launch group's evaluation lists exceed one launch's argument sets and travel through device memory
    (launch_many), and several such groups are in the air on different lanes at once -- each lane has its own staging pair, which
    only grows while the lane is idle. 44 frames mixing heavy frames, frames of small objects and empty ones == frame by frame."""
    import torch
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
    want = []
    for f in range(F):
        r = capi.rng_new(3 + f % 5)
        want.append((ctx.verify_device(d_kp[f].data_ptr(), nq, d_cloud[f].data_ptr(), 480, 640, d_counts[f].data_ptr(),
                                       d_m[f].data_ptr(), d_xyz[f].data_ptr(), k, spans, 8, 600, 0.01, r), r))
    for rep in range(2):
        rngs = (capi.Rng * F)(*[capi.rng_new(3 + f % 5) for f in range(F)])
        got = ctx.verify_batch_device(F, d_kp.data_ptr(), nq, d_cloud.data_ptr(), 480, 640, d_counts.data_ptr(), d_m.data_ptr(),
                                      d_xyz.data_ptr(), k, spans, 8, 600, 0.01, rngs)
        assert sum(len(p) for p in got) >= 30
        diffs = _frame_diffs(got, rngs, want)
        assert not diffs, "repetition %d: %s" % (rep, " | ".join(diffs))


@pytest.mark.parametrize("n_obj", [40, 200])
def test_many_small_distractor_objects(ctx, n_obj):
    """One true object among many that only collect random matches (15-75 each): every distractor burns its iteration
    budget or walks 1000 failing sample attempts. Exercises the lane-per-position draw table (objects of <= 128
    matches), windows that grow past the LDS part of the chain walk, evaluation batches of thousands of hypotheses and
    the all-objects preparation tick. Must equal the oracle draw for draw."""
    sc = synth.make_verify_scene(600, n_objects=n_obj, visible=((1, 0.30),), matches_per_kp=5, seed=77)
    rng_g, rng_o = capi.rng_new(1), O.rng_new(1)
    poses = ctx.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], 8, 2500, 0.01, rng_g)
    tr = [(r.object, r.iterations, r.best_iteration, r.best_count, r.draws_after) for r in ctx.verify_trace()]
    rc, o_poses, o_tr = O.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], 8, 2500,
                                 0.01, rng_o)
    assert rc == 0 and rng_g.draws == rng_o.draws and list(rng_g.s) == list(rng_o.s) and rng_g.draws > 500000
    assert len(poses) == len(o_poses) == 1 and poses[0]["object"] == o_poses[0]["object"] == 1
    assert np.array_equal(poses[0]["inliers"], o_poses[0]["inliers"])
    assert np.abs(poses[0]["R"] - o_poses[0]["R"]).max() < POSE_TOL and np.abs(poses[0]["t"] - o_poses[0]["t"]).max() < POSE_TOL
    assert len(tr) >= n_obj // 2


def test_object_with_more_than_4096_matches(ctx):
    """6500 matches on one object: bitset rows span 102 words, i.e. more than one word per lane of a wave (the
    reference has no size limit, adjacency_ransac.h:48-133; here the limit is 16384 matches per object)."""
    sc = synth.make_verify_scene(1300, n_objects=1, per_object=3000, visible=((0, 0.25),), matches_per_kp=5, seed=90,
                                 nan_frac=0.0)
    poses, rounds = _compare_frame(ctx, sc, 8, 30)
    assert len(poses) >= 1 and len(poses[0]["inliers"]) > 100


def test_shipped_config_worst_case_25000_matches_on_one_object(ctx):
    """conf/detection.ork:26 sets n_features 5000 and the cell's k is 5 (DescriptorMatcher.cpp:211): on a 1-object DB every
    one of the 25 000 matches can land on that object (the reference has no size limit, adjacency_ransac.h:48-133). Bitset
    rows span 391 words (7 per lane of a wave); equal to the oracle (few iterations: the CPU side is what takes time)."""
    sc = synth.make_verify_scene(5000, n_objects=1, per_object=8000, visible=((0, 0.2),), matches_per_kp=5, seed=93,
                                 nan_frac=0.0)
    assert len(sc["matches"]) == 25000
    poses, rounds = _compare_frame(ctx, sc, 8, 12)
    assert len(poses) == 1 and len(poses[0]["inliers"]) > 500


def test_maximum_object_size_and_one_beyond(ctx):
    """Exactly 32768 matches on one object (the documented maximum) equals the oracle; one keypoint more is refused
    with TODHIP_ESCRATCH instead of computing something else."""
    sc = synth.make_verify_scene(8192, n_objects=1, per_object=9000, visible=((0, 0.1),), matches_per_kp=4, seed=91,
                                 nan_frac=0.0)
    assert len(sc["matches"]) == 32768
    poses, rounds = _compare_frame(ctx, sc, 8, 8)
    assert len(poses) == 1 and len(poses[0]["inliers"]) > 300
    sc = synth.make_verify_scene(8193, n_objects=1, per_object=9000, visible=((0, 0.1),), matches_per_kp=4, seed=91,
                                 nan_frac=0.0)
    rng = capi.rng_new(1)
    with pytest.raises(capi.TodError) as e:
        ctx.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], 8, 6, 0.01, rng)
    assert e.value.status == capi.ESCRATCH


def test_the_same_frame_again_and_again_is_the_same_frame(ctx):
    """A frame of distractor objects only (a few dozen matches each, several objects sharing every 256-match chunk of the cluster
    kernel's grouping pass), 15 times per generator seed through todhip_verify_device: every run's round traces and final generator
    position equal the oracle's. (cluster_frame_kernel once read an object's running count while a faster wave of the same chunk was
    already updating it: one run in three put matches in the wrong order -- same poses, other draws.)"""
    import torch
    k, nq = 3, 400
    sc = synth.make_verify_scene(nq, visible=(), seed=642, matches_per_kp=3, n_objects=8)
    c, m, x = _pack_scene(sc, k)
    d_kp = torch.from_numpy(sc["kp_xy"].astype(np.float32)).cuda()
    d_cloud = torch.from_numpy(sc["cloud"].astype(np.float32)).cuda()
    d_c, d_m, d_x = torch.from_numpy(c).cuda(), torch.from_numpy(m).cuda(), torch.from_numpy(x).cuda()
    torch.cuda.synchronize()
    key = lambda r: (r.iterations, r.best_iteration, r.best_count, r.draws_before, r.draws_after)
    for seed in (4, 5, 6):
        rng_o = O.rng_new(seed)
        rc, want, o_rounds = O.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], 8, 600, 0.01, rng_o)
        assert rc == 0
        o_rounds = [key(r) for r in o_rounds if not (r.iterations == 0 and r.draws_after == r.draws_before and r.best_count == 0)]
        assert len(o_rounds) >= 6
        for rep in range(15):
            r = capi.rng_new(seed)
            got = ctx.verify_device(d_kp.data_ptr(), nq, d_cloud.data_ptr(), 480, 640, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr(), k,
                                    sc["spans"], 8, 600, 0.01, r)
            g_rounds = [key(t) for t in ctx.verify_trace() if not (t.iterations == 0 and t.draws_after == t.draws_before)]
            assert g_rounds == o_rounds and r.draws == rng_o.draws and len(got) == len(want), (seed, rep)
