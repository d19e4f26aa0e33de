"""Seeded randomized parity sweeps of the three stages through the C ABI against the CPU oracle: configurations
nobody hand-picked (odd sizes, tie-heavy descriptors, extreme radii, many or no visible objects, tiny images, deep
pyramids). Every case is reproducible from its seed; all comparisons are those of the per-stage parity tests."""
import numpy as np
import pytest

import oracle_lib as O
from tod_amd import capi, synth
from test_match_gpu import _assert_same as match_same
from test_orb_gpu import _same as orb_same
from test_verify_gpu import _compare_frame

pytestmark = pytest.mark.gpu

# TOD_FUZZ_OFFSET / TOD_FUZZ_SCALE widen the sweep for a one-off hunt (e.g. OFFSET=1000 SCALE=20): other seeds, more of them
import os
_OFF = int(os.environ.get("TOD_FUZZ_OFFSET", "0"))
_SCALE = int(os.environ.get("TOD_FUZZ_SCALE", "1"))


def _seeds(n):
    return range(_OFF, _OFF + n * _SCALE)


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("seed", _seeds(48))
def test_matcher_random_configurations(ctx, seed):
    rng = np.random.Generator(np.random.PCG64(31000 + seed))
    n_obj = int(rng.integers(1, 7))
    sizes = [int(rng.integers(0, 2500)) for _ in range(n_obj)]
    if sum(sizes) == 0:
        sizes[0] = 1
    nq = int(rng.integers(1, 700))
    k = int(rng.integers(1, 9))
    radius = int(rng.choice([1, 2, 20, 35, 37, 38, 47, 48, 55, 64, 79, 80, 100, 128, 256, 257, 1000]))
    n = sum(sizes)
    style = seed % 4
    if style == 0:                                             # iid bits
        desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
        q = rng.integers(0, 256, (nq, 32)).astype(np.uint8)
    elif style == 1:                                           # tie-heavy: only one byte varies, 4 values
        desc = np.zeros((n, 32), np.uint8); desc[:, 5] = rng.choice([0, 1, 3, 7], n)
        q = np.zeros((nq, 32), np.uint8); q[:, 5] = rng.choice([0, 1, 3, 7, 15], nq)
    elif style == 2:                                           # queries are noisy copies of rows
        desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
        q = desc[rng.integers(0, n, nq)].copy()
        flips = rng.integers(0, 256, (nq, 12))
        for i in range(nq):
            for b in flips[i, :rng.integers(0, 13)]:
                q[i, b >> 3] ^= np.uint8(1 << (b & 7))
    else:                                                      # biased bits (few ones): small distances everywhere
        desc = (rng.random((n, 32, 8)) < 0.08).astype(np.uint8)
        desc = np.packbits(desc, axis=2).reshape(n, 32)
        q = np.packbits((rng.random((nq, 32, 8)) < 0.08).astype(np.uint8), axis=2).reshape(nq, 32)
    pts = rng.random((n, 3)).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    match_same(ctx, np.ascontiguousarray(desc), pts, off, np.ascontiguousarray(q), k, radius)


@pytest.mark.parametrize("seed", _seeds(48))
def test_verifier_random_scenes(ctx, seed):
    rng = np.random.Generator(np.random.PCG64(32000 + seed))
    n_objects = int(rng.integers(1, 9))
    n_vis = int(rng.integers(0, min(3, n_objects) + 1))
    objs = rng.choice(n_objects, n_vis, replace=False)
    fr = rng.dirichlet(np.ones(n_vis + 1))[:n_vis] * rng.uniform(0.2, 0.9) if n_vis else []
    visible = tuple((int(o), float(f)) for o, f in zip(objs, fr))
    mpk = int(rng.integers(1, 7))
    sc = synth.make_verify_scene(int(rng.integers(40, 700)), n_objects=n_objects, per_object=int(rng.integers(50, 900)),
                                 visible=visible, matches_per_kp=mpk, seed=500 + seed, noise=float(rng.choice([0.0, 0.001, 0.003, 0.008])),
                                 nan_frac=float(rng.choice([0.0, 0.1, 0.4])), true_match_rank=int(rng.integers(0, mpk)))
    min_inliers, n_iter = int(rng.integers(5, 21)), int(rng.integers(30, 700))
    err, rseed = float(rng.choice([0.004, 0.01, 0.03])), int(rng.integers(1, 1 << 30))
    poses, rounds = _compare_frame(ctx, sc, min_inliers, n_iter, err=err, seed=rseed, max_poses=512)
    if len(poses) > 4:
        # the same frame with room for fewer poses than it yields: refused with TODHIP_ECAPACITY, never truncated silently
        with pytest.raises(capi.TodError) as e:
            ctx.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], min_inliers, n_iter,
                       err, capi.rng_new(rseed), max_poses=len(poses) - 1)
        assert e.value.status == capi.ECAPACITY


@pytest.mark.parametrize("seed", _seeds(24))
def test_orb_random_shapes(ctx, seed):
    rng = np.random.Generator(np.random.PCG64(33000 + seed))
    H, W = int(rng.integers(70, 620)), int(rng.integers(70, 820))
    img = synth.make_image(100 + seed, H=H, W=W, n_rect=int(rng.integers(5, 1500)))
    if seed % 5 == 4:
        img = rng.integers(0, 256, (H, W)).astype(np.uint8)   # white noise: corners everywhere
    orb_same(ctx, img, int(rng.integers(1, 1600)), int(rng.integers(1, 9)), float(rng.choice([1.1, 1.2, 1.3, 1.5, 2.0])))


@pytest.mark.parametrize("seed", _seeds(8))
def test_verifier_random_batches_against_the_oracle(ctx, seed):
    """todhip_verify_batch_device on 2..24 random frames (more than one launch group when > 16) that share a model
    set but differ in what is visible, in noise, missing depth and generator seed: every frame equals the oracle's
    result for that frame alone -- poses, inlier lists and the final generator state."""
    import torch
    from test_verify_gpu import _pack_scene, POSE_TOL
    rng = np.random.Generator(np.random.PCG64(34000 + seed))
    F, nq, k = int(rng.integers(2, 25)), int(rng.integers(60, 400)), int(rng.integers(1, 6))
    n_objects, per_object = int(rng.integers(2, 8)), int(rng.integers(100, 600))
    mpk = int(rng.integers(1, k + 1))
    scenes = []
    for f in range(F):
        n_vis = int(rng.integers(0, 3))
        objs = rng.choice(n_objects, n_vis, replace=False)
        visible = tuple((int(o), float(rng.uniform(0.1, 0.4))) for o in objs)
        scenes.append(synth.make_verify_scene(nq, n_objects=n_objects, per_object=per_object, visible=visible, matches_per_kp=mpk,
                                              seed=700 + seed, noise=float(rng.choice([0.0, 0.002, 0.006])),
                                              nan_frac=float(rng.choice([0.0, 0.2])), true_match_rank=int(rng.integers(0, mpk))))
    assert all(np.array_equal(s["spans"], scenes[0]["spans"]) for s in scenes)
    packed = [_pack_scene(s, k) for s in scenes]
    d_kp = torch.from_numpy(np.stack([s["kp_xy"] for s in scenes]).astype(np.float32)).cuda()
    d_cloud = torch.from_numpy(np.stack([s["cloud"] for s in scenes]).astype(np.float32)).cuda()
    d_counts = torch.from_numpy(np.stack([p[0] for p in packed])).cuda()
    d_m = torch.from_numpy(np.stack([p[1] for p in packed])).cuda()
    d_xyz = torch.from_numpy(np.stack([p[2] for p in packed])).cuda()
    torch.cuda.synchronize()
    min_inliers, n_iter, err = int(rng.integers(6, 15)), int(rng.integers(50, 500)), 0.01
    seeds = [int(x) for x in rng.integers(1, 1 << 30, F)]
    seeds[-1] = seeds[0]                                           # two frames sharing one rand() stream start
    rngs = (capi.Rng * F)(*[capi.rng_new(s) for s in seeds])
    got = ctx.verify_batch_device(F, d_kp.data_ptr(), nq, d_cloud.data_ptr(), 480, 640, d_counts.data_ptr(), d_m.data_ptr(),
                                  d_xyz.data_ptr(), k, scenes[0]["spans"], min_inliers, n_iter, err, rngs, max_poses=256)
    for f, sc in enumerate(scenes):
        rng_o = O.rng_new(seeds[f])
        rc, want, _ = O.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], min_inliers,
                               n_iter, err, rng_o, max_poses=256)
        assert rc == 0 and len(got[f]) == len(want), f
        assert rngs[f].draws == rng_o.draws and list(rngs[f].s) == list(rng_o.s)
        for a, b in zip(got[f], want):
            assert a["object"] == b["object"] and np.array_equal(a["inliers"], b["inliers"])
            assert np.abs(a["R"] - b["R"]).max() < POSE_TOL and np.abs(a["t"] - b["t"]).max() < POSE_TOL


@pytest.mark.parametrize("seed", _seeds(12))
def test_sharded_matcher_random_configurations(ctx, seed):
    """Object-aligned shards (1..9 of them, more shards than objects included) + merge == the unsharded oracle, for
    random object sizes (empty objects too), k, radius and tie-heavy descriptors."""
    from test_match_gpu import _shard_merge
    rng = np.random.Generator(np.random.PCG64(35000 + seed))
    sizes = [int(rng.integers(0, 1500)) for _ in range(int(rng.integers(1, 9)))]
    if sum(sizes) == 0:
        sizes[-1] = 7
    n, nq, k = sum(sizes), int(rng.integers(1, 400)), int(rng.integers(1, 9))
    radius = int(rng.choice([5, 35, 47, 90, 256, 300]))
    if seed % 3 == 0:
        desc = np.zeros((n, 32), np.uint8); desc[:, 9] = rng.choice([0, 1, 3], n)
        q = np.zeros((nq, 32), np.uint8); q[:, 9] = rng.choice([0, 1, 3, 7], nq)
    else:
        desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
        q = desc[rng.integers(0, n, nq)] ^ (rng.random((nq, 32)) < 0.1).astype(np.uint8)
    pts = rng.random((n, 3)).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    q = np.ascontiguousarray(q)
    counts, m, xyz, infos = _shard_merge(desc, pts, off, q, k, radius, int(rng.integers(1, 10)))
    rc, o_row_ptr, o_m, o_xyz = O.match(desc, off, pts, q, k, radius)
    assert rc == 0 and np.array_equal(np.diff(o_row_ptr.astype(np.int64)), counts)
    for f in ("queryIdx", "trainIdx", "imgIdx", "distance"):
        assert np.array_equal(m[f], o_m[f]), f
    assert np.array_equal(xyz, o_xyz)
    assert sum(i["shard_rows"] for i in infos) == n


@pytest.mark.parametrize("seed", _seeds(6))
def test_orb_random_batches(ctx, seed):
    """todhip_orb_batch_device on 1..20 frames of a random common shape == the CPU restatement frame by frame."""
    import torch
    rng = np.random.Generator(np.random.PCG64(36000 + seed))
    F, H, W = int(rng.integers(1, 21)), int(rng.integers(80, 400)), int(rng.integers(80, 500))
    nf, nl, sf = int(rng.integers(20, 900)), int(rng.integers(1, 7)), float(rng.choice([1.15, 1.2, 1.4]))
    imgs = [synth.make_image(300 + 20 * seed + f, H=H, W=W, n_rect=int(rng.integers(0, 800))) for f in range(F)]
    d = torch.from_numpy(np.stack(imgs)).cuda()
    kp = torch.zeros((F, nf, 2), device="cuda"); aux = torch.zeros((F, nf, 4), device="cuda")
    desc = torch.zeros((F, nf, 32), dtype=torch.uint8, device="cuda")
    n = ctx.orb_batch_device(d.data_ptr(), F, H * W, H, W, W, nf, nl, sf, kp.data_ptr(), aux.data_ptr(), desc.data_ptr(), nf)
    for f in range(F):
        o_kp, o_aux, o_desc, _ = O.orb(imgs[f], nf, nl, sf)
        assert n[f] == len(o_kp)
        assert np.array_equal(kp[f, :n[f]].cpu().numpy(), o_kp) and np.array_equal(desc[f, :n[f]].cpu().numpy(), o_desc)
        assert np.array_equal(aux[f, :n[f]].cpu().numpy()[:, [0, 2, 3]], o_aux[:, [0, 2, 3]])


@pytest.mark.parametrize("seed", _seeds(6))
def test_l2_random_configurations(ctx, seed):
    """Float-descriptor matcher on random sizes, k, radii and value ranges (integer-valued, tiny, huge norms)."""
    from test_l2_gpu import _assert_same as l2_same
    rng = np.random.Generator(np.random.PCG64(37000 + seed))
    sizes = [int(rng.integers(0, 3000)) for _ in range(int(rng.integers(1, 6)))]
    if sum(sizes) == 0:
        sizes[0] = 3
    n, nq, k = sum(sizes), int(rng.integers(1, 300)), int(rng.integers(1, 9))
    scale = float(rng.choice([1.0, 255.0, 1e-3, 3e3]))
    desc = (rng.random((n, 128)) * scale).astype(np.float32)
    if seed % 2:
        desc = np.rint(desc).astype(np.float32)
    q = (desc[rng.integers(0, n, nq)] + rng.normal(0, 0.05 * scale, (nq, 128))).astype(np.float32)
    pts = rng.random((n, 3)).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    ctx.db_load(desc, pts, off)
    radius = float(rng.choice([0.3, 0.6, 1.5, 1e9])) * scale
    l2_same(ctx, desc, pts, off, q, k, radius)


@pytest.mark.parametrize("mode", ["duplicate_model_points", "collinear_model", "huge_coordinates", "tiny_coordinates",
                                  "nan_inf_model_points", "coplanar_cloud", "zero_span", "all_same_keypoint_pixel"])
@pytest.mark.parametrize("seed", _seeds(3))
def test_verifier_degenerate_geometry(ctx, mode, seed):
    """Inputs the arithmetic was not designed for -- duplicate or collinear model points, coordinates at 1e3 / 1e-3 scale,
    NaN / Inf among the model points, a planar cloud, zero spans, every keypoint on one pixel: whatever the reference
    arithmetic makes of them, the GPU makes the same (comparisons with NaN, distances of 0, singular covariance)."""
    rng = np.random.Generator(np.random.PCG64(38000 + seed))
    sc = synth.make_verify_scene(int(rng.integers(120, 400)), n_objects=3, per_object=300, visible=((1, 0.4),), matches_per_kp=3,
                                 seed=800 + seed, nan_frac=0.05)
    mx, cloud, kp, spans = sc["matches_xyz"].copy(), sc["cloud"].copy(), sc["kp_xy"].copy(), sc["spans"].copy()
    if mode == "duplicate_model_points":
        mx[rng.random(len(mx)) < 0.4] = mx[0]
    elif mode == "collinear_model":
        mx[:, 1] = mx[:, 0] * np.float32(0.5); mx[:, 2] = mx[:, 0] * np.float32(-0.25)
    elif mode == "huge_coordinates":
        mx *= np.float32(1e3); cloud *= np.float32(1e3); spans = spans * np.float32(1e3)
    elif mode == "tiny_coordinates":
        mx *= np.float32(1e-3); cloud *= np.float32(1e-3); spans = spans * np.float32(1e-3)
    elif mode == "nan_inf_model_points":
        bad = rng.random(len(mx)) < 0.15
        mx[bad, rng.integers(0, 3, bad.sum())] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), bad.sum())
    elif mode == "coplanar_cloud":
        cloud[:, :, 2] = np.where(np.isnan(cloud[:, :, 2]), np.nan, np.float32(0.9))
    elif mode == "zero_span":
        spans = np.zeros_like(spans)
    elif mode == "all_same_keypoint_pixel":
        p = cloud[int(kp[0, 1]), int(kp[0, 0])].copy()
        kp[:] = kp[0]
        cloud[int(kp[0, 1]), int(kp[0, 0])] = np.where(np.isnan(p), np.float32(0.5), p)
    sc2 = dict(sc, matches_xyz=mx, cloud=cloud, kp_xy=kp, spans=spans)
    _compare_frame(ctx, sc2, 6, 150, err=[0.01, 10.0, 1e-5][seed % 3], seed=1 + seed, max_poses=512)


@pytest.mark.parametrize("pattern", ["checkerboard8", "checkerboard3", "stripes", "dots", "two_level_noise", "saturated_blocks",
                                     "periodic_texture"])
def test_orb_on_images_full_of_ties(ctx, pattern):
    """Periodic and piecewise-constant images: thousands of corners with EQUAL FAST scores and EQUAL Harris responses,
    so every selection (threshold by histogram, ties in (y, x) order, top-n by response) is decided by the tie rules."""
    H, W = 240, 320
    y, x = np.mgrid[0:H, 0:W]
    rng = np.random.Generator(np.random.PCG64(39000))
    if pattern == "checkerboard8":
        img = (((y // 8) + (x // 8)) % 2 * 200 + 20)
    elif pattern == "checkerboard3":
        img = (((y // 3) + (x // 3)) % 2 * 255)
    elif pattern == "stripes":
        img = ((x // 5) % 2 * 180 + (y // 40) % 2 * 40)
    elif pattern == "dots":
        img = np.full((H, W), 30); img[4::9, 4::9] = 250
    elif pattern == "two_level_noise":
        img = rng.integers(0, 2, (H, W)) * 255
    elif pattern == "saturated_blocks":
        img = np.where(((y // 16) % 2 == 0) & ((x // 16) % 3 == 0), 255, 0)
    else:
        img = (128 + 100 * np.sin(x * 0.7) * np.sin(y * 0.9)).astype(np.int64)
    img = np.ascontiguousarray(img, np.uint8)
    for nf, nl in ((300, 3), (1500, 1), (50, 5)):
        orb_same(ctx, img, nf, nl, 1.2)


@pytest.mark.parametrize("seed", _seeds(4))
def test_l2_random_large_databases(ctx, seed):
    """Float-descriptor matcher on DBs above 64k rows, where the seed of an evenly spaced sample is used as the candidate
    threshold and the A_k pass is skipped: clustered rows (the sample sees few members of a query's cluster), duplicates,
    integer values, several k."""
    from test_l2_gpu import _assert_same as l2_same
    rng = np.random.Generator(np.random.PCG64(40000 + seed))
    n_obj = int(rng.integers(3, 9))
    sizes = [int(rng.integers(8000, 30000)) for _ in range(n_obj)]
    while sum(sizes) < 66000:
        sizes.append(int(rng.integers(8000, 30000)))
    n, nq, k = sum(sizes), int(rng.integers(8, 50)), int(rng.integers(1, 9))
    centres = (rng.random((len(sizes), 128)) * 200).astype(np.float32)
    spread = float(rng.choice([2.0, 20.0, 200.0]))
    desc = np.concatenate([c[None, :] + rng.normal(0, spread, (m, 128)).astype(np.float32) for c, m in zip(centres, sizes)])
    if seed % 2:
        desc = np.rint(desc).astype(np.float32)
    desc[rng.integers(0, n, 50)] = desc[0]                                    # 50 duplicates of one row
    q = (desc[rng.integers(0, n, nq)] + rng.normal(0, 0.3 * spread, (nq, 128))).astype(np.float32)
    q[0] = desc[0]
    pts = rng.random((n, 3)).astype(np.float32)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    ctx.db_load(np.ascontiguousarray(desc, np.float32), pts, off)
    l2_same(ctx, desc, pts, off, q, k, float(rng.choice([0.5, 3.0, 1e9])) * spread * 11.3)


@pytest.mark.parametrize("seed", _seeds(16))
def test_lsh_mode_random_configurations(seed):
    """todhip_set_lsh against oracle/lsh_oracle.c on random table counts, key sizes, probe depths, DB shapes (ragged, tie-heavy,
    tiny) and k; a fresh context per case (the index belongs to the context)."""
    import torch
    rng = np.random.Generator(np.random.PCG64(36000 + seed))
    n_obj = int(rng.integers(1, 6))
    sizes = [int(rng.integers(0, 3000)) for _ in range(n_obj)]
    if sum(sizes) == 0:
        sizes[-1] = 3
    n, nq, k = sum(sizes), int(rng.integers(1, 300)), int(rng.integers(1, 9))
    tables, ks = int(rng.integers(1, 13)), int(rng.integers(1, 21))
    level = int(rng.integers(0, min(3, ks) + 1))
    if seed % 3 == 1:                                          # tie-heavy: few distinct descriptors, huge buckets
        desc = np.zeros((n, 32), np.uint8); desc[:, 9] = rng.choice([0, 1, 3, 7, 255], n); desc[:, 20] = rng.choice([0, 16], n)
        q = np.zeros((nq, 32), np.uint8); q[:, 9] = rng.choice([0, 1, 3, 7, 15], nq)
    else:
        desc = rng.integers(0, 256, (n, 32)).astype(np.uint8)
        q = desc[rng.integers(0, n, nq)].copy()
        for i in range(nq):
            for b in rng.integers(0, 256, int(rng.integers(0, 20))):
                q[i, b >> 3] ^= np.uint8(1 << (b & 7))
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    pts = rng.random((n, 3)).astype(np.float32)
    c = capi.Context(0)
    if seed % 2:
        c.set_lsh(tables, ks, level); c.db_load(desc, pts, off)
    else:
        c.db_load(desc, pts, off); c.set_lsh(tables, ks, level)
    want, _ = O.lsh_knn_keys(desc, q, k, tables, ks, level)
    d_q = torch.from_numpy(q).cuda()
    keys = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    c.match_shard_device(d_q.data_ptr(), nq, k, 256, keys.data_ptr())
    c.synchronize()
    got = keys.cpu().numpy().view(np.uint64)
    c.close()
    assert np.array_equal(got, want), (tables, ks, level, k)


@pytest.mark.parametrize("seed", _seeds(16))
def test_verify_2d_random_scenes(ctx, seed):
    """todhip_verify_2d against oracle/pnp_oracle.c on random scenes: 0-3 visible objects, 1-6 matches per keypoint, thresholds from
    sub-pixel to huge, few and many hypotheses, other intrinsics; objects, consensus sets and poses must be equal bit for bit."""
    rng = np.random.Generator(np.random.PCG64(37000 + seed))
    n_obj = int(rng.integers(2, 12))
    n_vis = int(rng.integers(0, min(3, n_obj) + 1))
    vis_objs = rng.choice(n_obj, n_vis, replace=False)
    fr = rng.dirichlet(np.ones(n_vis + 1))[:n_vis] * 0.8 if n_vis else []
    visible = tuple((int(o), float(max(f, 0.05))) for o, f in zip(vis_objs, fr))
    f = float(rng.choice([400.0, 525.0, 900.0]))
    sc = synth.make_verify_scene(int(rng.integers(30, 900)), n_objects=n_obj, per_object=int(rng.integers(20, 500)), visible=visible,
                                 matches_per_kp=int(rng.integers(1, 7)), seed=500 + seed, f=f, noise=float(rng.choice([0.0, 0.001, 0.004])))
    K = np.array([[f, 0, 320.0], [0, f * float(rng.choice([1.0, 0.97])), 240.0], [0, 0, 1]], np.float32)
    min_inl, n_iter = int(rng.integers(3, 30)), int(rng.choice([1, 7, 100, 600]))
    err_px = float(rng.choice([0.3, 2.0, 5.0, 300.0]))
    s = int(rng.integers(1, 1000))
    rng_o, rng_g = O.rng_new(s), capi.rng_new(s)
    rc, want, _ = O.verify_2d(sc["kp_xy"], K, sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], min_inl, n_iter, err_px, rng_o)
    assert rc == 0
    got = ctx.verify_2d(sc["kp_xy"], K, sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], min_inl, n_iter, err_px, rng_g)
    assert rng_g.draws == rng_o.draws
    assert [p["object"] for p in got] == [p["object"] for p in want]
    for g, w in zip(got, want):
        assert np.array_equal(g["inliers"], w["inliers"]) and np.array_equal(g["R"], w["R"]) and np.array_equal(g["t"], w["t"])
