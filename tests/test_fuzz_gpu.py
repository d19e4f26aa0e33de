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


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("seed", range(48))
def test_matcher_random_configurations(ctx, seed):
    rng = np.random.Generator(np.random.PCG64(31000 + seed))
    n_obj = int(rng.integers(1, 7))
    sizes = [int(rng.integers(0, 2500)) for _ in range(n_obj)]
    if sum(sizes) == 0:
        sizes[0] = 1
    nq = int(rng.integers(1, 700))
    k = int(rng.integers(1, 9))
    radius = int(rng.choice([1, 2, 20, 35, 38, 39, 46, 47, 64, 100, 128, 256, 257, 1000]))
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


@pytest.mark.parametrize("seed", range(48))
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


@pytest.mark.parametrize("seed", range(24))
def test_orb_random_shapes(ctx, seed):
    rng = np.random.Generator(np.random.PCG64(33000 + seed))
    H, W = int(rng.integers(70, 620)), int(rng.integers(70, 820))
    img = synth.make_image(100 + seed, H=H, W=W, n_rect=int(rng.integers(5, 1500)))
    if seed % 5 == 4:
        img = rng.integers(0, 256, (H, W)).astype(np.uint8)   # white noise: corners everywhere
    orb_same(ctx, img, int(rng.integers(1, 1600)), int(rng.integers(1, 9)), float(rng.choice([1.1, 1.2, 1.3, 1.5, 2.0])))
