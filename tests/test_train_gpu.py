"""GPU parity of the training path (Trainer.cpp:121-187, training.cpp:57-195) against the CPU
restatement (oracle/train_oracle.c + the masked ORB restatement). PARITY UNPINNED w.r.t. the reference (no fixture,
third-party cv::ORB / erode / rescaleDepth / depthTo3dSparse)."""
import numpy as np
import pytest

import oracle_lib as O
from tod_amd import capi, synth

pytestmark = pytest.mark.gpu


def _observation(view, u16):
    img = synth.make_image(40 + view)
    rng = np.random.Generator(np.random.PCG64(900 + view))
    mask = np.zeros((480, 640), np.uint8)
    mask[60 + 10 * view:420, 120:560 - 15 * view] = 255
    mask[200:230, 300:340] = 0                                  # a hole: exercises the +-2 pixel rescue and the erosion
    z = (0.7 + 0.2 * rng.random((480, 640))).astype(np.float32)
    z[rng.random((480, 640)) < 0.05] = np.nan
    if u16:
        d16 = np.where(np.isnan(z), 0, np.rint(z * 1000)).astype(np.uint16)
        z = np.where(d16 == 0, np.nan, d16.astype(np.float32) * np.float32(0.001)).astype(np.float32)
        depth = d16
    else:
        depth = z
    a = 0.3 * view
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    T = np.array([0.1 * view, -0.05, 0.6], np.float32)
    K = np.array([[525, 0, 319.5], [0, 525, 239.5], [0, 0, 1]], np.float32)
    return img, mask, depth, z, K, R, T


@pytest.mark.parametrize("u16", [False, True])
def test_model_from_three_observations(u16):
    ctx = capi.Context(0)
    model = capi.Model(ctx, 5000)
    want_d, want_p = [], []
    for view in range(3):
        img, mask, depth, z, K, R, T = _observation(view, u16)
        n = model.add_observation(img, mask, depth, K, R, T)                    # cv::ORB defaults: 500, 8 levels, 1.2
        kp, aux, desc, _ = O.orb(img, 500, 8, 1.2, mask=mask)
        od, op, src = O.train_observation(kp, desc, mask, z, K, R, T)
        assert n == len(od) and 300 < n <= 500
        want_d.append(od); want_p.append(op)
    desc, pts = model.finish()
    assert np.array_equal(desc, np.concatenate(want_d)) and np.array_equal(pts, np.concatenate(want_p))   # mergePoints order
    model.close()
    # the trained rows are a valid DB for the matcher: a view of the object matches its own model
    off = np.array([0, len(desc)], np.uint32)
    ctx.db_load(desc, pts, off)
    row_ptr, m, xyz = ctx.match(want_d[1][:50], 1, 35)
    assert (np.diff(row_ptr.astype(np.int64)) == 1).all() and (m["distance"] == 0).all()
    ctx.close()


def test_trained_models_load_device_to_device():
    """todhip_model_device + todhip_db_load_device: three models (one of them empty) go from the trainer's device buffers into the
    matcher's DB without a host copy; spans, shard layout and matches equal those of the finish() -> todhip_db_load path."""
    ctx, ref = capi.Context(0), capi.Context(0)
    models, descs, ptss = [], [], []
    for o in range(3):
        model = capi.Model(ctx, 3000)
        for view in range(0 if o == 1 else 2):                                  # object 1 stays empty
            img, mask, depth, z, K, R, T = _observation(view + o, False)
            model.add_observation(img, mask, depth, K, R, T)
        d, p = model.finish()
        models.append(model); descs.append(d); ptss.append(p)
    spans_dev, off_dev = ctx.db_load_models(models)
    desc, pts = np.concatenate(descs), np.concatenate(ptss)
    off = np.concatenate([[0], np.cumsum([len(d) for d in descs])]).astype(np.uint32)
    spans_host = ref.db_load(desc, pts, off)
    assert np.array_equal(off_dev, off) and len(descs[1]) == 0 and len(descs[0]) > 500
    assert np.array_equal(spans_dev.view(np.uint32), np.asarray(spans_host, np.float32).view(np.uint32))
    for m in models:
        m.close()                                                               # the DB holds its own copy
    q = np.concatenate([descs[0][5:60], descs[2][100:140]])
    a, b = ctx.match(q, 3, 40), ref.match(q, 3, 40)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert (a[1]["distance"][a[0][:-1]] == 0).all()
    for rank in range(2):                                                       # the sharded form of the same load
        models2 = []
        for o in range(3):
            model = capi.Model(ctx, 3000)
            for view in range(0 if o == 1 else 2):
                img, mask, depth, z, K, R, T = _observation(view + o, False)
                model.add_observation(img, mask, depth, K, R, T)
            models2.append(model)
        ctx.db_load_models(models2, rank, 2)
        ref.db_load(desc, pts, off, rank, 2)
        assert ctx.db_info() == ref.db_info()
        for m in models2:
            m.close()
    ctx.close(); ref.close()


def test_masked_orb_only_returns_keypoints_inside_the_mask():
    ctx = capi.Context(0)
    img, mask, depth, z, K, R, T = _observation(0, False)
    model = capi.Model(ctx, 1000)
    n = model.add_observation(img, np.zeros_like(mask), depth, K, R, T)
    assert n == 0
    model.close(); ctx.close()


@pytest.mark.parametrize("u16", [False, True])
@pytest.mark.parametrize("shape", [(480, 640), (240, 320), (300, 400), (200, 320), (960, 1280), (1024, 1280), (97, 131)])
@pytest.mark.parametrize("nearest", [False, True])
def test_rescale_depth(shape, u16, nearest):
    """rescale_depth (Trainer.cpp:62-81) for depth images of other sizes than the 480 x 640 image: bit-identical to
    the CPU restatement, bilinear (what the reference executes) and nearest (what it intends)."""
    rng = np.random.Generator(np.random.PCG64(shape[0] * 7 + shape[1]))
    z = (0.4 + 3.0 * rng.random(shape)).astype(np.float32)
    z[rng.random(shape) < 0.05] = np.nan
    depth = np.where(np.isnan(z), 0, np.rint(z * 1000)).astype(np.uint16) if u16 else z
    ctx = capi.Context(0)
    want = O.train_rescale_depth(depth, 480, 640, nearest)
    if want is None:                                              # (1024, 1280): scaled height 512 > 480 rows
        with pytest.raises(capi.TodError):
            ctx.rescale_depth(depth, 480, 640, nearest)
        return
    got = ctx.rescale_depth(depth, 480, 640, nearest)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.isfinite(got).any()


def test_rescaled_depth_feeds_the_verifier_lookup():
    """uint16 depth through todhip_rescale_depth_device (equal size) and then the float lookup gives the same query
    points as the fused uint16 lookup of todhip_verify_device_depth."""
    import torch
    H, W, nq = 480, 640, 300
    rng = np.random.Generator(np.random.PCG64(5))
    d16 = rng.integers(0, 4000, (H, W)).astype(np.uint16)
    ctx = capi.Context(0)
    t_in = torch.from_numpy(d16.view(np.int16)).cuda()
    t_out = torch.empty((H, W), dtype=torch.float32, device="cuda")
    ctx.rescale_depth_device(t_in.data_ptr(), True, H, W, t_out.data_ptr(), H, W)
    ctx.synchronize()
    want = np.where(d16 == 0, np.nan, d16.astype(np.float32) * np.float32(0.001)).astype(np.float32)
    assert np.array_equal(t_out.cpu().numpy().view(np.uint32), want.view(np.uint32))
