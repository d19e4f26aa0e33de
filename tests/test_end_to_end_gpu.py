"""End to end on the GPU, every stage feeding the next: train a model from a view (todhip_model_*: masked ORB, keypoint
validation, back-projection), load it as the DB, then detect the object in a different view (ORB -> Hamming matcher
-> verifier) and compare the recovered pose with the pose the view was rendered from. No oracle here: this is a
property test of the assembled pipeline (training.ork -> detection.ork)."""
import numpy as np
import pytest

from tod_amd import capi, synth

pytestmark = pytest.mark.gpu

H, W, F = 480, 640, 525.0
K = np.array([[F, 0, W / 2.0], [0, F, H / 2.0], [0, 0, 1]], np.float32)


def _render(texture, theta, shift_px, noise_seed):
    """The fronto-parallel textured plane after a rotation about the optical axis and an in-plane translation:
    pixel p2 = Rot(theta) (p1 - c) + c + shift. Bilinear resampling of the training view."""
    c, s = np.cos(theta), np.sin(theta)
    v2, u2 = np.mgrid[0:H, 0:W].astype(np.float32)
    x = u2 - W / 2.0 - shift_px[0]; y = v2 - H / 2.0 - shift_px[1]
    u1 = c * x + s * y + W / 2.0; v1 = -s * x + c * y + H / 2.0          # inverse rotation
    u0 = np.clip(np.floor(u1).astype(np.int64), 0, W - 2); v0 = np.clip(np.floor(v1).astype(np.int64), 0, H - 2)
    fu = np.clip(u1 - u0, 0, 1); fv = np.clip(v1 - v0, 0, 1)
    t = texture.astype(np.float32)
    img = (t[v0, u0] * (1 - fu) * (1 - fv) + t[v0, u0 + 1] * fu * (1 - fv) + t[v0 + 1, u0] * (1 - fu) * fv + t[v0 + 1, u0 + 1] * fu * fv)
    inside = (u1 >= 0) & (u1 <= W - 1) & (v1 >= 0) & (v1 <= H - 1)
    img = np.where(inside, img, 128.0)
    img += np.random.Generator(np.random.PCG64(noise_seed)).normal(0, 1.5, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("theta_deg,shift", [(0.0, (0.0, 0.0)), (25.0, (30.0, -18.0)), (-70.0, (-25.0, 22.0))])
def test_train_then_detect_recovers_the_view_pose(theta_deg, shift):
    Z = 0.8
    texture = synth.make_image(321)
    ctx = capi.Context(0)
    # ---- training: one observation, camera frame == object frame (R = I, T = 0), everything but a border is object
    mask = np.zeros((H, W), np.uint8); mask[40:H - 40, 40:W - 40] = 255
    depth = np.full((H, W), Z, np.float32)
    model = capi.Model(ctx, 4000)
    n = model.add_observation(texture, mask, depth, K, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), n_features=1500,
                              n_levels=3, scale_factor=1.2)
    desc, pts = model.finish(); model.close()
    assert n == len(desc) > 800 and abs(float(pts[:, 2].mean()) - Z) < 1e-5
    # a second, unrelated object in the DB as a distractor
    rng = np.random.Generator(np.random.PCG64(9))
    d2 = rng.integers(0, 256, (2000, 32), dtype=np.uint8); p2 = (rng.random((2000, 3)) * 0.2).astype(np.float32)
    off = np.array([0, len(d2), len(d2) + len(desc)], np.uint32)
    spans = ctx.db_load(np.concatenate([d2, desc]), np.concatenate([p2, pts]), off)
    # ---- detection view
    theta = np.deg2rad(theta_deg)
    view = _render(texture, theta, shift, 5)
    kp, aux, qd = ctx.orb(view, 1000, 3, 1.2)
    v, u = np.mgrid[0:H, 0:W].astype(np.float32)
    cloud = np.stack([(u - K[0, 2]) * Z / F, (v - K[1, 2]) * Z / F, np.full((H, W), Z, np.float32)], axis=2).astype(np.float32)
    row_ptr, m, xyz = ctx.match(qd, 5, 55)                                   # conf/detection.ros.ork:60
    assert (np.diff(row_ptr.astype(np.int64)) > 0).sum() > 150
    rng_g = capi.rng_new(1)
    poses = ctx.verify(kp, cloud, row_ptr, m, xyz, spans, 8, 2500, 0.01, rng_g)
    assert len(poses) >= 1 and poses[0]["object"] == 1 and len(poses[0]["inliers"]) > 100
    c, s = np.cos(theta), np.sin(theta)
    R_true = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)
    t_true = np.array([shift[0] * Z / F, shift[1] * Z / F, 0.0], np.float32)
    t_true = t_true + (np.eye(3, dtype=np.float32) - R_true) @ np.array([0, 0, Z], np.float32)   # rotation is about the optical axis
    assert np.abs(poses[0]["R"] - R_true).max() < 0.02, (poses[0]["R"], R_true)
    assert np.abs(poses[0]["t"] - t_true).max() < 0.004, (poses[0]["t"], t_true)
    ctx.close()


@pytest.mark.parametrize("theta_deg,shift", [(0.0, (0.0, 0.0)), (25.0, (30.0, -18.0)), (-70.0, (-25.0, 22.0))])
def test_train_then_detect_from_pixels_alone_recovers_the_view_pose(theta_deg, shift):
    """The same chain without any depth on the detection side: todhip_verify_2d (the PnP branch the reference leaves as a TODO,
    GuessGenerator.cpp:147-152) on this library's own ORB keypoints and the trained model's 3D points. The recovered pose must be
    the rendering pose; depth along the optical axis is the weak direction of a PnP solution, hence the wider tolerance on t_z."""
    Z = 0.8
    texture = synth.make_image(321)
    ctx = capi.Context(0)
    mask = np.zeros((H, W), np.uint8); mask[40:H - 40, 40:W - 40] = 255
    depth = np.full((H, W), Z, np.float32)
    model = capi.Model(ctx, 4000)
    model.add_observation(texture, mask, depth, K, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), n_features=1500, n_levels=3,
                          scale_factor=1.2)
    desc, pts = model.finish(); model.close()
    rng = np.random.Generator(np.random.PCG64(9))
    d2 = rng.integers(0, 256, (2000, 32), dtype=np.uint8); p2 = (rng.random((2000, 3)) * 0.2).astype(np.float32)
    off = np.array([0, len(d2), len(d2) + len(desc)], np.uint32)
    spans = ctx.db_load(np.concatenate([d2, desc]), np.concatenate([p2, pts]), off)
    theta = np.deg2rad(theta_deg)
    view = _render(texture, theta, shift, 5)
    kp, aux, qd = ctx.orb(view, 1000, 3, 1.2)
    row_ptr, m, xyz = ctx.match(qd, 5, 55)
    poses = ctx.verify_2d(kp, K, row_ptr, m, xyz, spans, 8, 1000, 3.0, capi.rng_new(1))
    assert len(poses) == 1 and poses[0]["object"] == 1 and len(poses[0]["inliers"]) > 100
    c, s = np.cos(theta), np.sin(theta)
    R_true = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)
    t_true = np.array([shift[0] * Z / F, shift[1] * Z / F, 0.0], np.float32)
    t_true = t_true + (np.eye(3, dtype=np.float32) - R_true) @ np.array([0, 0, Z], np.float32)
    assert np.abs(poses[0]["R"] - R_true).max() < 0.02, (poses[0]["R"], R_true)
    dt = np.abs(poses[0]["t"] - t_true)
    assert dt[:2].max() < 0.004 and dt[2] < 0.02, (poses[0]["t"], t_true)
    ctx.close()
