#!/usr/bin/env python3
"""Train an object model from one RGB-D view and detect it in another view -- the whole path on one MI355X:

    todhip_model_*  (masked ORB, keypoint validation, back-projection)   <- src/training/Trainer.cpp
    todhip_db_load                                                        <- DescriptorMatcher::parameter_callback
    todhip_orb -> todhip_match -> todhip_verify                           <- detector.py: features -> matcher -> guess generator

Synthetic data (a textured plane rendered at a known pose), so that the recovered pose can be checked.
Run on a machine with an MI355X:  python examples/train_and_detect.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tod_amd import capi, synth   # noqa: E402

H, W, F, Z = 480, 640, 525.0, 0.8
K = np.array([[F, 0, W / 2.0], [0, F, H / 2.0], [0, 0, 1]], np.float32)


def render(texture, theta, shift_px):
    """The plane rotated by theta about the optical axis and shifted in the image (bilinear resampling)."""
    c, s = np.cos(theta), np.sin(theta)
    v2, u2 = np.mgrid[0:H, 0:W].astype(np.float32)
    x, y = u2 - W / 2.0 - shift_px[0], v2 - H / 2.0 - shift_px[1]
    u1, v1 = c * x + s * y + W / 2.0, -s * x + c * y + H / 2.0
    u0 = np.clip(np.floor(u1).astype(np.int64), 0, W - 2); v0 = np.clip(np.floor(v1).astype(np.int64), 0, H - 2)
    fu, fv = np.clip(u1 - u0, 0, 1), np.clip(v1 - v0, 0, 1)
    t = texture.astype(np.float32)
    img = t[v0, u0] * (1 - fu) * (1 - fv) + t[v0, u0 + 1] * fu * (1 - fv) + t[v0 + 1, u0] * (1 - fu) * fv + t[v0 + 1, u0 + 1] * fu * fv
    inside = (u1 >= 0) & (u1 <= W - 1) & (v1 >= 0) & (v1 <= H - 1)
    return np.clip(np.rint(np.where(inside, img, 128.0)), 0, 255).astype(np.uint8)


def main():
    ctx = capi.Context(0)
    texture = synth.make_image(321)
    depth = np.full((H, W), Z, np.float32)
    mask = np.zeros((H, W), np.uint8); mask[40:H - 40, 40:W - 40] = 255

    # --- training: one observation; the object frame is the training camera frame (R = I, T = 0)
    model = capi.Model(ctx, 4000)
    model.add_observation(texture, mask, depth, K, np.eye(3, dtype=np.float32), np.zeros(3, np.float32), n_features=1500,
                          n_levels=3, scale_factor=1.2)
    desc, pts = model.finish(); model.close()
    print("model: %d descriptors with 3D points" % len(desc))
    spans = ctx.db_load(desc, pts, np.array([0, len(desc)], np.uint32))

    # --- detection in a rotated, shifted view
    theta, shift = np.deg2rad(33.0), (24.0, -15.0)
    view = render(texture, theta, shift)
    kp, aux, q = ctx.orb(view, 1000, 3, 1.2)
    v, u = np.mgrid[0:H, 0:W].astype(np.float32)
    cloud = np.stack([(u - K[0, 2]) * Z / F, (v - K[1, 2]) * Z / F, np.full((H, W), Z, np.float32)], axis=2).astype(np.float32)
    row_ptr, matches, xyz = ctx.match(q, 5, 55)
    poses = ctx.verify(kp, cloud, row_ptr, matches, xyz, spans, 8, 2500, 0.01, capi.rng_new(1))
    c, s = np.cos(theta), np.sin(theta)
    R_true = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)
    t_true = np.array([shift[0] * Z / F, shift[1] * Z / F, 0.0], np.float32) + (np.eye(3, dtype=np.float32) - R_true) @ np.array([0, 0, Z], np.float32)
    for p in poses:
        print("object %d: %d inlier keypoints\nR =\n%s\nt = %s" % (p["object"], len(p["inliers"]), np.round(p["R"], 4), np.round(p["t"], 4)))
    if poses:
        print("max |R - R_true| = %.4f, max |t - t_true| = %.4f m" % (np.abs(poses[0]["R"] - R_true).max(), np.abs(poses[0]["t"] - t_true).max()))
    # --- the same detection from pixels alone (no depth, no cloud): the PnP branch, todhip_verify_2d
    poses2 = ctx.verify_2d(kp, K, row_ptr, matches, xyz, spans, 8, 1000, 3.0, capi.rng_new(1))
    for p in poses2:
        print("2D only -- object %d: %d inlier keypoints, max |R - R_true| = %.4f, max |t - t_true| = %.4f m" %
              (p["object"], len(p["inliers"]), np.abs(p["R"] - R_true).max(), np.abs(p["t"] - t_true).max()))
    ctx.close()


if __name__ == "__main__":
    main()
