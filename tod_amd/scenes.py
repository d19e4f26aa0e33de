"""Rendered-view workload for the data-chained pipeline (train -> ORB -> match -> verify feeding each other).

Objects are textured fronto-parallel planes at depth Z (texture = the SURVEY 8(d) stage-A image generator). A view is
the plane after a rotation about the optical axis and an in-plane shift: pixel p2 = Rot(theta) (p1 - c) + c + shift,
so the camera-frame pose of the object is R = Rz(theta), t = (shift Z / f, 0) + (I - R) (0, 0, Z). Views are rendered
on the GPU with torch (bilinear resampling + sensor noise); models are trained from a few views through todhip_model_*
(masked ORB -> keypoint validation -> back-projection -> camera-to-world, Trainer.cpp:121-187), so the DB holds the
descriptors of this library's own ORB -- real rBRIEF statistics, not independent bits. Test / bench infrastructure: it
needs torch and a GPU, and nothing in the C ABI depends on it."""
import numpy as np

from . import synth

H, W, F, Z = 480, 640, 525.0, 0.8
K = np.array([[F, 0, W / 2.0], [0, F, H / 2.0], [0, 0, 1]], np.float32)
TEXTURE_SEED = 5000
TRAIN_VIEWS = ((0.0, (0.0, 0.0)), (14.0, (12.0, -9.0)), (-21.0, (-15.0, 11.0)), (33.0, (8.0, 14.0)))   # theta_deg, shift px


def view_pose(theta_deg, shift_px):
    """Camera-frame pose (R, t) of the plane in a view: p_cam = R p_obj + t."""
    th = np.deg2rad(theta_deg)
    c, s = np.cos(th), np.sin(th)
    R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)
    t = np.array([shift_px[0] * Z / F, shift_px[1] * Z / F, 0.0], np.float32)
    t = t + (np.eye(3, dtype=np.float32) - R) @ np.array([0, 0, Z], np.float32)
    return R, t.astype(np.float32)


def make_textures(n_objects):
    return np.stack([synth.make_image(TEXTURE_SEED + o) for o in range(n_objects)])


def render_views(textures_gpu, obj_ids, thetas_deg, shifts_px, noise_seed, noise_sigma=1.5):
    """textures_gpu: u8 [n_obj, H, W] on the GPU. Returns (images u8 [n, H, W], inside bool [n, H, W]) on the GPU."""
    import torch
    dev = textures_gpu.device
    n = len(obj_ids)
    th = torch.deg2rad(torch.tensor(thetas_deg, dtype=torch.float32, device=dev)).view(n, 1, 1)
    sh = torch.tensor(shifts_px, dtype=torch.float32, device=dev).view(n, 2)
    v2, u2 = torch.meshgrid(torch.arange(H, dtype=torch.float32, device=dev), torch.arange(W, dtype=torch.float32, device=dev),
                            indexing="ij")
    x = u2.unsqueeze(0) - W / 2.0 - sh[:, 0].view(n, 1, 1)
    y = v2.unsqueeze(0) - H / 2.0 - sh[:, 1].view(n, 1, 1)
    c, s = torch.cos(th), torch.sin(th)
    u1 = c * x + s * y + W / 2.0                                     # inverse rotation: where the pixel comes from
    v1 = -s * x + c * y + H / 2.0
    inside = (u1 >= 0) & (u1 <= W - 1) & (v1 >= 0) & (v1 <= H - 1)
    grid = torch.stack([2.0 * u1 / (W - 1) - 1.0, 2.0 * v1 / (H - 1) - 1.0], dim=-1)
    src = textures_gpu[torch.as_tensor(obj_ids, device=dev, dtype=torch.long)].to(torch.float32).unsqueeze(1) - 128.0
    img = torch.nn.functional.grid_sample(src, grid, mode="bilinear", padding_mode="zeros", align_corners=True).squeeze(1) + 128.0
    img = torch.where(inside, img, torch.full_like(img, 128.0))
    g = torch.Generator(device=dev)
    g.manual_seed(int(noise_seed))
    img = img + noise_sigma * torch.randn(img.shape, generator=g, device=dev)
    return torch.clamp(torch.round(img), 0, 255).to(torch.uint8), inside


def train_db(ctx, textures, n_features=1300, n_levels=3, views=TRAIN_VIEWS, rows_per_object=None):
    """One model per texture from `views` (todhip_model_*). Returns (desc u8[N, 32], pts f32[N, 3], obj_off u32[n + 1]).
    rows_per_object: keep at most that many rows of each model (the first ones: whole early views)."""
    import torch
    from . import capi
    n_obj = len(textures)
    tex = torch.from_numpy(textures).cuda()
    depth = np.full((H, W), Z, np.float32)
    descs, ptss, off = [], [], [0]
    border = torch.zeros((H, W), dtype=torch.bool, device="cuda")
    border[40:H - 40, 40:W - 40] = True
    for o in range(n_obj):
        imgs, inside = render_views(tex, [o] * len(views), [v[0] for v in views], [v[1] for v in views], 700000 + o)
        # the object region of a view: where the plane is seen, minus a border (validateKeyPoints erodes the mask further)
        masks = (inside & border).to(torch.uint8) * 255
        imgs, masks = imgs.cpu().numpy(), masks.cpu().numpy()
        model = capi.Model(ctx, len(views) * n_features + 16)
        for vi, (theta, shift) in enumerate(views):
            R, t = view_pose(theta, shift)
            model.add_observation(imgs[vi], masks[vi], depth, K, R, t, n_features=n_features, n_levels=n_levels, scale_factor=1.2)
        d, p = model.finish()
        model.close()
        if rows_per_object is not None:
            d, p = d[:rows_per_object], p[:rows_per_object]
        descs.append(d); ptss.append(p); off.append(off[-1] + len(d))
    return np.concatenate(descs), np.concatenate(ptss), np.asarray(off, np.uint32)


def make_detection_batches(textures, n_batches, frames_per_batch, seed=0, visible_fraction=0.30):
    """Detection frames resident on the GPU: per batch u8 images [B, H, W], f32 depth [B, H, W] (the plane's constant Z),
    plus the visible object and its true pose per frame. As in SURVEY 8(d)'s frames, about `visible_fraction` of the
    picture shows the object (a window of its plane, at a random place); the rest is clutter -- an unrelated texture that
    is in no model -- so ~30 % of the keypoints lie on the object and the others produce the stray matches a real scene has."""
    import torch
    n_obj = len(textures)
    tex = torch.from_numpy(textures).cuda()
    rng = np.random.Generator(np.random.PCG64(synth.FRAME_SEED + 90000 + seed))
    wh, ww = int(round(H * visible_fraction ** 0.5)), int(round(W * visible_fraction ** 0.5))
    out = []
    for b in range(n_batches):
        ids = [(17 * (b * frames_per_batch + f) + 3) % n_obj for f in range(frames_per_batch)]
        thetas = rng.uniform(-40.0, 40.0, frames_per_batch)
        shifts = rng.uniform(-30.0, 30.0, (frames_per_batch, 2))
        imgs, _ = render_views(tex, ids, thetas.tolist(), shifts.tolist(), 800000 + 1000 * seed + b)
        if visible_fraction < 1.0:
            clutter = torch.from_numpy(np.stack([synth.make_image(TEXTURE_SEED + 100000 + 1000 * seed + b * frames_per_batch + f)
                                                 for f in range(frames_per_batch)])).cuda()
            for f in range(frames_per_batch):
                y0 = int(rng.integers(40, H - 40 - wh)); x0 = int(rng.integers(40, W - 40 - ww))
                clutter[f, y0:y0 + wh, x0:x0 + ww] = imgs[f, y0:y0 + wh, x0:x0 + ww]
            imgs = clutter
        depth = torch.full((frames_per_batch, H, W), Z, dtype=torch.float32, device="cuda")
        out.append(dict(images=imgs.contiguous(), depth=depth, objects=ids,
                        poses=[view_pose(thetas[f], shifts[f]) for f in range(frames_per_batch)]))
    return out
