"""Seeded synthetic workloads for the detection hot path.

There is no dataset on the machines this runs on, so every test and bench input is generated
here: an object database (descriptors + model points), camera frames (keypoints, descriptors,
organised point cloud) with one visible object at a known pose, and grey images for stage A.
All generators are numpy.random.Generator(PCG64(seed)) so results are reproducible anywhere.
"""
import numpy as np

DB_SEED = 1001
FRAME_SEED = 2000
IMAGE_SEED = 3000

POSE_RZ = 0.7
POSE_T = np.array([0.05, -0.02, 0.8], np.float32)


def pose_R():
    c, s = np.cos(POSE_RZ), np.sin(POSE_RZ)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)


def make_db(n_objects, per_object=5000, desc_bytes=32, seed=DB_SEED):
    """Returns (desc u8[N,desc_bytes], pts f32[N,3], obj_off u32[n_objects+1])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = n_objects * per_object
    desc = rng.integers(0, 256, size=(n, desc_bytes), dtype=np.uint8)
    pts = (rng.random((n, 3), dtype=np.float32) - 0.5) * np.array([0.20, 0.15, 0.10], np.float32)
    obj_off = (np.arange(n_objects + 1, dtype=np.uint64) * per_object).astype(np.uint32)
    return desc, pts.astype(np.float32), obj_off


def make_db_ragged(sizes, desc_bytes=32, seed=DB_SEED):
    rng = np.random.Generator(np.random.PCG64(seed))
    n = int(np.sum(sizes))
    desc = rng.integers(0, 256, size=(n, desc_bytes), dtype=np.uint8)
    pts = (rng.random((n, 3), dtype=np.float32) - 0.5) * np.array([0.20, 0.15, 0.10], np.float32)
    obj_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
    return desc, pts.astype(np.float32), obj_off


def make_frame(desc, pts, obj_off, n_kp, frame=0, visible_object=0, H=480, W=640, f=525.0,
               on_object=0.30, flip_p=0.08, noise=0.002, nan_frac=0.10):
    """One camera frame. Returns dict(kp_xy f32[Q,2], q_desc u8[Q,B], cloud f32[H,W,3], truth_rows i64[Q])."""
    rng = np.random.Generator(np.random.PCG64(FRAME_SEED + frame))
    B = desc.shape[1]
    n_on = int(round(n_kp * on_object))
    lo, hi = int(obj_off[visible_object]), int(obj_off[visible_object + 1])
    rows = rng.choice(np.arange(lo, hi), size=min(n_on, hi - lo), replace=False)
    n_on = len(rows)
    q_desc = rng.integers(0, 256, size=(n_kp, B), dtype=np.uint8)
    flips = np.packbits(rng.random((n_on, B * 8)) < flip_p, axis=1, bitorder="little")
    q_desc[:n_on] = desc[rows] ^ flips
    R = pose_R()
    xyz = np.empty((n_kp, 3), np.float32)
    xyz[:n_on] = (pts[rows] @ R.T + POSE_T + rng.normal(0, noise, (n_on, 3))).astype(np.float32)
    n_cl = n_kp - n_on
    xyz[n_on:] = (rng.random((n_cl, 3)) * np.array([1.0, 0.8, 0.8]) + np.array([-0.5, -0.4, 0.6])).astype(np.float32)
    u = np.clip(f * xyz[:, 0] / xyz[:, 2] + W / 2.0, 0, W - 1.001).astype(np.float32)
    v = np.clip(f * xyz[:, 1] / xyz[:, 2] + H / 2.0, 0, H - 1.001).astype(np.float32)
    perm = rng.permutation(n_kp)
    u, v, xyz, q_desc = u[perm], v[perm], xyz[perm], q_desc[perm]
    truth = np.full(n_kp, -1, np.int64)
    truth[:n_on] = rows
    truth = truth[perm]
    cloud = np.full((H, W, 3), np.nan, np.float32)
    nan_kp = rng.random(n_kp) < nan_frac
    ok = ~nan_kp
    cloud[v[ok].astype(np.int64), u[ok].astype(np.int64)] = xyz[ok]
    kp_xy = np.stack([u, v], axis=1).astype(np.float32)
    return dict(kp_xy=kp_xy, q_desc=q_desc, cloud=cloud, truth_rows=truth)


def make_image(frame=0, H=480, W=640, n_rect=2000):
    """Grey test image for stage A: mid-grey, random-contrast rectangles, sigma=2 Gaussian noise."""
    rng = np.random.Generator(np.random.PCG64(IMAGE_SEED + frame))
    img = np.full((H, W), 128.0, np.float32)
    x0 = rng.integers(0, W, n_rect); y0 = rng.integers(0, H, n_rect)
    w = rng.integers(4, 60, n_rect); h = rng.integers(4, 60, n_rect)
    val = rng.integers(0, 256, n_rect)
    for i in range(n_rect):
        img[y0[i]:y0[i] + h[i], x0[i]:x0[i] + w[i]] = val[i]
    img += rng.normal(0, 2.0, (H, W))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def random_graph_edges(n, p, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    iu = np.triu_indices(n, 1)
    keep = rng.random(len(iu[0])) < p
    return np.stack([iu[0][keep], iu[1][keep]], axis=1).astype(np.uint32)


def make_verify_scene(n_kp, n_objects=6, per_object=400, visible=((1, 0.30),), matches_per_kp=5, seed=0, H=480, W=640,
                      f=525.0, noise=0.002, nan_frac=0.10, true_match_rank=0, placements=None):
    """A hard scene for stage C, with the matches given directly (SURVEY probe P4 style): every keypoint carries
    `matches_per_kp` matches; a keypoint on a visible object has its true match plus random distractors, a clutter
    keypoint has only distractors. `visible` = ((object, fraction of keypoints), ...), each object at its own pose.
    Returns dict(kp_xy, cloud, row_ptr, matches(DMATCH), matches_xyz, spans, poses{obj: (R, t)})."""
    from .capi import DMATCH_DTYPE
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    model = (rng.random((n_objects, per_object, 3)) - 0.5) * np.array([0.20, 0.15, 0.10])
    model = model.astype(np.float32)
    spans = np.sqrt((((model.max(1) - model.min(1)).astype(np.float32)) ** 2).sum(1, dtype=np.float32)).astype(np.float32)
    xyz = (rng.random((n_kp, 3)) * np.array([1.0, 0.8, 0.8]) + np.array([-0.5, -0.4, 0.6])).astype(np.float32)
    owner = np.full(n_kp, -1, np.int64)
    src = np.zeros(n_kp, np.int64)
    poses = {}
    start = 0
    for vi, (obj, frac) in enumerate(visible):
        cnt = int(round(n_kp * frac))
        ang = 0.7 + 0.9 * vi
        t = np.array([0.05 - 0.35 * vi, -0.02 + 0.2 * vi, 0.8 + 0.15 * vi], np.float32)
        if placements is not None:                             # (angle, (tx, ty, tz)) per visible object, in the camera's view
            ang, t = placements[vi][0], np.asarray(placements[vi][1], np.float32)
        c, s_ = np.cos(ang), np.sin(ang)
        R = np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float32)
        rows = rng.choice(per_object, size=min(cnt, per_object), replace=False)
        cnt = len(rows)
        xyz[start:start + cnt] = (model[obj][rows] @ R.T + t + rng.normal(0, noise, (cnt, 3))).astype(np.float32)
        owner[start:start + cnt] = obj
        src[start:start + cnt] = rows
        poses[obj] = (R, t)
        start += cnt
    u = np.clip(f * xyz[:, 0] / xyz[:, 2] + W / 2.0, 0, W - 1.001).astype(np.float32)
    v = np.clip(f * xyz[:, 1] / xyz[:, 2] + H / 2.0, 0, H - 1.001).astype(np.float32)
    perm = rng.permutation(n_kp)
    u, v, xyz, owner, src = u[perm], v[perm], xyz[perm], owner[perm], src[perm]
    cloud = np.full((H, W, 3), np.nan, np.float32)
    ok = rng.random(n_kp) >= nan_frac
    cloud[v[ok].astype(np.int64), u[ok].astype(np.int64)] = xyz[ok]
    kp_xy = np.stack([u, v], axis=1).astype(np.float32)
    matches = np.zeros(n_kp * matches_per_kp, DMATCH_DTYPE)
    mxyz = np.zeros((n_kp * matches_per_kp, 3), np.float32)
    row_ptr = (np.arange(n_kp + 1) * matches_per_kp).astype(np.uint32)
    for i in range(n_kp):
        objs = rng.integers(0, n_objects, matches_per_kp)
        rows = rng.integers(0, per_object, matches_per_kp)
        if owner[i] >= 0:
            objs[true_match_rank] = owner[i]
            rows[true_match_rank] = src[i]
        for j in range(matches_per_kp):
            m = i * matches_per_kp + j
            matches[m] = (i, rows[j], objs[j], float(rng.integers(0, 60)))
            mxyz[m] = model[objs[j]][rows[j]]
    return dict(kp_xy=kp_xy, cloud=cloud, row_ptr=row_ptr, matches=matches, matches_xyz=mxyz, spans=spans, poses=poses)


# ------------------------------------------------------------------------------------------ float descriptors (C4)
SIFT_DB_SEED = 1004


def make_sift_db(n_objects, per_object=5000, dim=128):
    """SIFT-like float DB of BASELINE configs[3]: f32 in [0, 255], rows L2-normalised to 512.
    Returns (desc f32[N, dim], pts f32[N, 3], obj_off u32[n_objects + 1])."""
    rng = np.random.Generator(np.random.PCG64(SIFT_DB_SEED))
    n = n_objects * per_object
    desc = rng.random((n, dim), dtype=np.float32) * np.float32(255.0)
    desc *= (np.float32(512.0) / np.linalg.norm(desc, axis=1, keepdims=True)).astype(np.float32)
    pts = ((rng.random((n, 3)) - 0.5) * np.array([0.20, 0.15, 0.10])).astype(np.float32)
    off = (np.arange(n_objects + 1) * per_object).astype(np.uint32)
    return np.ascontiguousarray(desc, np.float32), pts, off


def make_sift_queries(desc, n_q, frame=0, on_object=0.30, noise=12.0):
    """Queries: a fraction are DB rows plus Gaussian noise (true matches at distance ~ noise * sqrt(dim)), the rest
    random SIFT-like vectors. Returns (q f32[n_q, dim], truth_rows i64[n_q] with -1 for clutter)."""
    rng = np.random.Generator(np.random.PCG64(FRAME_SEED + 5000 + frame))
    dim = desc.shape[1]
    n_on = int(round(n_q * on_object))
    rows = rng.choice(desc.shape[0], size=n_on, replace=False)
    q = rng.random((n_q, dim), dtype=np.float32) * np.float32(255.0)
    q *= (np.float32(512.0) / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    q[:n_on] = desc[rows] + rng.normal(0, noise, (n_on, dim)).astype(np.float32)
    truth = np.full(n_q, -1, np.int64)
    truth[:n_on] = rows
    perm = rng.permutation(n_q)
    return np.ascontiguousarray(q[perm], np.float32), truth[perm]
