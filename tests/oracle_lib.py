"""ctypes loader for the CPU oracle (oracle/libtod_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench.py. Nothing under tod_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "libtod_oracle.so")


class DMatch(C.Structure):
    _fields_ = [("queryIdx", C.c_int32), ("trainIdx", C.c_int32), ("imgIdx", C.c_int32), ("distance", C.c_float)]


class Rng(C.Structure):
    _fields_ = [("s", C.c_uint32 * 31), ("f", C.c_uint32), ("b", C.c_uint32), ("draws", C.c_uint64)]


class VerifyParams(C.Structure):
    _fields_ = [("min_inliers", C.c_uint32), ("n_ransac_iterations", C.c_uint32), ("sensor_error", C.c_float)]


class Pose(C.Structure):
    _fields_ = [("object", C.c_uint32), ("R", C.c_float * 9), ("t", C.c_float * 3),
                ("inlier_begin", C.c_uint32), ("inlier_end", C.c_uint32)]


class RoundTrace(C.Structure):
    _fields_ = [("iterations", C.c_uint32), ("best_iteration", C.c_uint32), ("best_count", C.c_int32),
                ("draws_before", C.c_uint64), ("draws_after", C.c_uint64), ("n_model_inliers", C.c_uint32),
                ("n_final_inliers", C.c_uint32), ("growth_passes", C.c_uint32)]


DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])

_lib = None


def build():
    subprocess.run(["make", "-C", _ORACLE_DIR, "-s"], check=True)


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(_ORACLE_DIR, f) for f in ("tod_oracle.cpp", "orb_oracle.c", "train_oracle.c", "tod_oracle.h")]
    if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        build()
    L = C.CDLL(_LIB_PATH)
    L.orc_rng_next.restype = C.c_int32
    L.orc_clique.restype = C.c_uint32
    L.orc_cluster_new.restype = C.c_void_p
    L.orc_cluster_size.restype = C.c_uint32
    L.orc_cluster_valid.restype = C.c_uint32
    L.orc_cluster_draw.restype = C.c_uint32
    L.orc_cluster_consensus.restype = C.c_uint32
    L.orc_cluster_ransac.restype = C.c_uint32
    L.orc_match.restype = C.c_int
    L.orc_match_ratio.restype = C.c_int
    L.orc_verify.restype = C.c_int
    L.orc_cluster_kabsch.restype = C.c_int
    L.orb_detect.restype = C.c_uint32
    L.orb_detect_masked.restype = C.c_uint32
    L.train_observation.restype = C.c_uint32
    _lib = L
    return L


# ------------------------------------------------------------------------------------------ rng
def rng_new(seed=1):
    r = Rng()
    lib().orc_rng_seed(C.byref(r), C.c_uint32(seed))
    return r


def rng_next(r):
    return lib().orc_rng_next(C.byref(r))


# ------------------------------------------------------------------------------------------ stage B
def spans(pts, obj_off):
    pts = np.ascontiguousarray(pts, np.float32)
    obj_off = np.ascontiguousarray(obj_off, np.uint32)
    out = np.zeros(len(obj_off) - 1, np.float32)
    lib().orc_spans(_p(pts, C.c_float), _p(obj_off, C.c_uint32), C.c_uint32(len(out)), _p(out, C.c_float))
    return out


def knn_keys(db, q, k):
    db = np.ascontiguousarray(db, np.uint8)
    q = np.ascontiguousarray(q, np.uint8)
    keys = np.zeros((q.shape[0], k), np.uint64)
    lib().orc_knn_keys(_p(db, C.c_uint8), C.c_uint64(db.shape[0]), C.c_uint32(db.shape[1]), _p(q, C.c_uint8),
                       C.c_uint32(q.shape[0]), C.c_uint32(k), _p(keys, C.c_uint64))
    return keys


def lsh_knn_keys(db, q, k, n_tables, key_size, level):
    """oracle/lsh_oracle.c: keys over each query's LSH candidate set, and the sizes of those sets."""
    db = np.ascontiguousarray(db, np.uint8)
    q = np.ascontiguousarray(q, np.uint8)
    keys = np.zeros((q.shape[0], k), np.uint64)
    n_cand = np.zeros(q.shape[0], np.uint32)
    lib().orc_lsh_knn_keys(_p(db, C.c_uint8), C.c_uint64(db.shape[0]), _p(q, C.c_uint8), C.c_uint32(q.shape[0]), C.c_uint32(k),
                           C.c_uint32(n_tables), C.c_uint32(key_size), C.c_uint32(level), _p(keys, C.c_uint64), _p(n_cand, C.c_uint32))
    return keys, n_cand


def match(db, obj_off, db_pts, q, k, radius, ratio=0.0):
    db = np.ascontiguousarray(db, np.uint8)
    q = np.ascontiguousarray(q, np.uint8)
    obj_off = np.ascontiguousarray(obj_off, np.uint32)
    db_pts = np.ascontiguousarray(db_pts, np.float32)
    nq = q.shape[0]
    row_ptr = np.zeros(nq + 1, np.uint32)
    m = np.zeros(nq * k, DMATCH_DTYPE)
    xyz = np.zeros((nq * k, 3), np.float32)
    rc = lib().orc_match_ratio(_p(db, C.c_uint8), _p(obj_off, C.c_uint32), C.c_uint32(len(obj_off) - 1),
                               _p(db_pts, C.c_float), C.c_uint32(db.shape[1]), _p(q, C.c_uint8), C.c_uint32(nq),
                               C.c_uint32(k), C.c_uint32(radius), C.c_float(ratio), _p(row_ptr, C.c_uint32),
                               m.ctypes.data_as(C.c_void_p), _p(xyz, C.c_float))
    n = int(row_ptr[nq])
    return rc, row_ptr, m[:n].copy(), xyz[:n].copy()


# ------------------------------------------------------------------------------------------ clique
def clique(n, added, deleted=(), minimal_size=0xFFFFFFFF):
    added = np.ascontiguousarray(np.asarray(added, np.uint32).reshape(-1, 2))
    deleted = np.ascontiguousarray(np.asarray(deleted, np.uint32).reshape(-1, 2))
    out = np.zeros(max(n, 1), np.uint32)
    und = C.c_uint32(0)
    steps = C.c_uint32(0)
    sz = lib().orc_clique(C.c_uint32(n), _p(added, C.c_uint32), C.c_uint32(len(added)), _p(deleted, C.c_uint32),
                          C.c_uint32(len(deleted)), C.c_uint32(minimal_size), _p(out, C.c_uint32), C.byref(und),
                          C.byref(steps))
    return int(sz), out[:min(sz, n)].copy(), int(und.value), int(steps.value)


# ------------------------------------------------------------------------------------------ stage C stepwise
class Cluster:
    """One object's matches == one tod::AdjacencyRansac."""

    def __init__(self, train_xyz, query_xyz, query_idx):
        self.t = np.ascontiguousarray(train_xyz, np.float32)
        self.q = np.ascontiguousarray(query_xyz, np.float32)
        self.qi = np.ascontiguousarray(query_idx, np.uint32)
        self.n = len(self.qi)
        self.h = C.c_void_p(lib().orc_cluster_new(_p(self.t, C.c_float), _p(self.q, C.c_float),
                                                  _p(self.qi, C.c_uint32), C.c_uint32(self.n)))

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h and _lib is not None:                    # (at interpreter exit the module's globals may already be gone)
            try:
                _lib.orc_cluster_free(h)
            except Exception:
                pass

    def fill(self, kp_xy, span, err):
        kp = np.ascontiguousarray(kp_xy, np.float32)
        lib().orc_cluster_fill(self.h, _p(kp, C.c_float), C.c_uint32(len(kp)), C.c_float(span), C.c_float(err))

    def bits(self, which):
        wpr = (self.n + 63) // 64
        out = np.zeros((self.n, max(wpr, 1)), np.uint64)
        lib().orc_cluster_bits(self.h, C.c_int(which), _p(out, C.c_uint64), C.c_uint32(max(wpr, 1)))
        return out

    def valid(self):
        out = np.zeros(max(self.n, 1), np.uint32)
        k = lib().orc_cluster_valid(self.h, _p(out, C.c_uint32))
        return out[:k].copy()

    def draw(self, rng):
        s = np.zeros(3, np.uint32)
        k = lib().orc_cluster_draw(self.h, C.byref(rng), _p(s, C.c_uint32))
        return s[:k].copy()

    def consensus(self, samples3):
        s = np.ascontiguousarray(samples3, np.uint32)
        out = np.zeros(self.n + 3, np.uint32)
        gc, gs = C.c_uint32(0), C.c_uint32(0)
        k = lib().orc_cluster_consensus(self.h, _p(s, C.c_uint32), _p(out, C.c_uint32), C.byref(gc), C.byref(gs))
        return out[:k].copy(), int(gc.value), int(gs.value)

    def ransac(self, err, n_iter, rng, want_iters=False):
        kp = np.zeros(max(self.n, 1), np.uint32)
        R = np.zeros(9, np.float32)
        T = np.zeros(3, np.float32)
        tr = RoundTrace()
        mi = np.zeros(max(self.n, 1), np.uint32)
        counts = np.full(n_iter + 2, -(2 ** 31), np.int32)
        samples = np.zeros((n_iter + 2, 3), np.uint32)
        k = lib().orc_cluster_ransac(self.h, C.c_float(err), C.c_uint32(n_iter), C.byref(rng), _p(kp, C.c_uint32),
                                     _p(R, C.c_float), _p(T, C.c_float), C.byref(tr), _p(mi, C.c_uint32),
                                     _p(counts, C.c_int32) if want_iters else None,
                                     _p(samples, C.c_uint32) if want_iters else None)
        res = dict(inlier_kp=kp[:k].copy(), R=R.reshape(3, 3).copy(), T=T.copy(), trace=tr,
                   model_inliers=mi[:tr.n_model_inliers].copy())
        if want_iters:
            res["iter_counts"] = counts[:tr.iterations].copy()
            res["iter_samples"] = samples[:tr.iterations].copy()
        return res

    def invalidate_kp(self, kp):
        kp = np.ascontiguousarray(kp, np.uint32)
        lib().orc_cluster_invalidate_kp(self.h, _p(kp, C.c_uint32), C.c_uint32(len(kp)))

    def kabsch(self, idx):
        idx = np.ascontiguousarray(idx, np.uint32)
        R = np.zeros(9, np.float32)
        T = np.zeros(3, np.float32)
        rc = lib().orc_cluster_kabsch(self.h, _p(idx, C.c_uint32), C.c_uint32(len(idx)), _p(R, C.c_float),
                                      _p(T, C.c_float))
        return rc, R.reshape(3, 3).copy(), T.copy()


# ------------------------------------------------------------------------------------------ stage C whole frame
def verify(kp_xy, cloud, row_ptr, matches, matches_xyz, spans_per_obj, min_inliers, n_iter, err, rng,
           max_poses=64):
    kp = np.ascontiguousarray(kp_xy, np.float32)
    cloud = np.ascontiguousarray(cloud, np.float32)
    H, W = cloud.shape[0], cloud.shape[1]
    row_ptr = np.ascontiguousarray(row_ptr, np.uint32)
    matches = np.ascontiguousarray(matches, DMATCH_DTYPE)
    mxyz = np.ascontiguousarray(matches_xyz, np.float32)
    sp = np.ascontiguousarray(spans_per_obj, np.float32)
    prm = VerifyParams(min_inliers, n_iter, err)
    poses = (Pose * max_poses)()
    n_poses = C.c_uint32(max_poses)
    cap = max(len(kp), 1) * max_poses
    inl = np.zeros(cap, np.uint32)
    n_inl = C.c_uint32(cap)
    rounds = (RoundTrace * 1024)()
    n_rounds = C.c_uint32(1024)
    rc = lib().orc_verify(_p(kp, C.c_float), C.c_uint32(len(kp)), _p(cloud, C.c_float), C.c_uint32(H), C.c_uint32(W),
                          _p(row_ptr, C.c_uint32), matches.ctypes.data_as(C.c_void_p), _p(mxyz, C.c_float),
                          _p(sp, C.c_float), C.c_uint32(len(sp)), C.byref(prm), C.byref(rng), poses,
                          C.byref(n_poses), _p(inl, C.c_uint32), C.byref(n_inl), rounds, C.byref(n_rounds))
    out = []
    for i in range(n_poses.value):
        p = poses[i]
        out.append(dict(object=int(p.object), R=np.array(p.R[:], np.float32).reshape(3, 3),
                        t=np.array(p.t[:], np.float32), inliers=inl[p.inlier_begin:p.inlier_end].copy()))
    return rc, out, [rounds[i] for i in range(n_rounds.value)]


def verify_2d(kp_xy, K, row_ptr, matches, matches_xyz, spans_per_obj, min_inliers, n_iter, err_px, rng, max_poses=64):
    """oracle/pnp_oracle.c: the 2D-only branch (no cloud). Returns rc, poses, (best_hyp, best_count) per object."""
    kp = np.ascontiguousarray(kp_xy, np.float32)
    K9 = np.ascontiguousarray(K, np.float32).reshape(9)
    row_ptr = np.ascontiguousarray(row_ptr, np.uint32)
    matches = np.ascontiguousarray(matches, DMATCH_DTYPE)
    mxyz = np.ascontiguousarray(matches_xyz, np.float32)
    sp = np.ascontiguousarray(spans_per_obj, np.float32)
    prm = VerifyParams(min_inliers, n_iter, err_px)
    poses = (Pose * max_poses)()
    n_poses = C.c_uint32(max_poses)
    cap = max(len(kp), 1) * max_poses
    inl = np.zeros(cap, np.uint32)
    n_inl = C.c_uint32(cap)
    bh = np.zeros(len(sp), np.uint32); bc = np.zeros(len(sp), np.uint32)
    rc = lib().orc_verify_2d(_p(kp, C.c_float), C.c_uint32(len(kp)), _p(K9, C.c_float), _p(row_ptr, C.c_uint32),
                             matches.ctypes.data_as(C.c_void_p), _p(mxyz, C.c_float), _p(sp, C.c_float), C.c_uint32(len(sp)),
                             C.byref(prm), C.byref(rng), poses, C.byref(n_poses), _p(inl, C.c_uint32), C.byref(n_inl),
                             _p(bh, C.c_uint32), _p(bc, C.c_uint32))
    out = []
    for i in range(n_poses.value):
        p = poses[i]
        out.append(dict(object=int(p.object), R=np.array(p.R[:], np.float32).reshape(3, 3),
                        t=np.array(p.t[:], np.float32), inliers=inl[p.inlier_begin:p.inlier_end].copy()))
    return rc, out, (bh, bc)


# ------------------------------------------------------------------------------------------ stage A
def orb(gray, n_features=1000, n_levels=3, scale_factor=1.2, pattern=None, mask=None):
    """oracle/orb_oracle.c. Returns kp_xy f32[n,2], aux f32[n,4] (size, angle, response, octave), desc u8[n,32],
    lvl_xy i32[n,2] (integer position inside the level)."""
    g = np.ascontiguousarray(gray, np.uint8)
    H, W = g.shape
    cap = n_features
    kp = np.zeros((cap, 2), np.float32)
    aux = np.zeros((cap, 4), np.float32)
    desc = np.zeros((cap, 32), np.uint8)
    lvl = np.zeros((cap, 2), np.int32)
    pat = None if pattern is None else np.ascontiguousarray(pattern, np.int8)
    mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    n = lib().orb_detect_masked(_p(g, C.c_uint8), None if mk is None else _p(mk, C.c_uint8), C.c_uint32(H), C.c_uint32(W),
                                C.c_uint32(W), C.c_uint32(n_features), C.c_uint32(n_levels), C.c_float(scale_factor),
                                None if pat is None else _p(pat, C.c_int8), C.c_uint32(cap), _p(kp, C.c_float),
                                _p(aux, C.c_float), _p(desc, C.c_uint8), _p(lvl, C.c_int32))
    return kp[:n].copy(), aux[:n].copy(), desc[:n].copy(), lvl[:n].copy()


def orb_default_pattern():
    pat = np.zeros((256, 4), np.int8)
    lib().orb_default_pattern(_p(pat, C.c_int8))
    return pat


# ------------------------------------------------------------------------------------------ N2: training
def train_observation(kp_xy, desc, mask, depth_m, K, R, T):
    kp = np.ascontiguousarray(kp_xy, np.float32)
    d = np.ascontiguousarray(desc, np.uint8)
    mk = np.ascontiguousarray(mask, np.uint8)
    dm = np.ascontiguousarray(depth_m, np.float32)
    H, W = mk.shape
    K9 = np.ascontiguousarray(K, np.float32).reshape(9)
    R9 = np.ascontiguousarray(R, np.float32).reshape(9)
    T3 = np.ascontiguousarray(T, np.float32).reshape(3)
    od = np.zeros((max(len(kp), 1), 32), np.uint8)
    op = np.zeros((max(len(kp), 1), 3), np.float32)
    src = np.zeros(max(len(kp), 1), np.uint32)
    n = lib().train_observation(_p(kp, C.c_float), _p(d, C.c_uint8), C.c_uint32(len(kp)), _p(mk, C.c_uint8),
                                _p(dm, C.c_float), C.c_uint32(H), C.c_uint32(W), _p(K9, C.c_float), _p(R9, C.c_float),
                                _p(T3, C.c_float), _p(od, C.c_uint8), _p(op, C.c_float), _p(src, C.c_uint32))
    return od[:n].copy(), op[:n].copy(), src[:n].copy()


def train_rescale_depth(depth, H, W, nearest=False):
    u16 = depth.dtype == np.uint16
    d = np.ascontiguousarray(depth, np.uint16 if u16 else np.float32)
    out = np.empty((H, W), np.float32)
    rc = lib().train_rescale_depth(d.ctypes.data_as(C.c_void_p), C.c_int(1 if u16 else 0), C.c_uint32(d.shape[0]),
                                   C.c_uint32(d.shape[1]), _p(out, C.c_float), C.c_uint32(H), C.c_uint32(W), C.c_int(1 if nearest else 0))
    return out if rc == 0 else None


def train_erode4(mask):
    mk = np.ascontiguousarray(mask, np.uint8)
    out = np.zeros_like(mk)
    lib().train_erode4(_p(mk, C.c_uint8), C.c_uint32(mk.shape[0]), C.c_uint32(mk.shape[1]), _p(out, C.c_uint8))
    return out


# ------------------------------------------------------------------------------------------ float descriptors (C4)
def l2_knn_keys(db, q, k):
    db = np.ascontiguousarray(db, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    keys = np.zeros((q.shape[0], k), np.uint64)
    lib().l2_knn_keys(_p(db, C.c_float), C.c_uint64(db.shape[0]), C.c_uint32(db.shape[1]), _p(q, C.c_float),
                      C.c_uint32(q.shape[0]), C.c_uint32(k), _p(keys, C.c_uint64))
    return keys


def l2_match(db, obj_off, db_pts, q, k, radius):
    db = np.ascontiguousarray(db, np.float32)
    q = np.ascontiguousarray(q, np.float32)
    obj_off = np.ascontiguousarray(obj_off, np.uint32)
    db_pts = np.ascontiguousarray(db_pts, np.float32)
    nq = q.shape[0]
    row_ptr = np.zeros(nq + 1, np.uint32)
    m = np.zeros(nq * k, DMATCH_DTYPE)
    xyz = np.zeros((nq * k, 3), np.float32)
    rc = lib().l2_match(_p(db, C.c_float), _p(obj_off, C.c_uint32), C.c_uint32(len(obj_off) - 1), _p(db_pts, C.c_float),
                        C.c_uint32(db.shape[1]), _p(q, C.c_float), C.c_uint32(nq), C.c_uint32(k), C.c_float(radius),
                        _p(row_ptr, C.c_uint32), m.ctypes.data_as(C.c_void_p), _p(xyz, C.c_float))
    n = int(row_ptr[nq])
    return rc, row_ptr, m[:n].copy(), xyz[:n].copy()
