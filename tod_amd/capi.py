"""ctypes binding of libtodhip.so (include/todhip.h).

The shared library is the product; this module only loads it and marshals numpy / torch
buffers to plain pointers. There is no CPU fallback: if the library or the GPU is missing,
loading or context creation raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# TODHIP_LIB_PATH: another build of the same library (the diagnostics builds of tools/*_ablate.sh, an A/B against HEAD) -- loaded
# instead of, never copied over, the product file
LIB_PATH = os.environ.get("TODHIP_LIB_PATH") or os.path.join(_PKG, "libtodhip.so")

OK, EINVAL, ENODB, EHIP, ECAPACITY, ERANGE, ENOMEM, ESCRATCH = 0, -1, -2, -3, -4, -5, -6, -7

DMATCH_DTYPE = np.dtype([("queryIdx", "<i4"), ("trainIdx", "<i4"), ("imgIdx", "<i4"), ("distance", "<f4")])


class TodObject(C.Structure):
    _fields_ = [("desc", C.c_void_p), ("pts_xyz", C.c_void_p), ("n", C.c_uint32)]


class Rng(C.Structure):
    _fields_ = [("s", C.c_uint32 * 31), ("f", C.c_uint32), ("b", C.c_uint32), ("draws", C.c_uint64)]


class VerifyParams(C.Structure):
    _fields_ = [("min_inliers", C.c_uint32), ("n_ransac_iterations", C.c_uint32), ("sensor_error", C.c_float)]


class Pose(C.Structure):
    _fields_ = [("object", C.c_uint32), ("R", C.c_float * 9), ("t", C.c_float * 3),
                ("inlier_begin", C.c_uint32), ("inlier_end", C.c_uint32)]


class RoundTrace(C.Structure):
    _fields_ = [("object", C.c_uint32), ("iterations", C.c_uint32), ("best_iteration", C.c_uint32),
                ("best_count", C.c_int32), ("draws_before", C.c_uint64), ("draws_after", C.c_uint64),
                ("n_inlier_kp", C.c_uint32), ("accepted", C.c_uint32)]


class Counters(C.Structure):
    _fields_ = [("db_rows", C.c_uint64), ("db_objects", C.c_uint64), ("last_nq", C.c_uint32), ("last_k", C.c_uint32),
                ("last_matches", C.c_uint32), ("last_objects_verified", C.c_uint32), ("last_rounds", C.c_uint32),
                ("last_hypotheses", C.c_uint32), ("last_gate_calls", C.c_uint32), ("last_poses", C.c_uint32),
                ("last_match_kernel_ms", C.c_double), ("sum_match_kernel_ms", C.c_double),
                ("n_match_kernel_launches", C.c_uint64), ("last_sprint_launches", C.c_uint32), ("last_sprint_rounds", C.c_uint32),
                ("last_verify_ticks", C.c_uint32), ("last_block_split", C.c_uint32), ("k4x_half_blocks", C.c_uint64),
                ("k4x_half_blocks_completed", C.c_uint64)]


# every symbol include/todhip.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = [
    "todhip_version", "todhip_create", "todhip_destroy", "todhip_stream", "todhip_last_hip_error",
    "todhip_synchronize", "todhip_get_counters", "todhip_set_cu_partition", "todhip_stream_create", "todhip_stream_destroy", "todhip_set_kernel_timing", "todhip_set_matcher_engine", "todhip_set_matcher_block_split", "todhip_set_ratio_test", "todhip_db_load", "todhip_db_load_device", "todhip_db_info",
    "todhip_match", "todhip_match_device", "todhip_match_shard_device", "todhip_merge_shards_device", "todhip_merge_shards_device_on",
    "todhip_rng_seed", "todhip_verify", "todhip_orb", "todhip_orb_masked", "todhip_test_clique", "todhip_test_clique_gate",
    "todhip_verify_trace", "todhip_test_adjacency", "todhip_test_consensus", "todhip_verify_device",
    "todhip_orb_device", "todhip_verify_device_depth", "todhip_orb_batch_device",
    "todhip_verify_batch_device", "todhip_verify_batch_device_depth",
    "todhip_match_l2", "todhip_match_l2_device",
    "todhip_model_begin", "todhip_model_add_observation", "todhip_model_finish", "todhip_model_device", "todhip_model_free",
    "todhip_rescale_depth", "todhip_rescale_depth_device", "todhip_verify_2d", "todhip_verify_2d_device", "todhip_verify_2d_batch_device", "todhip_set_lsh",
]

_lib = None


def build():
    """Compile libtodhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", os.path.join(_PKG, "csrc"), "-s", "-j4"], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libtodhip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        # PyTorch ships its own HIP runtime; when both live in one process (tests, bench) torch must bring
        # the runtime up first, otherwise its later initialisation finds no device.
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.todhip_stream.restype = C.c_void_p
        L.todhip_destroy.restype = None
        L.todhip_rng_seed.restype = None
        L.todhip_model_free.restype = None
        _lib = L
    return _lib


class TodError(RuntimeError):
    def __init__(self, status, what):
        super().__init__("%s failed with todhip_status %d" % (what, status))
        self.status = status


def _check(rc, what):
    if rc != OK:
        raise TodError(rc, what)


def _np_ptr(a):
    return C.c_void_p(a.ctypes.data)


class Context:
    """One HIP device + stream (todhip_ctx)."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        _check(lib().todhip_create(C.c_int(device), C.c_void_p(stream), C.byref(self._h)), "todhip_create")
        self._keep = None

    def close(self):
        if self._h:
            lib().todhip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return lib().todhip_stream(self._h)

    def synchronize(self):
        _check(lib().todhip_synchronize(self._h), "todhip_synchronize")

    def counters(self):
        c = Counters()
        _check(lib().todhip_get_counters(self._h, C.byref(c)), "todhip_get_counters")
        return c

    def set_matcher_engine(self, engine):
        """0 auto, 1 vector ALU (K4), 2 matrix cores (K4x): identical results"""
        _check(lib().todhip_set_matcher_engine(self._h, C.c_int({"auto": 0, "valu": 1, "mfma": 2}.get(engine, engine))),
               "todhip_set_matcher_engine")

    def set_matcher_block_split(self, split):
        """-1 adaptive (default), 0 whole blocks, 2 / 3: the matrix-core engine's blocks stop after that many of their 4 matrix
        instructions when no partial sum can still reach a threshold: identical results"""
        _check(lib().todhip_set_matcher_block_split(self._h, C.c_int(split)), "todhip_set_matcher_block_split")

    def set_ratio_test(self, ratio):
        """Lowe's ratio test on the two nearest neighbours (0 = off, the reference's effective setting)"""
        _check(lib().todhip_set_ratio_test(self._h, C.c_float(ratio)), "todhip_set_ratio_test")

    def set_lsh(self, n_tables, key_size=16, multi_probe_level=1):
        """LSH-approximate mode of the Hamming matcher (0 tables = the exact search, the default)."""
        _check(lib().todhip_set_lsh(self._h, C.c_uint32(n_tables), C.c_uint32(key_size), C.c_uint32(multi_probe_level)), "todhip_set_lsh")

    def set_kernel_timing(self, enable):
        _check(lib().todhip_set_kernel_timing(self._h, C.c_int(1 if enable else 0)), "todhip_set_kernel_timing")

    # ---------------------------------------------------------------- stage B
    def db_load(self, desc, pts, obj_off, shard_rank=0, shard_count=1):
        """desc u8[N,32] (binary, Hamming) or f32[N,128] (float, L2), pts f32[N,3], obj_off u32[n_obj+1] (rows of object o
        are obj_off[o]:obj_off[o+1])."""
        desc = np.ascontiguousarray(desc, np.float32 if np.asarray(desc).dtype == np.float32 else np.uint8)
        pts = np.ascontiguousarray(pts, np.float32)
        obj_off = np.asarray(obj_off, np.int64)
        n_obj = len(obj_off) - 1
        objs = (TodObject * max(n_obj, 1))()
        B = (desc.shape[1] if desc.ndim == 2 else 32) * desc.itemsize
        for o in range(n_obj):
            lo, hi = int(obj_off[o]), int(obj_off[o + 1])
            objs[o].desc = desc.ctypes.data + lo * B
            objs[o].pts_xyz = pts.ctypes.data + lo * 12
            objs[o].n = hi - lo
        spans = np.zeros(max(n_obj, 1), np.float32)
        rc = lib().todhip_db_load(self._h, objs, C.c_uint32(n_obj), C.c_uint32(B), C.c_uint32(shard_rank),
                                  C.c_uint32(shard_count), _np_ptr(spans))
        _check(rc, "todhip_db_load")
        return spans[:n_obj]

    def db_load_models(self, models, shard_rank=0, shard_count=1):
        """Object DB straight from trained models (capi.Model) in this device's memory: no host copy of descriptors or points
        (todhip_model_device + todhip_db_load_device). Returns (spans, obj_off)."""
        objs = (TodObject * len(models))()
        off = [0]
        for i, m in enumerate(models):
            d, p, n = m.device()
            objs[i].desc, objs[i].pts_xyz, objs[i].n = d, p, n
            off.append(off[-1] + n)
        spans = np.zeros(len(models), np.float32)
        _check(lib().todhip_db_load_device(self._h, objs, C.c_uint32(len(models)), C.c_uint32(32), C.c_uint32(shard_rank),
                                           C.c_uint32(shard_count), _np_ptr(spans)), "todhip_db_load_device")
        return spans, np.asarray(off, np.uint32)

    def db_info(self):
        tot, first, rows, nobj = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint32()
        _check(lib().todhip_db_info(self._h, C.byref(tot), C.byref(first), C.byref(rows), C.byref(nobj)),
               "todhip_db_info")
        return dict(total_rows=tot.value, shard_first=first.value, shard_rows=rows.value, n_objs=nobj.value)

    def match(self, q_desc, k, radius):
        """Host-buffer form. Returns (row_ptr u32[nq+1], matches DMATCH[n], xyz f32[n,3])."""
        q = np.ascontiguousarray(q_desc, np.uint8)
        nq = q.shape[0]
        row_ptr = np.zeros(nq + 1, np.uint32)
        m = np.zeros(max(nq * k, 1), DMATCH_DTYPE)
        xyz = np.zeros((max(nq * k, 1), 3), np.float32)
        rc = lib().todhip_match(self._h, _np_ptr(q), C.c_uint32(nq), C.c_uint32(k), C.c_uint32(radius),
                                _np_ptr(row_ptr), _np_ptr(m), _np_ptr(xyz))
        _check(rc, "todhip_match")
        n = int(row_ptr[nq])
        return row_ptr, m[:n].copy(), xyz[:n].copy()

    def match_l2(self, q_desc, k, radius):
        """Float descriptors, host-buffer form. Returns (row_ptr u32[nq+1], matches DMATCH[n], xyz f32[n,3])."""
        q = np.ascontiguousarray(q_desc, np.float32)
        nq = q.shape[0]
        row_ptr = np.zeros(nq + 1, np.uint32)
        m = np.zeros(max(nq * k, 1), DMATCH_DTYPE)
        xyz = np.zeros((max(nq * k, 1), 3), np.float32)
        rc = lib().todhip_match_l2(self._h, _np_ptr(q), C.c_uint32(nq), C.c_uint32(k), C.c_float(radius), _np_ptr(row_ptr),
                                   _np_ptr(m), _np_ptr(xyz))
        _check(rc, "todhip_match_l2")
        n = int(row_ptr[nq])
        return row_ptr, m[:n].copy(), xyz[:n].copy()

    def match_l2_device(self, d_q, nq, k, radius, d_counts, d_matches, d_xyz):
        rc = lib().todhip_match_l2_device(self._h, C.c_void_p(d_q), C.c_uint32(nq), C.c_uint32(k), C.c_float(radius),
                                          C.c_void_p(d_counts), C.c_void_p(d_matches), C.c_void_p(d_xyz))
        _check(rc, "todhip_match_l2_device")

    def match_device(self, d_q, nq, k, radius, d_counts, d_matches, d_xyz):
        """Device-pointer form (ints from tensor.data_ptr())."""
        rc = lib().todhip_match_device(self._h, C.c_void_p(d_q), C.c_uint32(nq), C.c_uint32(k), C.c_uint32(radius),
                                       C.c_void_p(d_counts), C.c_void_p(d_matches), C.c_void_p(d_xyz))
        _check(rc, "todhip_match_device")

    def match_shard_device(self, d_q, nq, k, radius, d_keys):
        rc = lib().todhip_match_shard_device(self._h, C.c_void_p(d_q), C.c_uint32(nq), C.c_uint32(k), C.c_uint32(radius),
                                             C.c_void_p(d_keys))
        _check(rc, "todhip_match_shard_device")

    def merge_shards_device(self, d_keys_all, n_shards, nq, k, radius, d_counts, d_matches, d_xyz):
        rc = lib().todhip_merge_shards_device(self._h, C.c_void_p(d_keys_all), C.c_uint32(n_shards), C.c_uint32(nq),
                                              C.c_uint32(k), C.c_uint32(radius), C.c_void_p(d_counts),
                                              C.c_void_p(d_matches), C.c_void_p(d_xyz))
        _check(rc, "todhip_merge_shards_device")

    def merge_shards_device_on(self, stream, d_keys_all, n_shards, nq, k, radius, d_counts, d_matches, d_xyz):
        """The merge on a stream of the caller's (it only reads the context's immutable tables)."""
        rc = lib().todhip_merge_shards_device_on(self._h, C.c_void_p(stream), C.c_void_p(d_keys_all), C.c_uint32(n_shards),
                                                 C.c_uint32(nq), C.c_uint32(k), C.c_uint32(radius), C.c_void_p(d_counts),
                                                 C.c_void_p(d_matches), C.c_void_p(d_xyz))
        _check(rc, "todhip_merge_shards_device_on")

    # ---------------------------------------------------------------- stage C
    def verify(self, kp_xy, cloud, row_ptr, matches, matches_xyz, spans, min_inliers, n_iter, err, rng,
               max_poses=64):
        kp = np.ascontiguousarray(kp_xy, np.float32)
        cloud = np.ascontiguousarray(cloud, np.float32)
        H, W = (cloud.shape[0], cloud.shape[1]) if cloud.ndim == 3 else (0, 0)
        row_ptr = np.ascontiguousarray(row_ptr, np.uint32)
        matches = np.ascontiguousarray(matches, DMATCH_DTYPE)
        mxyz = np.ascontiguousarray(matches_xyz, np.float32)
        sp = np.ascontiguousarray(spans, np.float32)
        prm = VerifyParams(min_inliers, n_iter, err)
        poses = (Pose * max_poses)()
        n_poses = C.c_uint32(max_poses)
        cap = max(len(kp), 1) * max_poses
        inl = np.zeros(cap, np.uint32)
        n_inl = C.c_uint32(cap)
        rc = lib().todhip_verify(self._h, _np_ptr(kp), C.c_uint32(len(kp)), _np_ptr(cloud), C.c_uint32(H),
                                 C.c_uint32(W), _np_ptr(row_ptr), _np_ptr(matches), _np_ptr(mxyz), _np_ptr(sp),
                                 C.c_uint32(len(sp)), C.byref(prm), C.byref(rng), poses, C.byref(n_poses),
                                 _np_ptr(inl), C.byref(n_inl))
        _check(rc, "todhip_verify")
        out = []
        for i in range(n_poses.value):
            p = poses[i]
            out.append(dict(object=int(p.object), R=np.array(p.R[:], np.float32).reshape(3, 3),
                            t=np.array(p.t[:], np.float32), inliers=inl[p.inlier_begin:p.inlier_end].copy()))
        return out

    def verify_2d(self, kp_xy, K, row_ptr, matches, matches_xyz, spans, min_inliers, n_iter, err_px, rng, max_poses=64):
        """The 2D-only branch (no cloud): todhip_verify_2d. err_px: reprojection threshold in pixels."""
        kp = np.ascontiguousarray(kp_xy, np.float32)
        K9 = np.ascontiguousarray(K, np.float32).reshape(9)
        row_ptr = np.ascontiguousarray(row_ptr, np.uint32)
        matches = np.ascontiguousarray(matches, DMATCH_DTYPE)
        mxyz = np.ascontiguousarray(matches_xyz, np.float32)
        sp = np.ascontiguousarray(spans, np.float32)
        prm = VerifyParams(min_inliers, n_iter, err_px)
        poses = (Pose * max_poses)()
        n_poses = C.c_uint32(max_poses)
        cap = max(len(kp), 1) * max_poses
        inl = np.zeros(cap, np.uint32)
        n_inl = C.c_uint32(cap)
        rc = lib().todhip_verify_2d(self._h, _np_ptr(kp), C.c_uint32(len(kp)), _np_ptr(K9), _np_ptr(row_ptr),
                                    matches.ctypes.data_as(C.c_void_p), _np_ptr(mxyz), _np_ptr(sp), C.c_uint32(len(sp)),
                                    C.byref(prm), C.byref(rng), poses, C.byref(n_poses), _np_ptr(inl), C.byref(n_inl))
        _check(rc, "todhip_verify_2d")
        return [dict(object=int(poses[i].object), R=np.array(poses[i].R[:], np.float32).reshape(3, 3),
                     t=np.array(poses[i].t[:], np.float32),
                     inliers=inl[poses[i].inlier_begin:poses[i].inlier_end].copy()) for i in range(n_poses.value)]

    def verify_2d_device(self, d_kp_xy, nq, K, d_counts, d_matches, d_xyz, k, spans, min_inliers, n_iter, err_px, rng, max_poses=64):
        K9 = np.ascontiguousarray(K, np.float32).reshape(9)
        sp = np.ascontiguousarray(spans, np.float32)
        prm = VerifyParams(min_inliers, n_iter, err_px)
        poses = (Pose * max_poses)()
        n_poses = C.c_uint32(max_poses)
        cap = max(nq, 1) * max_poses
        inl = np.zeros(cap, np.uint32)
        n_inl = C.c_uint32(cap)
        rc = lib().todhip_verify_2d_device(self._h, C.c_void_p(d_kp_xy), C.c_uint32(nq), _np_ptr(K9), C.c_void_p(d_counts),
                                           C.c_void_p(d_matches), C.c_void_p(d_xyz), C.c_uint32(k), _np_ptr(sp), C.c_uint32(len(sp)),
                                           C.byref(prm), C.byref(rng), poses, C.byref(n_poses), _np_ptr(inl), C.byref(n_inl))
        _check(rc, "todhip_verify_2d_device")
        return [dict(object=int(poses[i].object), R=np.array(poses[i].R[:], np.float32).reshape(3, 3),
                     t=np.array(poses[i].t[:], np.float32),
                     inliers=inl[poses[i].inlier_begin:poses[i].inlier_end].copy()) for i in range(n_poses.value)]

    def verify_2d_batch_device(self, n_frames, d_kp_xy, nq, K, d_counts, d_matches, d_xyz, k, spans, min_inliers, n_iter, err_px, rngs,
                               max_poses=16):
        """rngs: ctypes array (Rng * n_frames). Returns a list (per frame) of lists of pose dicts."""
        K9 = np.ascontiguousarray(K, np.float32).reshape(9)
        sp = np.ascontiguousarray(spans, np.float32)
        prm = VerifyParams(min_inliers, n_iter, err_px)
        cap_p = max_poses * n_frames
        poses = (Pose * cap_p)()
        n_poses = C.c_uint32(cap_p)
        pose_ptr = (C.c_uint32 * (n_frames + 1))()
        cap = max(nq, 1) * cap_p
        inl = np.zeros(cap, np.uint32)
        n_inl = C.c_uint32(cap)
        rc = lib().todhip_verify_2d_batch_device(self._h, C.c_uint32(n_frames), C.c_void_p(d_kp_xy), C.c_uint32(nq), _np_ptr(K9),
                                                 C.c_void_p(d_counts), C.c_void_p(d_matches), C.c_void_p(d_xyz), C.c_uint32(k), _np_ptr(sp),
                                                 C.c_uint32(len(sp)), C.byref(prm), rngs, poses, C.byref(n_poses), pose_ptr, _np_ptr(inl),
                                                 C.byref(n_inl))
        _check(rc, "todhip_verify_2d_batch_device")
        return [[dict(object=int(poses[i].object), R=np.array(poses[i].R[:], np.float32).reshape(3, 3), t=np.array(poses[i].t[:], np.float32),
                      inliers=inl[poses[i].inlier_begin:poses[i].inlier_end].copy()) for i in range(pose_ptr[f], pose_ptr[f + 1])]
                for f in range(n_frames)]

    def verify_device(self, d_kp_xy, nq, d_cloud, H, W, d_counts, d_matches, d_xyz, k, spans, min_inliers, n_iter,
                      err, rng, max_poses=64):
        """Device-pointer form (ints from tensor.data_ptr()); poses are returned on the host."""
        sp = np.ascontiguousarray(spans, np.float32)
        prm = VerifyParams(min_inliers, n_iter, err)
        poses = (Pose * max_poses)()
        n_poses = C.c_uint32(max_poses)
        cap = max(nq, 1) * max_poses
        if getattr(self, "_inl_cap", 0) < cap:
            self._inl = np.zeros(cap, np.uint32)
            self._inl_cap = cap
        inl = self._inl
        n_inl = C.c_uint32(cap)
        rc = lib().todhip_verify_device(self._h, C.c_void_p(d_kp_xy), C.c_uint32(nq), C.c_void_p(d_cloud),
                                        C.c_uint32(H), C.c_uint32(W), C.c_void_p(d_counts), C.c_void_p(d_matches),
                                        C.c_void_p(d_xyz), C.c_uint32(k), _np_ptr(sp), C.c_uint32(len(sp)),
                                        C.byref(prm), C.byref(rng), poses, C.byref(n_poses), _np_ptr(inl),
                                        C.byref(n_inl))
        _check(rc, "todhip_verify_device")
        out = []
        for i in range(n_poses.value):
            p = poses[i]
            out.append(dict(object=int(p.object), R=np.array(p.R[:], np.float32).reshape(3, 3),
                            t=np.array(p.t[:], np.float32), inliers=inl[p.inlier_begin:p.inlier_end].copy()))
        return out

    def verify_device_depth(self, d_kp_xy, nq, d_depth, depth_is_u16, H, W, K, d_counts, d_matches, d_xyz, k, spans,
                            min_inliers, n_iter, err, rng, max_poses=64):
        sp = np.ascontiguousarray(spans, np.float32)
        K9 = np.ascontiguousarray(K, np.float32).reshape(9)
        prm = VerifyParams(min_inliers, n_iter, err)
        poses = (Pose * max_poses)()
        n_poses = C.c_uint32(max_poses)
        cap = max(nq, 1) * max_poses
        inl = np.zeros(cap, np.uint32)
        n_inl = C.c_uint32(cap)
        rc = lib().todhip_verify_device_depth(self._h, C.c_void_p(d_kp_xy), C.c_uint32(nq), C.c_void_p(d_depth),
                                              C.c_int(1 if depth_is_u16 else 0), C.c_uint32(H), C.c_uint32(W),
                                              _np_ptr(K9), C.c_void_p(d_counts), C.c_void_p(d_matches),
                                              C.c_void_p(d_xyz), C.c_uint32(k), _np_ptr(sp), C.c_uint32(len(sp)),
                                              C.byref(prm), C.byref(rng), poses, C.byref(n_poses), _np_ptr(inl),
                                              C.byref(n_inl))
        _check(rc, "todhip_verify_device_depth")
        return [dict(object=int(poses[i].object), R=np.array(poses[i].R[:], np.float32).reshape(3, 3),
                     t=np.array(poses[i].t[:], np.float32),
                     inliers=inl[poses[i].inlier_begin:poses[i].inlier_end].copy()) for i in range(n_poses.value)]

    def verify_batch_device(self, n_frames, d_kp_xy, nq, d_cloud, H, W, d_counts, d_matches, d_xyz, k, spans, min_inliers,
                            n_iter, err, rngs, max_poses=16, depth=None):
        """rngs: ctypes array (Rng * n_frames). depth = (d_depth, is_u16, K) selects the depth form (d_cloud ignored).
        Returns a list (per frame) of lists of pose dicts."""
        sp = np.ascontiguousarray(spans, np.float32)
        prm = VerifyParams(min_inliers, n_iter, err)
        cap_p = max_poses * n_frames
        poses = (Pose * cap_p)()
        n_poses = C.c_uint32(cap_p)
        pose_ptr = (C.c_uint32 * (n_frames + 1))()
        cap = max(nq, 1) * cap_p
        if getattr(self, "_inl_cap", 0) < cap:                # (a megabyte per call otherwise: kept per context, results are copied out)
            self._inl = np.zeros(cap, np.uint32)
            self._inl_cap = cap
        inl = self._inl
        n_inl = C.c_uint32(cap)
        if depth is None:
            rc = lib().todhip_verify_batch_device(self._h, C.c_uint32(n_frames), C.c_void_p(d_kp_xy), C.c_uint32(nq),
                                                  C.c_void_p(d_cloud), C.c_uint32(H), C.c_uint32(W), C.c_void_p(d_counts),
                                                  C.c_void_p(d_matches), C.c_void_p(d_xyz), C.c_uint32(k), _np_ptr(sp),
                                                  C.c_uint32(len(sp)), C.byref(prm), rngs, poses, C.byref(n_poses), pose_ptr,
                                                  _np_ptr(inl), C.byref(n_inl))
        else:
            d_depth, is_u16, K = depth
            K9 = np.ascontiguousarray(K, np.float32).reshape(9)
            rc = lib().todhip_verify_batch_device_depth(self._h, C.c_uint32(n_frames), C.c_void_p(d_kp_xy), C.c_uint32(nq),
                                                        C.c_void_p(d_depth), C.c_int(1 if is_u16 else 0), C.c_uint32(H),
                                                        C.c_uint32(W), _np_ptr(K9), C.c_void_p(d_counts), C.c_void_p(d_matches),
                                                        C.c_void_p(d_xyz), C.c_uint32(k), _np_ptr(sp), C.c_uint32(len(sp)),
                                                        C.byref(prm), rngs, poses, C.byref(n_poses), pose_ptr, _np_ptr(inl),
                                                        C.byref(n_inl))
        _check(rc, "todhip_verify_batch_device")
        out = []
        for f in range(n_frames):
            out.append([dict(object=int(poses[i].object), R=np.array(poses[i].R[:], np.float32).reshape(3, 3),
                             t=np.array(poses[i].t[:], np.float32),
                             inliers=inl[poses[i].inlier_begin:poses[i].inlier_end].copy())
                        for i in range(pose_ptr[f], pose_ptr[f + 1])])
        return out

    def verify_trace(self, cap=4096):
        arr = (RoundTrace * cap)()
        n = C.c_uint32(cap)
        _check(lib().todhip_verify_trace(self._h, arr, C.byref(n)), "todhip_verify_trace")
        return [arr[i] for i in range(n.value)]

    def test_adjacency(self, train, query, kp_per_match, span, err):
        t = np.ascontiguousarray(train, np.float32)
        q = np.ascontiguousarray(query, np.float32)
        kp = np.ascontiguousarray(kp_per_match, np.float32)
        n = len(t)
        W = (n + 63) // 64
        phys = np.zeros((n, W), np.uint64)
        samp = np.zeros((n, W), np.uint64)
        rc = lib().todhip_test_adjacency(self._h, _np_ptr(t), _np_ptr(q), _np_ptr(kp), C.c_uint32(n), C.c_float(span),
                                         C.c_float(err), _np_ptr(phys), _np_ptr(samp))
        _check(rc, "todhip_test_adjacency")
        return phys, samp

    def test_consensus(self, train, query, kp_per_match, span, err, triples, stop_level=0, dbg_stride=0):
        t = np.ascontiguousarray(train, np.float32)
        q = np.ascontiguousarray(query, np.float32)
        kp = np.ascontiguousarray(kp_per_match, np.float32)
        tr = np.ascontiguousarray(triples, np.uint32).reshape(-1, 3)
        counts = np.zeros(len(tr), np.int32)
        dbg = np.zeros((len(tr), dbg_stride), np.uint32) if dbg_stride else None
        rc = lib().todhip_test_consensus(self._h, _np_ptr(t), _np_ptr(q), _np_ptr(kp), C.c_uint32(len(t)),
                                         C.c_float(span), C.c_float(err), _np_ptr(tr), C.c_uint32(len(tr)),
                                         C.c_uint32(stop_level), _np_ptr(counts),
                                         None if dbg is None else _np_ptr(dbg), C.c_uint32(dbg_stride))
        _check(rc, "todhip_test_consensus")
        return counts, dbg

    def test_clique(self, m, edges, minimal_size=0xFFFFFFFF, gate=False):
        e = np.ascontiguousarray(np.asarray(edges, np.uint32).reshape(-1, 2))
        out = np.zeros(3, np.uint32)
        fn = lib().todhip_test_clique_gate if gate else lib().todhip_test_clique
        rc = fn(self._h, C.c_uint32(m), _np_ptr(e), C.c_uint32(len(e)), C.c_uint32(minimal_size), _np_ptr(out))
        _check(rc, "todhip_test_clique")
        return int(out[0]), int(out[1]), int(out[2])

    # ---------------------------------------------------------------- stage A
    def orb(self, gray, n_features=1000, n_levels=3, scale_factor=1.2, pattern=None, mask=None):
        g = np.ascontiguousarray(gray, np.uint8)
        mk = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        H, W = g.shape
        kp = np.zeros((n_features, 2), np.float32)
        aux = np.zeros((n_features, 4), np.float32)
        desc = np.zeros((n_features, 32), np.uint8)
        n_out = C.c_uint32(n_features)
        pat = None if pattern is None else np.ascontiguousarray(pattern, np.int8)
        rc = lib().todhip_orb_masked(self._h, _np_ptr(g), None if mk is None else _np_ptr(mk), C.c_uint32(H), C.c_uint32(W),
                                     C.c_uint32(W), C.c_uint32(n_features), C.c_uint32(n_levels), C.c_float(scale_factor),
                                     None if pat is None else _np_ptr(pat), _np_ptr(kp), _np_ptr(aux), _np_ptr(desc),
                                     C.byref(n_out))
        _check(rc, "todhip_orb_masked")
        n = n_out.value
        return kp[:n].copy(), aux[:n].copy(), desc[:n].copy()


def _orb_device(self, d_gray, H, W, stride, n_features, n_levels, scale_factor, d_kp_xy, d_kp_aux, d_desc, cap):
    n_out = C.c_uint32(cap)
    rc = lib().todhip_orb_device(self._h, C.c_void_p(d_gray), C.c_uint32(H), C.c_uint32(W), C.c_uint32(stride),
                                 C.c_uint32(n_features), C.c_uint32(n_levels), C.c_float(scale_factor), None,
                                 C.c_void_p(d_kp_xy), C.c_void_p(d_kp_aux), C.c_void_p(d_desc), C.byref(n_out))
    _check(rc, "todhip_orb_device")
    return n_out.value


Context.orb_device = _orb_device


def _orb_batch_device(self, d_gray, n_frames, frame_stride, H, W, stride, n_features, n_levels, scale_factor, d_kp_xy, d_kp_aux,
                      d_desc, cap):
    n = (C.c_uint32 * n_frames)()
    rc = lib().todhip_orb_batch_device(self._h, C.c_void_p(d_gray), C.c_uint32(n_frames), C.c_uint64(frame_stride), C.c_uint32(H),
                                       C.c_uint32(W), C.c_uint32(stride), C.c_uint32(n_features), C.c_uint32(n_levels),
                                       C.c_float(scale_factor), None, C.c_void_p(d_kp_xy), C.c_void_p(d_kp_aux),
                                       C.c_void_p(d_desc), C.c_uint32(cap), n)
    _check(rc, "todhip_orb_batch_device")
    return list(n)


Context.orb_batch_device = _orb_batch_device


def _rescale_depth(self, depth, H, W, nearest=False):
    """rescale_depth (Trainer.cpp:62-81): depth [dH, dW] float32 metres / uint16 mm -> [H, W] float32 metres."""
    u16 = depth.dtype == np.uint16
    d = np.ascontiguousarray(depth, np.uint16 if u16 else np.float32)
    out = np.empty((H, W), np.float32)
    rc = lib().todhip_rescale_depth(self._h, _np_ptr(d), C.c_int(1 if u16 else 0), C.c_uint32(d.shape[0]), C.c_uint32(d.shape[1]),
                                    _np_ptr(out), C.c_uint32(H), C.c_uint32(W), C.c_int(1 if nearest else 0))
    _check(rc, "todhip_rescale_depth")
    return out


def _rescale_depth_device(self, d_depth_in, is_u16, dH, dW, d_depth_out, H, W, nearest=False):
    rc = lib().todhip_rescale_depth_device(self._h, C.c_void_p(d_depth_in), C.c_int(1 if is_u16 else 0), C.c_uint32(dH),
                                           C.c_uint32(dW), C.c_void_p(d_depth_out), C.c_uint32(H), C.c_uint32(W), C.c_int(1 if nearest else 0))
    _check(rc, "todhip_rescale_depth_device")


Context.rescale_depth = _rescale_depth
Context.rescale_depth_device = _rescale_depth_device


class Model:
    """One object's model being trained (todhip_model): add observations, then finish() -> (desc, pts)."""

    def __init__(self, ctx, capacity_rows=100000):
        self._ctx = ctx
        self._h = C.c_void_p()
        self._cap = capacity_rows
        _check(lib().todhip_model_begin(ctx._h, C.c_uint32(capacity_rows), C.byref(self._h)), "todhip_model_begin")

    def add_observation(self, gray, mask, depth, K, R, T, n_features=500, n_levels=8, scale_factor=1.2):
        g = np.ascontiguousarray(gray, np.uint8)
        mk = np.ascontiguousarray(mask, np.uint8)
        H, W = g.shape
        if tuple(depth.shape) != (H, W):                       # rescale_depth's resize branch (Trainer.cpp:73-80)
            depth = self._ctx.rescale_depth(depth, H, W)
        u16 = depth.dtype == np.uint16
        d = np.ascontiguousarray(depth, np.uint16 if u16 else np.float32)
        K9 = np.ascontiguousarray(K, np.float32).reshape(9)
        R9 = np.ascontiguousarray(R, np.float32).reshape(9)
        T3 = np.ascontiguousarray(T, np.float32).reshape(3)
        n = C.c_uint32(0)
        rc = lib().todhip_model_add_observation(self._ctx._h, self._h, _np_ptr(g), _np_ptr(mk), _np_ptr(d),
                                                C.c_int(1 if u16 else 0), C.c_uint32(H), C.c_uint32(W), _np_ptr(K9),
                                                _np_ptr(R9), _np_ptr(T3), C.c_uint32(n_features), C.c_uint32(n_levels),
                                                C.c_float(scale_factor), None, C.byref(n))
        _check(rc, "todhip_model_add_observation")
        return n.value

    def device(self):
        """(device pointer of the descriptors, device pointer of the points, rows) -- valid until close()."""
        d, p, n = C.c_void_p(), C.c_void_p(), C.c_uint32()
        _check(lib().todhip_model_device(self._ctx._h, self._h, C.byref(d), C.byref(p), C.byref(n)), "todhip_model_device")
        return d.value, p.value, n.value

    def finish(self):
        desc = np.zeros((self._cap, 32), np.uint8)
        pts = np.zeros((self._cap, 3), np.float32)
        n = C.c_uint32(self._cap)
        _check(lib().todhip_model_finish(self._ctx._h, self._h, _np_ptr(desc), _np_ptr(pts), C.byref(n)), "todhip_model_finish")
        return desc[:n.value].copy(), pts[:n.value].copy()

    def close(self):
        if self._h:
            lib().todhip_model_free(self._ctx._h, self._h)
            self._h = C.c_void_p()


def set_cu_partition(latency_cus):
    """Reserve the device's last `latency_cus` compute units for latency streams (todhip_set_cu_partition); 0 = none."""
    _check(lib().todhip_set_cu_partition(C.c_uint32(latency_cus)), "todhip_set_cu_partition")


def stream_create(device=0, latency=False):
    """A HIP stream handle (int) of the given kind, honouring the CU partition (todhip_stream_create)."""
    out = C.c_void_p()
    _check(lib().todhip_stream_create(C.c_int(device), C.c_int(1 if latency else 0), C.byref(out)), "todhip_stream_create")
    return out.value


def rng_new(seed=1):
    r = Rng()
    lib().todhip_rng_seed(C.byref(r), C.c_uint32(seed))
    return r
