"""The ecto adapter (adapter/ecto_cells.hpp) compiled against the mini_ecto test double: cell/tendril names on CPU,
and on the GPU a full DescriptorMatcher -> GuessGenerator frame whose outputs must equal the C-ABI results."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "adapter_test")


def _build():
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-o", EXE, os.path.join(ROOT, "tests", "adapter_test.cpp"),
           "-L" + os.path.join(ROOT, "tod_amd"), "-ltodhip", "-Wl,-rpath," + os.path.join(ROOT, "tod_amd"),
           "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)


def test_adapter_compiles_and_declares_reference_names():
    """tendril and parameter names/defaults of DescriptorMatcher.cpp:131-152 and GuessGenerator.cpp:71-99"""
    _build()
    out = subprocess.run([EXE, tempfile.gettempdir(), "declare-only"], capture_output=True, text=True)
    assert out.returncode == 0 and "declare ok" in out.stdout, out.stderr


@pytest.mark.gpu
def test_adapter_frame_equals_c_abi():
    from tod_amd import capi, synth
    if not os.path.exists(EXE):
        _build()
    desc, pts, off = synth.make_db_ragged([3000, 10, 2500], seed=5)
    fr = synth.make_frame(desc, pts, off, 500, frame=3, visible_object=2)
    image = synth.make_image(11)
    mask = np.zeros((480, 640), np.uint8); mask[100:400, 150:500] = 255
    with tempfile.TemporaryDirectory() as d:
        for name, arr in (("desc", desc), ("pts", pts), ("obj_off", off.astype(np.uint32)), ("q_desc", fr["q_desc"]),
                          ("kp_xy", fr["kp_xy"]), ("cloud", fr["cloud"]), ("image", image), ("mask", mask)):
            np.ascontiguousarray(arr).tofile(os.path.join(d, name + ".bin"))
        out = subprocess.run([EXE, d], capture_output=True, text=True)
        assert out.returncode == 0, out.stdout + out.stderr
        orb_out = {tag: (np.fromfile(os.path.join(d, "out_orb_kp%s.bin" % tag), np.float32).reshape(-1, 6),
                         np.fromfile(os.path.join(d, "out_orb_desc%s.bin" % tag), np.uint8).reshape(-1, 32)) for tag in ("", "_masked")}
        desc_bgr = np.fromfile(os.path.join(d, "out_orb_desc_bgr.bin"), np.uint8).reshape(-1, 32)
        m = np.fromfile(os.path.join(d, "out_matches.bin"), np.int32).reshape(-1, 3)
        dist = np.fromfile(os.path.join(d, "out_dist.bin"), np.float32)
        rt = np.fromfile(os.path.join(d, "out_poses.bin"), np.float32).reshape(-1, 12)
        inl = np.fromfile(os.path.join(d, "out_inliers.bin"), np.uint32)
        rt2 = np.fromfile(os.path.join(d, "out_poses_2d.bin"), np.float32).reshape(-1, 12)
        inl2 = np.fromfile(os.path.join(d, "out_inliers_2d.bin"), np.uint32)
    ctx = capi.Context(0)
    spans = ctx.db_load(desc, pts, off)
    row_ptr, gm, xyz = ctx.match(fr["q_desc"], 5, 35)          # the cell's k is 5 (DescriptorMatcher.cpp:211)
    assert np.array_equal(m[:, 0], gm["queryIdx"]) and np.array_equal(m[:, 1], gm["trainIdx"])
    assert np.array_equal(m[:, 2], gm["imgIdx"]) and np.array_equal(dist, gm["distance"])
    rng = capi.rng_new(1)
    poses = ctx.verify(fr["kp_xy"], fr["cloud"], row_ptr, gm, xyz, spans, 8, 2500, 0.01, rng)
    assert len(poses) == len(rt) == 1
    assert np.array_equal(rt[0, :9].reshape(3, 3), poses[0]["R"]) and np.array_equal(rt[0, 9:], poses[0]["t"])
    assert inl[0] == poses[0]["object"] == 2 and inl[1] == len(poses[0]["inliers"])
    assert np.array_equal(inl[2:], poses[0]["inliers"])
    # the cell's 2D-only branch (points3d empty, K connected) == todhip_verify_2d continuing the cell's rand() stream: the 3D call
    # above left `rng` where the cell's generator stood after its first frame, and the cloudless, K-less call in between draws nothing
    K = np.array([[525.0, 0, 320.0], [0, 525.0, 240.0], [0, 0, 1]], np.float32)
    poses2 = ctx.verify_2d(fr["kp_xy"], K, row_ptr, gm, xyz, spans, 8, 2500, 3.0, rng)
    assert len(poses2) == len(rt2) == 1 and poses2[0]["object"] == inl2[0] == 2
    assert np.array_equal(rt2[0, :9].reshape(3, 3), poses2[0]["R"]) and np.array_equal(rt2[0, 9:], poses2[0]["t"])
    assert inl2[1] == len(poses2[0]["inliers"]) and np.array_equal(inl2[2:], poses2[0]["inliers"])
    assert np.abs(poses2[0]["R"] - poses[0]["R"]).max() < 0.05 and np.abs(poses2[0]["t"] - poses[0]["t"]).max() < 0.02   # the same pose, from pixels alone
    # the FeatureDescriptor cell == the C ABI's ORB, with and without the cell's mask input
    for tag, mk in (("", None), ("_masked", mask)):
        kp, aux, de = ctx.orb(image, 500, 3, 1.2, mask=mk)
        kpf, dd = orb_out[tag]
        assert len(kp) == len(kpf) > 100 and np.array_equal(kpf[:, :2], kp) and np.array_equal(kpf[:, 2:], aux) and np.array_equal(dd, de)
        if mk is not None:
            l0 = aux[:, 3] == 0
            assert (mask[kp[l0, 1].astype(int), kp[l0, 0].astype(int)] != 0).all()
    # a BGR image with three equal channels converts to the gray image itself ((1868 + 9617 + 4899) g + 2^13 >> 14 == g)
    assert np.array_equal(desc_bgr, orb_out[""][1])
    ctx.close()
