"""GPU parity of stage A (ORB) against this repo's CPU restatement (oracle/orb_oracle.c). PARITY UNPINNED with
respect to the reference: there cv::ORB is third-party code outside the tree (detector.py:10,27) and OpenCV is not
in this image, so the only checks possible are bit-exactness against the restatement of the published algorithm
and properties of the result."""
import numpy as np
import pytest

import oracle_lib as O
from tod_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _same(ctx, img, n_features, n_levels, sf):
    kp, aux, desc = ctx.orb(img, n_features, n_levels, sf)
    o_kp, o_aux, o_desc, _ = O.orb(img, n_features, n_levels, sf)
    assert len(kp) == len(o_kp)
    assert np.array_equal(kp, o_kp)                                   # positions (level-0 pixels)
    assert np.array_equal(aux[:, [0, 2, 3]], o_aux[:, [0, 2, 3]])     # size, Harris response, octave: bit-exact
    da = np.abs(aux[:, 1] - o_aux[:, 1])
    assert len(kp) == 0 or np.minimum(da, 360 - da).max() < 1e-2      # reported angle: atan2f of two libraries
    assert np.array_equal(desc, o_desc)                               # 256-bit descriptors: bit-exact
    return kp, aux, desc


@pytest.mark.parametrize("frame", [0, 1, 2])
def test_vga_frame_orb1000(ctx, frame):
    """BASELINE C2/C3 front end: 640x480, ORB-1000, n_levels 3, scale 1.2 (conf/detection.ork:23-31)"""
    kp, aux, desc = _same(ctx, synth.make_image(frame), 1000, 3, 1.2)
    assert len(kp) == 1000 and set(np.unique(aux[:, 3])) == {0.0, 1.0, 2.0}
    assert kp[:, 0].min() >= 31 and kp[:, 0].max() <= 640 - 31 and kp[:, 1].min() >= 31 and kp[:, 1].max() <= 480 - 31


@pytest.mark.parametrize("shape,nf,nl,sf", [((480, 640), 500, 3, 1.2), ((240, 320), 300, 5, 1.2), ((1080, 1920), 2000, 3, 1.2),
                                            ((200, 333), 64, 8, 1.5), ((100, 100), 50, 2, 1.2)])
def test_other_shapes_and_pyramids(ctx, shape, nf, nl, sf):
    img = synth.make_image(5, H=shape[0], W=shape[1], n_rect=max(200, shape[0] * shape[1] // 150))
    _same(ctx, img, nf, nl, sf)


def test_flat_image_has_no_keypoints(ctx):
    kp, aux, desc = ctx.orb(np.full((480, 640), 77, np.uint8), 1000, 3, 1.2)
    assert len(kp) == 0


def test_descriptor_survives_in_plane_rotation(ctx):
    """property: rBRIEF is steered by the intensity centroid, so a 90-degree image rotation keeps descriptors close"""
    img = synth.make_image(7, H=480, W=480, n_rect=1500)
    kp0, aux0, d0 = ctx.orb(img, 400, 1, 1.2)
    rot = np.ascontiguousarray(np.rot90(img))                        # (x, y) -> (y, W-1-x)
    kp1, aux1, d1 = ctx.orb(rot, 400, 1, 1.2)
    pos1 = {(int(x), int(y)): i for i, (x, y) in enumerate(kp1)}
    lut = np.array([bin(i).count("1") for i in range(256)])
    dists = []
    for i, (x, y) in enumerate(kp0):
        j = pos1.get((int(y), 479 - int(x)))
        if j is not None:
            dists.append(int(lut[d0[i] ^ d1[j]].sum()))
    assert len(dists) > 200 and np.median(dists) < 40                # random pairs sit at ~128


def test_batch_of_frames_equals_frame_by_frame(ctx):
    """todhip_orb_batch_device: five different frames (one flat, with no keypoints) in the launches of one; every
    frame must come out exactly as from the single-frame call and as from the CPU restatement."""
    import torch
    imgs = [synth.make_image(10 + i) for i in range(4)] + [np.full((480, 640), 90, np.uint8)]
    F, cap = len(imgs), 1000
    d = torch.from_numpy(np.stack(imgs)).cuda()
    kp = torch.zeros((F, cap, 2), device="cuda"); aux = torch.zeros((F, cap, 4), device="cuda")
    desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device="cuda")
    n = ctx.orb_batch_device(d.data_ptr(), F, 480 * 640, 480, 640, 640, 1000, 3, 1.2, kp.data_ptr(), aux.data_ptr(),
                             desc.data_ptr(), cap)
    assert n == [1000, 1000, 1000, 1000, 0]
    for f in range(F - 1):
        o_kp, o_aux, o_desc, _ = O.orb(imgs[f], 1000, 3, 1.2)
        assert np.array_equal(kp[f].cpu().numpy(), o_kp) and np.array_equal(desc[f].cpu().numpy(), o_desc)
        s_kp, s_aux, s_desc = ctx.orb(imgs[f], 1000, 3, 1.2)
        assert np.array_equal(aux[f].cpu().numpy(), s_aux) and np.array_equal(desc[f].cpu().numpy(), s_desc)
    # a second batch of another size through the same context (graph re-captured for the new batch size)
    n2 = ctx.orb_batch_device(d[1:3].contiguous().data_ptr(), 2, 480 * 640, 480, 640, 640, 1000, 3, 1.2, kp.data_ptr(),
                              aux.data_ptr(), desc.data_ptr(), cap)
    assert n2 == [1000, 1000] and np.array_equal(desc[1].cpu().numpy(), O.orb(imgs[2], 1000, 3, 1.2)[2])


def test_c5_shape_batch_of_1080p_frames_orb2000(ctx):
    """BASELINE configs[4] front end on one GPU: a batch of 1920x1080 frames, ORB-2000 (3 levels, scale 1.2)."""
    import torch
    imgs = [synth.make_image(50 + i, H=1080, W=1920, n_rect=6000) for i in range(3)]
    F, cap = len(imgs), 2000
    d = torch.from_numpy(np.stack(imgs)).cuda()
    kp = torch.zeros((F, cap, 2), device="cuda"); aux = torch.zeros((F, cap, 4), device="cuda")
    desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device="cuda")
    n = ctx.orb_batch_device(d.data_ptr(), F, 1080 * 1920, 1080, 1920, 1920, 2000, 3, 1.2, kp.data_ptr(), aux.data_ptr(),
                             desc.data_ptr(), cap)
    assert n == [2000] * F
    for f in range(F):
        o_kp, o_aux, o_desc, _ = O.orb(imgs[f], 2000, 3, 1.2)
        assert np.array_equal(kp[f].cpu().numpy(), o_kp) and np.array_equal(desc[f].cpu().numpy(), o_desc)
        assert np.array_equal(aux[f].cpu().numpy()[:, [0, 2, 3]], o_aux[:, [0, 2, 3]])


def test_masked_orb_equals_cpu_restatement():
    """todhip_orb_masked (the FeatureDescriptor cell's `mask` input, detector.py:41): keypoints only where mask != 0,
    bit-identical to the CPU restatement with the same mask (parity with cv::ORB unpinned, as for the unmasked form)."""
    ctx = capi.Context(0)
    img = synth.make_image(5)
    mask = np.zeros(img.shape, np.uint8)
    mask[60:420, 90:560] = 255
    mask[200:260, 300:380] = 0                                          # a hole
    kp, aux, d = ctx.orb(img, 800, 3, 1.2, mask=mask)
    o_kp, o_aux, o_d, _ = O.orb(img, 800, 3, 1.2, mask=mask)
    assert len(kp) == len(o_kp) > 300
    assert np.array_equal(kp, o_kp) and np.array_equal(d, o_d)
    assert np.array_equal(aux[:, [0, 2, 3]], o_aux[:, [0, 2, 3]])     # size, Harris response, octave: bit-exact
    da = np.abs(aux[:, 1] - o_aux[:, 1])
    assert np.minimum(da, 360.0 - da).max() < 1e-2                     # the angle in degrees goes through atan2f (device libm)
    l0 = aux[:, 3] == 0                                                 # level-0 keypoints sit on the pixel the mask was sampled at
    assert l0.sum() > 100 and (mask[kp[l0, 1].astype(int), kp[l0, 0].astype(int)] != 0).all()
    ctx.close()
