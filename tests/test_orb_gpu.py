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
    assert np.minimum(da, 360 - da).max() < 1e-2                      # reported angle: atan2f of two libraries
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
