"""Resource behaviour over many calls: device memory held by a context stops growing once its workspaces have seen the
largest shapes, contexts give everything back when closed, and results do not drift between the first and the last call."""
import numpy as np
import pytest

import oracle_lib as O
from tod_amd import capi, synth

pytestmark = pytest.mark.gpu


def _free_mb():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0] / 2**20


def test_repeated_calls_do_not_leak_and_do_not_drift():
    import torch
    torch.cuda.init()
    desc, pts, off = synth.make_db(8, per_object=3000)
    base = None
    ctx = capi.Context(0)
    spans = ctx.db_load(desc, pts, off)
    frames = [synth.make_frame(desc, pts, off, 600, frame=f, visible_object=f % 8) for f in range(4)]
    scenes = [synth.make_verify_scene(300 + 50 * i, visible=((1, 0.3),), seed=900 + i) for i in range(3)]
    imgs = [synth.make_image(70 + i, H=240 + 40 * i, W=320 + 64 * i) for i in range(3)]

    def one_round(i):
        fr = frames[i % 4]
        nq = 100 + 125 * (i % 5)
        row_ptr, m, xyz = ctx.match(fr["q_desc"][:nq], 1 + i % 5, 35 + 10 * (i % 3))
        sc = scenes[i % 3]
        poses = ctx.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], 8, 200, 0.01,
                           capi.rng_new(1 + i % 7))
        kp, aux, d = ctx.orb(imgs[i % 3], 200 + 100 * (i % 4), 1 + i % 4, 1.2)
        return (int(m["trainIdx"].astype(np.int64).sum()), len(poses), int(d.astype(np.int64).sum()))

    first = [one_round(i) for i in range(60)]                      # 60 = lcm of the shape periods: every combination once
    ctx.close()                                                     # the HIP runtime has loaded its code objects by now:
    base = _free_mb()                                               # what it keeps for itself is not the context's
    ctx = capi.Context(0)
    spans = ctx.db_load(desc, pts, off)
    assert [one_round(i) for i in range(60)] == first               # a fresh context reproduces the results
    warm = _free_mb()
    for rep in range(5):
        again = [one_round(i) for i in range(60)]
        assert again == first                                       # identical results on every repetition
    after = _free_mb()
    assert warm - after < 8.0, "device memory kept growing: %.1f MB over 300 calls" % (warm - after)
    ctx.close()
    assert base - _free_mb() < 64.0                                 # the context's DB and workspaces are returned


def test_contexts_can_be_created_and_destroyed_repeatedly():
    import torch
    torch.cuda.init()
    desc, pts, off = synth.make_db(2, per_object=2000)
    q = synth.make_frame(desc, pts, off, 200)["q_desc"]
    base = None
    for i in range(30):
        c = capi.Context(0)
        c.db_load(desc, pts, off)
        row_ptr, m, xyz = c.match(q, 2, 40)
        c.close()
        if i == 4:
            base = _free_mb()
    assert base - _free_mb() < 16.0
