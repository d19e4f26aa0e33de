"""The optional LSH-approximate mode (todhip_set_lsh, tod_amd/csrc/lsh.hip) against its definition oracle/lsh_oracle.c.
PARITY UNPINNED with respect to the reference: its index is OpenCV's FLANN (DescriptorMatcher.cpp:175-180), which is not in the
reference tree; the scheme (key_size-bit keys per table, multi-probe by flipped key bits, exact ranking of the bucket members) is
FLANN's published one, the choice of key bits this repo's own. Integer work: keys must be equal bit for bit."""
import numpy as np
import pytest

import oracle_lib as O
from tod_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _gpu_keys(ctx, q, k, radius=256):
    import torch
    d_q = torch.from_numpy(q).cuda()
    keys = torch.empty((len(q), k), dtype=torch.int64, device="cuda")
    ctx.match_shard_device(d_q.data_ptr(), len(q), k, radius, keys.data_ptr())
    ctx.synchronize()
    return keys.cpu().numpy().view(np.uint64)


@pytest.mark.parametrize("tables,key_size,level", [(10, 16, 1), (8, 24, 2), (1, 8, 0), (4, 12, 3), (32, 20, 1)])
def test_keys_equal_the_definition(ctx, tables, key_size, level):
    desc, pts, off = synth.make_db(12, per_object=4000)                              # 48k rows
    fr = synth.make_frame(desc, pts, off, 300, frame=2, visible_object=5)
    ctx.set_lsh(tables, key_size, level)                                             # before the load ...
    ctx.db_load(desc, pts, off)
    for k in (1, 2, 5, 8):
        want, n_cand = O.lsh_knn_keys(desc, fr["q_desc"], k, tables, key_size, level)
        assert np.array_equal(_gpu_keys(ctx, fr["q_desc"], k), want), k
    # approximate means approximate: some queries lose a true neighbour, and the candidate sets are a fraction of the DB
    exact = O.knn_keys(desc, fr["q_desc"], 2)
    want, n_cand = O.lsh_knn_keys(desc, fr["q_desc"], 2, tables, key_size, level)
    assert n_cand.mean() < len(desc)
    assert (want >= exact).all()                                                     # never better than the exact answer
    planted = fr["truth_rows"] >= 0                                                  # 8 % flipped bits: most planted rows share a bucket with their query
    if (tables, key_size, level) == (10, 16, 1):
        found = (want[planted, 0] & np.uint64(0xFFFFFFFF)) == fr["truth_rows"][planted].astype(np.uint64)
        assert found.mean() > 0.9


def test_switching_modes_and_full_match_path(ctx):
    """... or after it; 0 tables switches back to the exact search; the whole todhip_match path (radius cut, object lookup, 3D gather)
    runs on the index's lists."""
    desc, pts, off = synth.make_db_ragged([3000, 0, 1500, 7000, 20, 4000], seed=3)
    fr = synth.make_frame(desc, pts, off, 400, frame=1, visible_object=3)
    ctx.db_load(desc, pts, off)
    exact_rp, exact_m, _ = ctx.match(fr["q_desc"], 5, 35)
    ctx.set_lsh(10, 16, 1)
    row_ptr, m, xyz = ctx.match(fr["q_desc"], 5, 35)
    want, _ = O.lsh_knn_keys(desc, fr["q_desc"], 5, 10, 16, 1)
    rows = off[m["imgIdx"]].astype(np.int64) + m["trainIdx"]
    for q in range(len(fr["q_desc"])):
        w = [int(kk) for kk in want[q] if (int(kk) >> 32) <= 35]
        g = [(int(d) << 32) | int(r) for d, r in zip(m["distance"][row_ptr[q]:row_ptr[q + 1]], rows[row_ptr[q]:row_ptr[q + 1]])]
        assert g == w, q
    assert np.array_equal(xyz, pts[rows])
    assert row_ptr[-1] <= exact_rp[-1]
    ctx.set_lsh(0)
    rp2, m2, _ = ctx.match(fr["q_desc"], 5, 35)
    assert np.array_equal(rp2, exact_rp) and np.array_equal(m2, exact_m)


def test_sharded_index_merges_to_the_unsharded_one(ctx):
    import torch
    desc, pts, off = synth.make_db(9, per_object=3000)
    fr = synth.make_frame(desc, pts, off, 200, frame=4, visible_object=2)
    want, _ = O.lsh_knn_keys(desc, fr["q_desc"], 3, 6, 14, 1)
    d_q = torch.from_numpy(fr["q_desc"]).cuda()
    keys = torch.empty((3, 200, 3), dtype=torch.int64, device="cuda")
    for s in range(3):
        c = capi.Context(0)
        c.set_lsh(6, 14, 1)
        c.db_load(desc, pts, off, shard_rank=s, shard_count=3)
        c.match_shard_device(d_q.data_ptr(), 200, 3, 256, keys[s].data_ptr())
        c.synchronize(); c.close()
    merged = np.sort(keys.cpu().numpy().view(np.uint64).transpose(1, 0, 2).reshape(200, 9), axis=1)[:, :3]
    assert np.array_equal(merged, want)


def test_bad_parameters(ctx):
    for args in ((33, 16, 1), (4, 0, 0), (4, 25, 1), (4, 16, 4), (4, 2, 3)):
        with pytest.raises(capi.TodError):
            ctx.set_lsh(*args)
