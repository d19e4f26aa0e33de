"""GPU parity of stage B (DescriptorMatcher) through the C ABI, against the CPU oracle.
Bit-exact: match indices, distances, radius truncation, gathered 3D points, spans."""
import numpy as np
import pytest

import oracle_lib as O
from tod_amd import capi, synth

pytestmark = pytest.mark.gpu


ENGINE = {"name": "auto"}     # the engine of the context under test; the sharded helper's own contexts follow it


@pytest.fixture(scope="module", params=["valu", "mfma", "mfma-whole", "mfma-split2", "mfma-split3"])
def ctx(request):
    """Every test of this module runs on both engines of the exact search: K4 (vector ALU) and K4x (matrix cores) -- the latter with
    its adaptive partial-distance elimination and with each of its three block forms forced (todhip_set_matcher_block_split)."""
    c = capi.Context(0)
    engine, _, form = request.param.partition("-")
    c.set_matcher_engine(engine)
    if form:
        c.set_matcher_block_split({"whole": 0, "split2": 2, "split3": 3}[form])
    ENGINE["name"] = engine
    yield c
    c.close()
    ENGINE["name"] = "auto"


@pytest.fixture(scope="module")
def ctx_auto():
    """The engine the library picks by launch shape (the big, oracle-bound tests run once, on that)."""
    c = capi.Context(0)
    ENGINE["name"] = "auto"
    yield c
    c.close()


def _assert_same(ctx, desc, pts, off, q, k, radius):
    sp = ctx.db_load(desc, pts, off)
    assert np.array_equal(sp, O.spans(pts, off))
    row_ptr, m, xyz = ctx.match(q, k, radius)
    rc, o_row_ptr, o_m, o_xyz = O.match(desc, off, pts, q, k, radius)
    assert rc == 0
    assert np.array_equal(row_ptr, o_row_ptr)
    for f in ("queryIdx", "trainIdx", "imgIdx", "distance"):
        assert np.array_equal(m[f], o_m[f]), f
    assert np.array_equal(xyz, o_xyz)
    return len(m)


def test_c1_single_object_orb500(ctx):
    """BASELINE configs[0]: ORB-500 frame vs a 1-object (5k) DB, k=5, radius 35 (conf/detection.ork:32-37)."""
    desc, pts, off = synth.make_db(1)
    fr = synth.make_frame(desc, pts, off, 500)
    n = _assert_same(ctx, desc, pts, off, fr["q_desc"], 5, 35)
    assert n > 100


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7, 8])
def test_all_k(ctx, k):
    desc, pts, off = synth.make_db_ragged([700, 1, 0, 333, 2049], seed=5)
    fr = synth.make_frame(desc, pts, off, 300, frame=k, visible_object=4)
    _assert_same(ctx, desc, pts, off, fr["q_desc"], k, 255)   # radius 255: nothing is cut for 256-bit rows


@pytest.mark.parametrize("nq", [1, 63, 64, 65, 255, 256, 257, 1000])
def test_ragged_query_counts(ctx, nq):
    desc, pts, off = synth.make_db(4, per_object=777)
    fr = synth.make_frame(desc, pts, off, nq, frame=nq, visible_object=2)
    _assert_same(ctx, desc, pts, off, fr["q_desc"], 2, 55)


@pytest.mark.parametrize("n_rows", [1, 2, 3, 4, 5, 7, 63, 64, 65, 257, 4099])
def test_tiny_and_odd_databases(ctx, n_rows):
    """fewer rows than k, tails that are not a multiple of the 4-row group, single tiles"""
    desc, pts, off = synth.make_db_ragged([n_rows], seed=n_rows)
    rng = np.random.Generator(np.random.PCG64(n_rows))
    q = rng.integers(0, 256, (70, 32), dtype=np.uint8)
    q[:min(70, n_rows)] = desc[:min(70, n_rows)]
    _assert_same(ctx, desc, pts, off, q, 5, 200)


def test_ties_resolve_by_ascending_row(ctx):
    """duplicated DB rows give equal distances: order must be (distance asc, global row asc) -- decision D1"""
    desc, pts, off = synth.make_db_ragged([5000, 5000, 5000], seed=21)
    for dup in (17, 4000, 5001, 9999, 14999):
        desc[dup] = desc[3]
    q = np.repeat(desc[3:4], 10, axis=0)
    q[1:, 0] ^= np.arange(1, 10, dtype=np.uint8)
    _assert_same(ctx, desc, pts, off, q, 8, 255)
    row_ptr, m, _ = ctx.match(q[:1], 6, 255)
    rows = off[m["imgIdx"]].astype(np.int64) + m["trainIdx"]
    assert rows.tolist() == [3, 17, 4000, 5001, 9999, 14999] and (m["distance"] == 0).all()


def test_radius_truncation_edges(ctx):
    desc, pts, off = synth.make_db(2, per_object=3000)
    fr = synth.make_frame(desc, pts, off, 200, frame=9, visible_object=1, flip_p=0.12)
    for radius in (1, 20, 30, 31, 35, 37, 38, 40, 45, 47, 48, 55, 70, 79, 80, 100, 128, 255, 256, 1000):   # 37|38, 47|48, 79|80: schedule changes
        _assert_same(ctx, desc, pts, off, fr["q_desc"], 5, radius)


@pytest.mark.parametrize("n_rows,k", [(5, 8), (33, 8), (40, 2), (95, 5)])
def test_no_radius_cut_on_short_tiles_keeps_only_real_rows(ctx, n_rows, k):
    """radius >= 256 (no cut: the threshold starts below every dot product) on DBs whose only tile is shorter than a step or
    ends in a partial one, with fewer rows than k in one case: the matrix-core engine's 'no pending block' marker must stay below
    that threshold too, or it would enter the lists as rows at distance 1023."""
    desc, pts, off = synth.make_db(1, per_object=n_rows)
    fr = synth.make_frame(desc, pts, off, 70, frame=3, visible_object=0, flip_p=0.3)
    for radius in (256, 1000):
        _assert_same(ctx, desc, pts, off, fr["q_desc"], k, radius)


def test_error_statuses(ctx):
    desc, pts, off = synth.make_db(1, per_object=100)
    ctx.db_load(desc, pts, off)
    q = desc[:4]
    with pytest.raises(capi.TodError) as e:
        ctx.match(q, 5, 0)          # radius 0: the reference indexes an empty vector (DescriptorMatcher.cpp:237)
    assert e.value.status == capi.EINVAL
    with pytest.raises(capi.TodError):
        ctx.match(q, 0, 35)
    with pytest.raises(capi.TodError):
        ctx.match(q, 9, 35)
    empty = capi.Context(0)
    with pytest.raises(capi.TodError) as e:
        empty.match(q, 5, 35)       # "No descriptors loaded" (DescriptorMatcher.cpp:204-208)
    assert e.value.status == capi.ENODB
    empty.close()
    row_ptr, m, xyz = ctx.match(np.zeros((0, 32), np.uint8), 5, 35)
    assert row_ptr.tolist() == [0] and len(m) == 0


def test_c2_100k_database(ctx):
    """BASELINE configs[1]: ORB-1000 vs 100k-descriptor DB, Hamming BF k=2."""
    desc, pts, off = synth.make_db(20)
    fr = synth.make_frame(desc, pts, off, 1000, frame=1, visible_object=7)
    n = _assert_same(ctx, desc, pts, off, fr["q_desc"], 2, 35)
    assert 250 <= n <= 330


def _shard_merge(desc, pts, off, q, k, radius, n_shards):
    import torch
    nq = q.shape[0]
    d_q = torch.from_numpy(q).cuda()
    keys_all = torch.empty((n_shards, nq, k), dtype=torch.int64, device="cuda")
    ctxs = []
    for s in range(n_shards):
        c = capi.Context(0)
        c.set_matcher_engine(ENGINE["name"])
        c.db_load(desc, pts, off, shard_rank=s, shard_count=n_shards)
        c.match_shard_device(d_q.data_ptr(), nq, k, radius, keys_all[s].data_ptr())
        c.synchronize()
        ctxs.append(c)
    counts = torch.empty(nq, dtype=torch.int32, device="cuda")
    m = torch.empty((nq * k, 4), dtype=torch.int32, device="cuda")
    xyz = torch.empty((nq * k, 3), dtype=torch.float32, device="cuda")
    ctxs[0].merge_shards_device(keys_all.data_ptr(), n_shards, nq, k, radius, counts.data_ptr(), m.data_ptr(),
                                xyz.data_ptr())
    ctxs[0].synchronize()
    infos = [c.db_info() for c in ctxs]
    for c in ctxs:
        c.close()
    counts = counts.cpu().numpy()
    m = m.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(nq, k)
    xyz = xyz.cpu().numpy().reshape(nq, k, 3)
    keep = np.arange(k)[None, :] < counts[:, None]
    return counts, m[keep], xyz[keep], infos


@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_sharded_equals_unsharded(ctx, n_shards):
    """8(e): object-aligned shards + merge with the order (distance asc, global row asc) == 1-GPU result"""
    desc, pts, off = synth.make_db_ragged([900, 50, 0, 1200, 700, 5, 333, 2000, 41, 800], seed=77)
    desc[4000] = desc[100]          # a tie that straddles shards
    fr = synth.make_frame(desc, pts, off, 500, frame=5, visible_object=3)
    q = fr["q_desc"]
    q[0] = desc[100]
    counts, m, xyz, infos = _shard_merge(desc, pts, off, q, 3, 60, n_shards)
    rc, o_row_ptr, o_m, o_xyz = O.match(desc, off, pts, q, 3, 60)
    assert np.array_equal(np.diff(o_row_ptr.astype(np.int64)), counts)
    for f in ("queryIdx", "trainIdx", "imgIdx", "distance"):
        assert np.array_equal(m[f], o_m[f]), f
    assert np.array_equal(xyz, o_xyz)
    # shards are contiguous, object aligned and cover the DB exactly once
    assert sum(i["shard_rows"] for i in infos) == off[-1]
    starts = [i["shard_first"] for i in infos]
    assert starts == sorted(starts) and all(s in set(off.tolist()) for s in starts)


def test_c3_full_size_properties(ctx):
    """BASELINE configs[2] on one GPU: ORB-1000 vs the 1M-descriptor DB. The oracle needs ~1e9 distances
    (seconds) for a subset; the rest is checked through properties that do not depend on size."""
    desc, pts, off = synth.make_db(200)
    fr = synth.make_frame(desc, pts, off, 1000, frame=2, visible_object=123)
    ctx.db_load(desc, pts, off)
    row_ptr, m, xyz = ctx.match(fr["q_desc"], 2, 255)
    assert np.array_equal(np.diff(row_ptr.astype(np.int64)), np.full(1000, 2))
    rows = off[m["imgIdx"]].astype(np.int64) + m["trainIdx"]
    # (1) every reported distance is the true Hamming distance of that pair
    lut = np.array([bin(i).count("1") for i in range(256)], np.uint32)
    true_d = lut[np.bitwise_xor(desc[rows], fr["q_desc"][m["queryIdx"]])].sum(axis=1)
    assert np.array_equal(true_d.astype(np.float32), m["distance"])
    # (2) lists ascend in (distance, row)
    d2 = m["distance"].reshape(1000, 2)
    r2 = rows.reshape(1000, 2)
    assert ((d2[:, 0] < d2[:, 1]) | ((d2[:, 0] == d2[:, 1]) & (r2[:, 0] < r2[:, 1]))).all()
    # (3) planted queries find their source row first
    planted = fr["truth_rows"] >= 0
    assert np.array_equal(r2[planted, 0], fr["truth_rows"][planted])
    # (4) gather is the model point of the row
    assert np.array_equal(xyz, pts[rows])
    # (5) bit-exact against the oracle on a 64-query subset
    sub = np.arange(0, 1000, 16)
    keys = O.knn_keys(desc, fr["q_desc"][sub], 2)
    assert np.array_equal(keys >> np.uint64(32), d2[sub].astype(np.uint64))
    assert np.array_equal(keys & np.uint64(0xFFFFFFFF), r2[sub].astype(np.uint64))


def test_shard_layout_matches_python_mirror(ctx):
    from tod_amd import sharded
    desc, pts, off = synth.make_db_ragged([900, 50, 0, 1200, 700, 5, 333, 2000, 41, 800], seed=77)
    for world in (1, 2, 3, 8):
        for r in range(world):
            c = capi.Context(0)
            c.db_load(desc, pts, off, shard_rank=r, shard_count=world)
            info = c.db_info()
            lo, hi, row_lo, row_hi = sharded.shard_bounds(off, r, world)
            assert (info["shard_first"], info["shard_rows"]) == (row_lo, row_hi - row_lo)
            c.close()


def test_c5_shape_1080p_orb2000_2m_rows(ctx_auto):
    ctx = ctx_auto
    """BASELINE configs[4] on one GPU: ORB-2000 frame vs a 2M-descriptor DB (400 objects x 5000), k=5, radius 35.
    Size-independent properties for all queries, the oracle for a subset."""
    desc, pts, off = synth.make_db(400)
    fr = synth.make_frame(desc, pts, off, 2000, frame=9, visible_object=321, H=1080, W=1920, f=1400.0)
    ctx.db_load(desc, pts, off)
    row_ptr, m, xyz = ctx.match(fr["q_desc"], 5, 35)
    rows = off[m["imgIdx"]].astype(np.int64) + m["trainIdx"]
    lut = np.array([bin(i).count("1") for i in range(256)], np.uint32)
    true_d = lut[np.bitwise_xor(desc[rows], fr["q_desc"][m["queryIdx"]])].sum(axis=1)
    assert np.array_equal(true_d.astype(np.float32), m["distance"]) and (m["distance"] <= 35).all()
    planted = np.flatnonzero(fr["truth_rows"] >= 0)
    has = np.diff(row_ptr.astype(np.int64))[planted] >= 1            # 8 % bit flips: a few planted rows end up > radius
    assert has.mean() > 0.99
    first = rows[row_ptr[planted[has]]]                              # planted queries: the source row comes first
    assert np.array_equal(first, fr["truth_rows"][planted[has]])
    assert np.array_equal(xyz, pts[rows])
    sub = np.arange(0, 2000, 125)                                    # 16 queries against all 2M rows on the CPU
    keys = O.knn_keys(desc, fr["q_desc"][sub], 5)
    for i, q in enumerate(sub):
        want = [int(kk) for kk in keys[i] if (int(kk) >> 32) <= 35]
        got = [(int(d) << 32) | int(r) for d, r in zip(m["distance"][row_ptr[q]:row_ptr[q + 1]], rows[row_ptr[q]:row_ptr[q + 1]])]
        assert got == want


def test_batch_of_frames_in_one_launch_equals_frame_by_frame(ctx):
    """The bench's launch shape: the descriptors of 16 frames share one pass over the DB (16 x Q queries in one
    todhip_match_device call). Every frame's slice must equal that frame's own call, and the oracle on two of them."""
    import torch
    desc, pts, off = synth.make_db(20)                                   # C2: 100k rows
    ctx.db_load(desc, pts, off)
    F, nq, k = 16, 500, 2
    frames = [synth.make_frame(desc, pts, off, nq, frame=40 + f, visible_object=f % 20) for f in range(F)]
    d_q = torch.from_numpy(np.concatenate([fr["q_desc"] for fr in frames])).cuda()
    cb = torch.empty(F * nq, dtype=torch.int32, device="cuda"); mb = torch.empty((F * nq * k, 4), dtype=torch.int32, device="cuda")
    xb = torch.empty((F * nq * k, 3), dtype=torch.float32, device="cuda")
    ctx.match_device(d_q.data_ptr(), F * nq, k, 35, cb.data_ptr(), mb.data_ptr(), xb.data_ptr())
    ctx.synchronize()
    c1 = torch.empty(nq, dtype=torch.int32, device="cuda"); m1 = torch.empty((nq * k, 4), dtype=torch.int32, device="cuda")
    x1 = torch.empty((nq * k, 3), dtype=torch.float32, device="cuda")
    for f in range(F):
        ctx.match_device(d_q[f * nq:].data_ptr(), nq, k, 35, c1.data_ptr(), m1.data_ptr(), x1.data_ptr())
        ctx.synchronize()
        cnt = c1.cpu().numpy()
        assert np.array_equal(cb[f * nq:(f + 1) * nq].cpu().numpy(), cnt)
        got = mb[f * nq * k:(f + 1) * nq * k].cpu().numpy().reshape(nq, k, 4).copy()
        one = m1.cpu().numpy().reshape(nq, k, 4).copy()
        got[:, :, 0] -= f * nq                                            # queryIdx counts from the start of the batch
        valid = np.arange(k)[None, :] < cnt[:, None]
        assert np.array_equal(got[valid], one[valid])
        assert np.array_equal(xb[f * nq * k:(f + 1) * nq * k].cpu().numpy().reshape(nq, k, 3)[valid],
                              x1.cpu().numpy().reshape(nq, k, 3)[valid])
    for f in (0, 9):
        rc, row_ptr, m, xyz = O.match(desc, off, pts, frames[f]["q_desc"], k, 35)
        assert np.array_equal(cb[f * nq:(f + 1) * nq].cpu().numpy(), np.diff(row_ptr.astype(np.int64)))


def test_ten_million_row_database_properties(ctx_auto):
    ctx = ctx_auto
    """Beyond the benchmark's size: 10M rows (2000 objects x 5000, 320 MB of descriptors) in one shard -- row indices past
    2^23, the tile count at its cap. Planted rows are found first, every distance is the pair's true distance, lists
    ascend, and 16 queries equal the oracle bit for bit. Then the same DB cut into 3 shards and merged."""
    desc, pts, off = synth.make_db(2000)
    assert desc.shape[0] == 10_000_000
    fr = synth.make_frame(desc, pts, off, 512, frame=3, visible_object=1777)
    planted = fr["truth_rows"] >= 0
    assert planted.sum() > 100 and fr["truth_rows"][planted].min() >= 1777 * 5000
    ctx.db_load(desc, pts, off)
    row_ptr, m, xyz = ctx.match(fr["q_desc"], 3, 255)
    assert np.array_equal(np.diff(row_ptr.astype(np.int64)), np.full(512, 3))
    rows = off[m["imgIdx"]].astype(np.int64) + m["trainIdx"]
    lut = np.array([bin(i).count("1") for i in range(256)], np.uint32)
    true_d = lut[np.bitwise_xor(desc[rows], fr["q_desc"][m["queryIdx"]])].sum(axis=1)
    assert np.array_equal(true_d.astype(np.float32), m["distance"])
    d3, r3 = m["distance"].reshape(512, 3), rows.reshape(512, 3)
    for a in range(2):
        assert ((d3[:, a] < d3[:, a + 1]) | ((d3[:, a] == d3[:, a + 1]) & (r3[:, a] < r3[:, a + 1]))).all()
    assert np.array_equal(r3[planted, 0], fr["truth_rows"][planted])
    assert np.array_equal(xyz, pts[rows])
    sub = np.arange(0, 512, 32)
    keys = O.knn_keys(desc, fr["q_desc"][sub], 3)
    assert np.array_equal(keys >> np.uint64(32), d3[sub].astype(np.uint64))
    assert np.array_equal(keys & np.uint64(0xFFFFFFFF), r3[sub].astype(np.uint64))
    counts, ms, xyzs, infos = _shard_merge(desc, pts, off, fr["q_desc"], 3, 255, 3)
    assert np.array_equal(ms["trainIdx"], m["trainIdx"]) and np.array_equal(ms["imgIdx"], m["imgIdx"])
    assert np.array_equal(ms["distance"], m["distance"]) and sum(i["shard_rows"] for i in infos) == 10_000_000


def test_bench_launch_shape_16000_queries_1m_rows(ctx):
    """The exact launch bench.py times: 16 frames x 1000 descriptors in one pass over the 1M-row DB, k = 2, radius 35
    (BASELINE configs[2] on one GPU). Size-independent properties for all 16 000 queries, the oracle bit for bit on 64."""
    import torch
    desc, pts, off = synth.make_db(200)
    ctx.db_load(desc, pts, off)
    F, nq, k, radius = 16, 1000, 2, 35
    frames = [synth.make_frame(desc, pts, off, nq, frame=f, visible_object=(17 * f + 3) % 200) for f in range(F)]
    q = np.concatenate([fr["q_desc"] for fr in frames])
    truth = np.concatenate([fr["truth_rows"] for fr in frames])
    n = F * nq
    d_q = torch.from_numpy(q).cuda()
    cnt = torch.zeros(n, dtype=torch.int32, device="cuda"); mm = torch.zeros((n * k, 4), dtype=torch.int32, device="cuda")
    xx = torch.zeros((n * k, 3), dtype=torch.float32, device="cuda")     # zeroed: slots beyond a query's count are not written
    ctx.match_device(d_q.data_ptr(), n, k, radius, cnt.data_ptr(), mm.data_ptr(), xx.data_ptr())
    ctx.synchronize()
    cnt = cnt.cpu().numpy(); m = mm.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(n, k); xyz = xx.cpu().numpy().reshape(n, k, 3)
    keep = np.arange(k)[None, :] < cnt[:, None]
    rows = off[m["imgIdx"]].astype(np.int64) + m["trainIdx"]
    lut = np.array([bin(i).count("1") for i in range(256)], np.uint32)
    qq = np.broadcast_to(np.arange(n)[:, None], (n, k))
    assert np.array_equal(m["queryIdx"][keep], qq[keep])
    true_d = lut[np.bitwise_xor(desc[rows[keep]], q[qq[keep]])].sum(axis=1)
    assert np.array_equal(true_d.astype(np.float32), m["distance"][keep]) and (true_d <= radius).all()
    both = cnt == 2
    assert ((m["distance"][both, 0] < m["distance"][both, 1]) |
            ((m["distance"][both, 0] == m["distance"][both, 1]) & (rows[both, 0] < rows[both, 1]))).all()
    planted = np.flatnonzero(truth >= 0)
    found = cnt[planted] >= 1
    assert found.mean() > 0.99 and np.array_equal(rows[planted[found], 0], truth[planted[found]])
    assert np.array_equal(xyz[keep], pts[rows[keep]])
    sub = np.arange(7, n, 250)                                         # 64 queries against all 1M rows on the CPU
    keys = O.knn_keys(desc, q[sub], k)
    for i, qi in enumerate(sub):
        want = [int(kk) for kk in keys[i] if (int(kk) >> 32) <= radius]
        got = [(int(d) << 32) | int(r) for d, r in zip(m["distance"][qi, :cnt[qi]], rows[qi, :cnt[qi]])]
        assert got == want


def test_c3_partition_1m_rows_8_shards_equals_unsharded(ctx):
    """BASELINE configs[2]'s own partition: the 1M-row DB cut into 8 object-aligned shards (125k rows each), every shard
    searched on its own, keys merged -- equal to the single-device result for a whole frame."""
    import torch
    desc, pts, off = synth.make_db(200)
    fr = synth.make_frame(desc, pts, off, 1000, frame=11, visible_object=42)
    q = fr["q_desc"]
    q[5] = desc[124_999]; q[6] = desc[125_000]                         # rows either side of a shard boundary
    counts, ms, xyzs, infos = _shard_merge(desc, pts, off, q, 2, 35, 8)
    assert [i["shard_rows"] for i in infos] == [125_000] * 8
    ctx.db_load(desc, pts, off)
    row_ptr, m, xyz = ctx.match(q, 2, 35)
    assert np.array_equal(np.diff(row_ptr.astype(np.int64)), counts)
    for f in ("queryIdx", "trainIdx", "imgIdx", "distance"):
        assert np.array_equal(ms[f], m[f]), f
    assert np.array_equal(xyzs, xyz)


@pytest.mark.parametrize("k", [1, 2, 5])
@pytest.mark.parametrize("ratio", [0.6, 0.8, 1.0])
def test_ratio_test_equals_its_definition(ctx, k, ratio):
    """N4: Lowe's ratio test on the two exact nearest neighbours (the block DescriptorMatcher.cpp:223-227 leaves empty),
    defined by orc_match_ratio -- parity unpinned by construction. Duplicated DB rows make d1 == d2 (ratio 1.0 must still
    reject them: the test is strict), k = 1 needs the second neighbour internally, and the radius applies afterwards."""
    desc, pts, off = synth.make_db_ragged([1500, 40, 900], seed=31)
    desc[2000] = desc[10]                                              # two identical rows: d1 == d2 for their queries
    fr = synth.make_frame(desc, pts, off, 300, frame=3, visible_object=0, flip_p=0.10)
    q = fr["q_desc"]
    q[0] = desc[10]
    ctx.db_load(desc, pts, off)
    ctx.set_ratio_test(ratio)
    try:
        for radius in (30, 60, 255):
            row_ptr, m, xyz = ctx.match(q, k, radius)
            rc, o_row_ptr, o_m, o_xyz = O.match(desc, off, pts, q, k, radius, ratio)
            assert rc == 0 and np.array_equal(row_ptr, o_row_ptr) and np.array_equal(xyz, o_xyz)
            for f in ("queryIdx", "trainIdx", "imgIdx", "distance"):
                assert np.array_equal(m[f], o_m[f]), f
            assert row_ptr[1] == row_ptr[0]                            # query 0 has two neighbours at distance 0: ambiguous
        plain = O.match(desc, off, pts, q, k, 255, 0.0)[1]               # (o_row_ptr is the radius-255 pass of the loop)
        assert int(o_row_ptr[-1]) < int(plain[-1])                       # the test removes something (query 0 at the least)
    finally:
        ctx.set_ratio_test(0.0)


def test_ratio_test_sharded_equals_unsharded(ctx):
    desc, pts, off = synth.make_db_ragged([900, 50, 0, 1200, 700, 5, 333, 2000, 41, 800], seed=77)
    fr = synth.make_frame(desc, pts, off, 400, frame=8, visible_object=3, flip_p=0.10)
    import torch
    nq, k, radius, n_shards = 400, 2, 55, 3
    d_q = torch.from_numpy(fr["q_desc"]).cuda()
    keys_all = torch.empty((n_shards, nq, k), dtype=torch.int64, device="cuda")
    ctxs = []
    for s in range(n_shards):
        c = capi.Context(0); c.set_matcher_engine(ENGINE["name"]); c.set_ratio_test(0.8)
        c.db_load(desc, pts, off, shard_rank=s, shard_count=n_shards)
        c.match_shard_device(d_q.data_ptr(), nq, k, radius, keys_all[s].data_ptr()); c.synchronize()
        ctxs.append(c)
    counts = torch.zeros(nq, dtype=torch.int32, device="cuda"); mm = torch.zeros((nq * k, 4), dtype=torch.int32, device="cuda")
    xx = torch.zeros((nq * k, 3), dtype=torch.float32, device="cuda")
    ctxs[0].merge_shards_device(keys_all.data_ptr(), n_shards, nq, k, radius, counts.data_ptr(), mm.data_ptr(), xx.data_ptr())
    ctxs[0].synchronize()
    with pytest.raises(capi.TodError):
        ctxs[0].match_shard_device(d_q.data_ptr(), nq, 1, radius, keys_all[0].data_ptr())   # k = 1 cannot carry the second neighbour
    for c in ctxs:
        c.close()
    rc, o_row_ptr, o_m, o_xyz = O.match(desc, off, pts, fr["q_desc"], k, radius, 0.8)
    cnt = counts.cpu().numpy()
    assert np.array_equal(cnt, np.diff(o_row_ptr.astype(np.int64)))
    keep = np.arange(k)[None, :] < cnt[:, None]
    m = mm.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(nq, k)[keep]
    for f in ("queryIdx", "trainIdx", "imgIdx", "distance"):
        assert np.array_equal(m[f], o_m[f]), f


def test_block_split_adapts_to_the_data():
    """The matrix-core engine's partial-distance elimination picks its block form from the launches' own statistics:
    rows that agree with the queries on the bit positions of the first 2 (3) of a block's 4 matrix instructions and differ on all
    the others make every block survive that split, so the context moves up a level after one report, holds it for 32 launches,
    probes one level down once, and goes on; on independent bits it stays with the 2-split. Results == oracle on every launch."""
    rng = np.random.default_rng(3)
    q1 = rng.integers(0, 256, 32, dtype=np.uint8)
    nq, n_rows = 1024, 20000

    def db_of(flip_words):                                   # matrix instruction s covers words s and 4 + s of a 256-bit row
        row = q1.copy().view(np.uint32)
        row[flip_words] ^= 0xFFFFFFFF
        desc = np.tile(row.view(np.uint8), (n_rows, 1))
        for r, nb in ((777, 3), (15000, 5), (15001, 5)):     # three true neighbours: ranks and ties as usual
            d = q1.copy()
            d[:nb] ^= 1
            desc[r] = d
        return desc

    cases = {"first two agree": (db_of([2, 3, 6, 7]), [2] + [3] * 32 + [2, 3]),
             "first three agree": (db_of([3, 7]), [2, 3] + [4] * 32 + [3, 4]),
             "independent bits": (rng.integers(0, 256, (n_rows, 32), dtype=np.uint8), [2] * 6)}
    pts = rng.standard_normal((n_rows, 3)).astype(np.float32)
    off = np.array([0, n_rows], dtype=np.int64)
    q = np.tile(q1, (nq, 1))
    q[5:40, 31] ^= 0x80                                      # a few queries one bit away from the rest
    for name, (desc, forms) in cases.items():
        c = capi.Context(0)
        c.set_matcher_engine("mfma")
        c.db_load(desc, pts, off)
        rc, o_row_ptr, o_m, _ = O.match(desc, off, pts, q, 2, 35)
        assert rc == 0
        seen = []
        for _ in forms:
            row_ptr, m, _ = c.match(q, 2, 35)
            seen.append(c.counters().last_block_split)
            assert np.array_equal(row_ptr, o_row_ptr), name
            for f in ("trainIdx", "distance"):
                assert np.array_equal(m[f], o_m[f]), (name, f)
        assert seen == forms, (name, seen)
        cnt = c.counters()
        assert cnt.k4x_half_blocks > 0 and cnt.k4x_half_blocks_completed <= cnt.k4x_half_blocks
        c.close()


def test_block_split_setter_takes_only_its_four_values():
    c = capi.Context(0)
    for ok in (-1, 0, 2, 3, -1):
        c.set_matcher_block_split(ok)
    for bad in (1, 4, -2, 64):
        with pytest.raises(Exception):
            c.set_matcher_block_split(bad)
    c.close()
