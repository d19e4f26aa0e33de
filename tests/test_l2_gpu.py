"""GPU parity of the float-descriptor matcher (BASELINE configs[3]: SIFT-128, L2 brute force as a bf16 MFMA GEMM with
exact refinement) against oracle/l2_oracle.c. PARITY UNPINNED with respect to the reference: its matcher throws for
anything but FLANN-LSH on binary descriptors (DescriptorMatcher.cpp:154-188); the oracle only DEFINES the result."""
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


def _assert_same(ctx, desc, pts, off, q, k, radius):
    row_ptr, m, xyz = ctx.match_l2(q, k, radius)
    rc, o_row_ptr, o_m, o_xyz = O.l2_match(desc, off, pts, q, k, radius)
    assert rc == 0
    assert np.array_equal(row_ptr, o_row_ptr)
    for f in ("queryIdx", "trainIdx", "imgIdx"):
        assert np.array_equal(m[f], o_m[f]), f
    assert np.array_equal(m["distance"], o_m["distance"])            # sqrtf of the same sequential f32 sum: bit-exact
    assert np.array_equal(xyz, o_xyz)
    return row_ptr, m


@pytest.mark.parametrize("k", [1, 2, 5, 8])
def test_sift_like_db_exact_indices_and_distances(ctx, k):
    desc, pts, off = synth.make_sift_db(6, per_object=5000)                  # 30k rows
    q, truth = synth.make_sift_queries(desc, 300, frame=k)
    ctx.db_load(desc, pts, off)
    row_ptr, m = _assert_same(ctx, desc, pts, off, q, k, 1.0e9)             # no radius cut: plain top-k
    first = m[row_ptr[:-1][np.diff(row_ptr.astype(np.int64)) > 0]]
    planted = truth >= 0
    glob = off[first["imgIdx"]].astype(np.int64) + first["trainIdx"]
    assert (glob[planted] == truth[planted]).mean() > 0.99                   # the planted rows are the nearest ones


def test_radius_cut_ragged_and_tiny_dbs(ctx):
    desc, pts, off = synth.make_sift_db(3, per_object=700)                   # 2100 rows: not a multiple of the 32-row tile
    off = np.array([0, 700, 700, 2100], np.uint32)                           # an empty object in the middle
    q, truth = synth.make_sift_queries(desc, 77, frame=9)
    ctx.db_load(desc, pts, off)
    for radius in (50.0, 136.0, 200.0, 400.0, 1.0e9):                        # planted matches sit at ~12 * sqrt(128) = 136
        _assert_same(ctx, desc, pts, off, q, 3, radius)
    # fewer rows than k
    ctx.db_load(desc[:2], pts[:2], np.array([0, 2], np.uint32))
    row_ptr, m = _assert_same(ctx, desc[:2], pts[:2], np.array([0, 2], np.uint32), q[:10], 5, 1.0e9)
    assert (np.diff(row_ptr.astype(np.int64)) == 2).all()
    with pytest.raises(capi.TodError):
        ctx.match_l2(q, 9, 10.0)                                             # k > 8
    with pytest.raises(capi.TodError):
        ctx.match(np.zeros((4, 32), np.uint8), 2, 35)                        # the Hamming matcher refuses a float DB


def test_adversarial_near_ties_use_the_exact_fallback(ctx):
    """Many rows almost equidistant from the query: the bf16 filter cannot separate them, the candidate list of some
    queries overflows and those queries are redone by the exact scan; the result must not change."""
    rng = np.random.Generator(np.random.PCG64(77))
    base = rng.random(128, dtype=np.float32) * 100
    desc = (base[None, :] + rng.normal(0, 0.05, (4000, 128))).astype(np.float32)    # a tight cluster: 4000 near ties
    pts = rng.random((4000, 3)).astype(np.float32)
    off = np.array([0, 4000], np.uint32)
    q = (base[None, :] + rng.normal(0, 0.05, (40, 128))).astype(np.float32)
    ctx.db_load(desc, pts, off)
    _assert_same(ctx, desc, pts, off, q, 4, 1.0e9)


def test_integer_valued_descriptors_and_duplicates(ctx):
    """OpenCV's SIFT stores integer-valued floats 0..255; duplicates of a row must come out in row order."""
    rng = np.random.Generator(np.random.PCG64(5))
    desc = rng.integers(0, 256, (5000, 128)).astype(np.float32)
    desc[100] = desc[4000]; desc[2500] = desc[4000]                                     # three identical rows
    pts = rng.random((5000, 3)).astype(np.float32)
    off = np.array([0, 2000, 5000], np.uint32)
    q = desc[[4000, 7, 4999]] + 0.0
    ctx.db_load(desc, pts, off)
    row_ptr, m = _assert_same(ctx, desc, pts, off, q, 3, 1.0e9)
    assert list(off[m["imgIdx"][:3]] + m["trainIdx"][:3]) == [100, 2500, 4000] and (m["distance"][:3] == 0).all()


def test_c4_full_size_properties(ctx):
    """BASELINE configs[3] at full size (1000 SIFT-128 queries x 500k rows, k = 2): the oracle would take minutes on
    all of it, so 24 queries are checked against it bit for bit and the rest through size-independent properties --
    the planted row is the nearest one, distances ascend with ties in row order, and every reported distance is the
    exact f32 distance of the reported row."""
    desc, pts, off = synth.make_sift_db(100, per_object=5000)
    q, truth = synth.make_sift_queries(desc, 1000, frame=0)
    ctx.db_load(desc, pts, off)
    row_ptr, m, xyz = ctx.match_l2(q, 2, 1.0e9)
    assert (np.diff(row_ptr.astype(np.int64)) == 2).all()
    glob = off[m["imgIdx"]].astype(np.int64) + m["trainIdx"]
    g2, d2 = glob.reshape(1000, 2), m["distance"].reshape(1000, 2)
    planted = truth >= 0
    assert (g2[planted, 0] == truth[planted]).all()
    assert ((d2[:, 0] < d2[:, 1]) | ((d2[:, 0] == d2[:, 1]) & (g2[:, 0] < g2[:, 1]))).all()
    assert np.array_equal(xyz, pts[glob])
    pick = np.r_[np.flatnonzero(planted)[:12], np.flatnonzero(~planted)[:12]]
    rc, o_row_ptr, o_m, o_xyz = O.l2_match(desc, off, pts, q[pick], 2, 1.0e9)
    assert rc == 0
    sel = (pick[:, None] * 2 + np.arange(2)).ravel()
    for f in ("trainIdx", "imgIdx", "distance"):
        assert np.array_equal(m[f][sel], o_m[f]), f


def test_sixteen_frames_per_call_equal_their_single_frame_calls_and_the_oracle(ctx):
    """The batch form: F x Q queries of F frames share one pass over the DB (queryIdx counts through the call: frame f's query q
    is f Q + q). 16 frames x 1000 queries against the 500k-row DB: every frame's matches equal its own single-frame call's (the
    query preparation, the seed, the thresholds and the re-ranking are per query), and 32 queries spread over the frames equal the
    oracle bit for bit. A smaller DB (the classic two-GEMM path) with 5 frames of ragged size against the oracle in full."""
    desc, pts, off = synth.make_sift_db(100, per_object=5000)
    ctx.db_load(desc, pts, off)
    F, Q, k = 16, 1000, 2
    qs = [synth.make_sift_queries(desc, Q, frame=40 + f)[0] for f in range(F)]
    row_ptr, m, xyz = ctx.match_l2(np.concatenate(qs), k, 400.0)
    assert len(row_ptr) == F * Q + 1
    for f in range(F):
        rp1, m1, x1 = ctx.match_l2(qs[f], k, 400.0)
        lo, hi = int(row_ptr[f * Q]), int(row_ptr[(f + 1) * Q])
        assert np.array_equal(row_ptr[f * Q:(f + 1) * Q + 1] - row_ptr[f * Q], rp1), f
        assert np.array_equal(m["queryIdx"][lo:hi], m1["queryIdx"] + f * Q)
        for fld in ("trainIdx", "imgIdx", "distance"):
            assert np.array_equal(m[fld][lo:hi], m1[fld]), (f, fld)
        assert np.array_equal(xyz[lo:hi], x1)
    pick = np.arange(32) * 499 + 7                                            # two queries of every frame
    allq = np.concatenate(qs)
    rc, o_rp, o_m, o_xyz = O.l2_match(desc, off, pts, allq[pick], k, 400.0)
    assert rc == 0
    for j, g in enumerate(pick):
        lo, hi = int(row_ptr[g]), int(row_ptr[g + 1])
        olo, ohi = int(o_rp[j]), int(o_rp[j + 1])
        assert hi - lo == ohi - olo
        for fld in ("trainIdx", "imgIdx", "distance"):
            assert np.array_equal(m[fld][lo:hi], o_m[fld][olo:ohi]), (g, fld)
    desc, pts, off = synth.make_sift_db(4, per_object=1500)                  # 6000 rows: two GEMM passes
    ctx.db_load(desc, pts, off)
    qs = [synth.make_sift_queries(desc, n, frame=60 + i)[0] for i, n in enumerate((130, 1, 257, 64, 300))]
    _assert_same(ctx, desc, pts, off, np.concatenate(qs), 3, 500.0)


def test_one_gemm_path_with_overflowing_candidate_lists(ctx):
    """A DB large enough for the one-GEMM path (>= 64k rows: the seed of an evenly spaced sample IS the candidate threshold)
    that holds 1500 near-copies of one vector: queries next to it collect more candidates than a list holds and are redone
    by the exact scan; the other queries take the fast path. All of them equal the oracle."""
    rng = np.random.Generator(np.random.PCG64(123))
    n = 70000
    desc = (rng.random((n, 128)) * 200).astype(np.float32)
    base = (rng.random(128) * 200).astype(np.float32)
    where = rng.choice(n, 1500, replace=False)
    desc[where] = base[None, :] + rng.normal(0, 0.02, (1500, 128)).astype(np.float32)
    pts = rng.random((n, 3)).astype(np.float32)
    off = np.array([0, 20000, 20000, 70000], np.uint32)
    q = np.concatenate([base[None, :] + rng.normal(0, 0.02, (12, 128)).astype(np.float32),      # overflow -> exact scan
                        desc[rng.integers(0, n, 40)] + rng.normal(0, 3.0, (40, 128)).astype(np.float32),
                        (rng.random((30, 128)) * 200).astype(np.float32)]).astype(np.float32)
    ctx.db_load(desc, pts, off)
    for k in (1, 4, 8):
        _assert_same(ctx, desc, pts, off, q, k, 1.0e9)


def test_candidates_clustered_in_one_chunk_spill_from_slot_to_shared_list(ctx):
    """Pass 2 keeps a (query, DB chunk, lane half)'s first 8 candidates in a private slot and appends the rest to the query's shared
    list (csrc/l2.hip, CandSink). 60 near-copies of one vector in CONSECUTIVE rows land in one or two chunks, so both
    containers are in use for the queries next to it and neither overflows; pass 3 must merge them into the oracle's answer."""
    rng = np.random.Generator(np.random.PCG64(321))
    n = 70000
    desc = (rng.random((n, 128)) * 200).astype(np.float32)
    base = (rng.random(128) * 200).astype(np.float32)
    desc[31000:31060] = base[None, :] + rng.normal(0, 0.02, (60, 128)).astype(np.float32)
    pts = rng.random((n, 3)).astype(np.float32)
    off = np.array([0, 30000, 70000], np.uint32)
    q = np.concatenate([base[None, :] + rng.normal(0, 0.02, (6, 128)).astype(np.float32),
                        (rng.random((20, 128)) * 200).astype(np.float32)]).astype(np.float32)
    ctx.db_load(desc, pts, off)
    for k in (2, 8):
        row_ptr, m = _assert_same(ctx, desc, pts, off, q, k, 1.0e9)
        glob = (off[m["imgIdx"]].astype(np.int64) + m["trainIdx"]).reshape(len(q), k)
        assert ((glob[:6] >= 31000) & (glob[:6] < 31060)).all()

