"""CPU tests of the oracle itself: golden vectors of the reference's own tests, libc rand(),
and independent numpy restatements of the integer parts. No GPU needed."""
import ctypes
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from tod_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _graph(spec):
    n = spec["n"]
    add = spec["add_edges"]
    if add == "complete":
        add = [(i, j) for i in range(n) for j in range(i + 1, n)]
    return n, add, spec["delete_edges"]


@pytest.mark.parametrize("name", ["Graph1", "Graph2"])
def test_clique_reference_gtests(name):
    """reference test/test_maximum_clique.cpp:7-53 -- the only known answers the reference holds."""
    spec = json.load(open(os.path.join(GOLD, "clique_reference_tests.json")))[name]
    n, add, dele = _graph(spec)
    size, verts, underruns, steps = O.clique(n, add, dele)
    assert size == spec["expected_maximum_clique_size"]
    # the returned set really is a clique of the graph
    es = {tuple(sorted(e)) for e in add} - {tuple(sorted(e)) for e in dele}
    for a in range(len(verts)):
        for b in range(a + 1, len(verts)):
            assert tuple(sorted((int(verts[a]), int(verts[b])))) in es


def test_clique_is_valid_but_not_always_maximum():
    """The reference shares one colour vector across recursion levels (maximum_clique.cpp:287,315,320): a child
    reads C.back() at the parent's top, not the colour ColorSort just wrote for its own last vertex, so the
    bound is a heuristic and FindMaximumClique often misses the maximum (18 of 60 exact on these graphs).
    The restatement must return a valid clique, never larger than the brute-force maximum."""
    import itertools
    exact = total = 0
    for seed in range(60):
        n = 6 + seed % 7
        edges = synth.random_graph_edges(n, 0.3 + 0.1 * (seed % 5), 9000 + seed)
        if len(edges) == 0:
            continue
        size, verts, underruns, _ = O.clique(n, edges)
        es = {tuple(e) for e in edges.tolist()}
        assert len(set(verts.tolist())) == size
        assert all((min(a, b), max(a, b)) in es for a, b in itertools.combinations(verts.tolist(), 2))
        best = 1
        for r in range(2, n + 1):
            if any(all((min(a, b), max(a, b)) in es for a, b in itertools.combinations(c, 2))
                   for c in itertools.combinations(range(n), r)):
                best = r
            else:
                break
        assert 1 <= size <= best
        total += 1
        exact += size == best
    assert 0 < exact < total      # documents the reference's behaviour: neither always right nor always wrong


def test_rand_model_equals_libc_golden():
    gold = json.load(open(os.path.join(GOLD, "libc_rand_seed1.json")))["values"]
    r = O.rng_new(1)
    assert [O.rng_next(r) for _ in range(len(gold))] == gold


def test_rand_model_equals_libc_live():
    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 2, 12345):
        libc.srand(seed)
        r = O.rng_new(seed)
        for _ in range(20000):
            assert O.rng_next(r) == libc.rand()
    libc.srand(1)


def _numpy_knn(db, q, k):
    lut = np.array([bin(i).count("1") for i in range(256)], np.uint32)
    out = np.full((q.shape[0], k), np.iinfo(np.uint64).max, np.uint64)
    for i in range(q.shape[0]):
        d = lut[np.bitwise_xor(db, q[i])].sum(axis=1).astype(np.uint64)
        key = (d << np.uint64(32)) | np.arange(db.shape[0], dtype=np.uint64)
        key.sort()
        out[i, :min(k, len(key))] = key[:k]
    return out


def test_knn_oracle_against_numpy():
    desc, pts, off = synth.make_db_ragged([37, 1, 400, 0, 250], seed=7)
    q = synth.make_frame(desc, pts, off, 40, frame=3, visible_object=2)["q_desc"]
    desc[100] = desc[50]          # duplicated rows: equal distances must come out in row order
    desc[600] = desc[50]
    q[0] = desc[50]
    for k in (1, 2, 5, 8):
        assert np.array_equal(O.knn_keys(desc, q, k), _numpy_knn(desc, q, k))
    keys = O.knn_keys(desc, q, 5)
    assert [int(v & 0xFFFFFFFF) for v in keys[0, :3]] == [50, 100, 600] and int(keys[0, 0] >> 32) == 0


def test_match_radius_cut_and_gather():
    desc, pts, off = synth.make_db_ragged([300, 200, 100], seed=11)
    fr = synth.make_frame(desc, pts, off, 64, frame=1, visible_object=1, on_object=0.5)
    rc, row_ptr, m, xyz = O.match(desc, off, pts, fr["q_desc"], 5, 35)
    assert rc == 0 and row_ptr[-1] == len(m)
    keys = _numpy_knn(desc, fr["q_desc"], 5)
    for qi in range(64):
        mine = m[row_ptr[qi]:row_ptr[qi + 1]]
        want = [kk for kk in keys[qi] if (int(kk) >> 32) <= 35]
        # truncation at the first distance > radius (DescriptorMatcher.cpp:212-220) == prefix of the sorted list
        assert len(mine) == len(want)
        for a, kk in zip(mine, want):
            row = int(kk) & 0xFFFFFFFF
            obj = int(np.searchsorted(off, row, side="right") - 1)
            assert (a["imgIdx"], a["trainIdx"], a["queryIdx"]) == (obj, row - int(off[obj]), qi)
            assert a["distance"] == float(int(kk) >> 32)
    rows = off[m["imgIdx"]].astype(np.int64) + m["trainIdx"]
    assert np.array_equal(xyz, pts[rows])
    # radius 0: the reference skips matching and then indexes an empty vector (:237) -> error
    assert O.match(desc, off, pts, fr["q_desc"], 5, 0)[0] != 0


def test_spans_bbox_diagonal():
    desc, pts, off = synth.make_db_ragged([10, 5000], seed=3)
    sp = O.spans(pts, off)
    for o in range(2):
        p = pts[off[o]:off[o + 1]]
        e = (p.max(0) - p.min(0)).astype(np.float32)
        assert abs(sp[o] - np.sqrt((e * e).sum(dtype=np.float32))) < 1e-6
    assert abs(sp[1] - 0.269) < 2e-3       # 0.20 x 0.15 x 0.10 m box


def test_verify_recovers_known_pose():
    desc, pts, off = synth.make_db(3, per_object=800)
    fr = synth.make_frame(desc, pts, off, 400, frame=0, visible_object=1)
    rc, row_ptr, m, xyz = O.match(desc, off, pts, fr["q_desc"], 5, 35)
    sp = O.spans(pts, off)
    rng = O.rng_new(1)
    rc, poses, rounds = O.verify(fr["kp_xy"], fr["cloud"], row_ptr, m, xyz, sp, 8, 2500, 0.01, rng)
    assert rc == 0 and len(poses) == 1 and poses[0]["object"] == 1
    assert np.abs(poses[0]["R"] - synth.pose_R()).max() < 0.03      # 2 mm noise on a 0.2 m object
    assert np.abs(poses[0]["t"] - synth.POSE_T).max() < 0.01
    assert rounds[0].draws_after > rounds[0].draws_before


def test_train_observation_against_numpy():
    """validateKeyPoints + depthTo3dSparse + cameraToWorld (training.cpp:57-195, Trainer.cpp:166-178) on a hand case,
    checked with independent numpy arithmetic. PARITY UNPINNED w.r.t. the reference (no fixture for this path)."""
    H, W = 48, 64
    mask = np.zeros((H, W), np.uint8)
    mask[8:40, 10:50] = 255
    depth = np.full((H, W), 0.8, np.float32)
    depth[20, 30] = np.nan
    K = np.array([[100, 0, 31.5], [0, 100, 23.5], [0, 0, 1]], np.float32)
    R = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], np.float32)
    T = np.array([0.1, 0.2, 0.3], np.float32)
    kp = np.array([[30.2, 24.6],     # inside the eroded mask, valid depth -> kept
                   [30.0, 20.0],     # depth NaN at the pixel -> rescued by a neighbour within +-2 or dropped
                   [2.0, 2.0],       # outside the mask -> dropped
                   [11.0, 24.0],     # inside the mask but within 4 px of its border -> eroded away
                   [45.0, 12.0]], np.float32)
    desc = np.arange(5 * 32, dtype=np.uint8).reshape(5, 32)
    od, op, src = O.train_observation(kp, desc, mask, depth, K, R, T)
    assert src[0] == 0 and 2 not in src and 3 not in src
    assert np.array_equal(od, desc[src])
    er = O.train_erode4(mask)
    assert er[12, 14] == 255 and er[11, 14] == 0 and er[12, 13] == 0 and er.sum() // 255 == 24 * 32
    for row, s in zip(op, src):
        # the world point maps back through R, T and K to a pixel within the +-2 rescue window of the keypoint
        cam = R @ row + T
        u, v = cam[0] / cam[2] * 100 + 31.5, cam[1] / cam[2] * 100 + 23.5
        assert abs(cam[2] - 0.8) < 1e-5 and abs(u - kp[s, 0]) <= 2.5 and abs(v - kp[s, 1]) <= 2.5


def test_l2_oracle_against_numpy():
    """oracle/l2_oracle.c (the DEFINITION of the float-descriptor result, BASELINE configs[3]) against float64 numpy:
    same neighbours wherever float64 separates them by more than the f32 summation error."""
    from tod_amd import synth
    desc, pts, off = synth.make_sift_db(2, per_object=1500)
    q, truth = synth.make_sift_queries(desc, 40, frame=1)
    k = 4
    rc, row_ptr, m, xyz = O.l2_match(desc, off, pts, q, k, 1.0e9)
    assert rc == 0 and (np.diff(row_ptr.astype(np.int64)) == k).all()
    d2 = ((q[:, None, :].astype(np.float64) - desc[None, :, :].astype(np.float64)) ** 2).sum(-1)
    order = np.argsort(d2, axis=1, kind="stable")[:, :k]
    got = (off[m["imgIdx"]].astype(np.int64) + m["trainIdx"]).reshape(-1, k)
    gap = np.diff(np.sort(d2, axis=1)[:, :k + 1], axis=1).min(axis=1)
    clear = gap > 1.0                                               # f32 error of a 128-term sum at d2 ~ 1e5 is ~ 0.1
    assert clear.sum() > 30 and np.array_equal(got[clear], order[clear])
    assert np.allclose(m["distance"].reshape(-1, k), np.sqrt(np.take_along_axis(d2, got, axis=1)), rtol=1e-5)
    assert np.array_equal(xyz, pts[got.reshape(-1)])
    # radius truncation is strict: a cut exactly at a distance keeps it
    r0 = float(m["distance"][1])
    rc, row_ptr2, m2, _ = O.l2_match(desc, off, pts, q[:1], k, r0)
    assert row_ptr2[1] == 2 and np.array_equal(m2["trainIdx"], m["trainIdx"][:2])


def test_rescale_depth_known_answers():
    """rescale_depth (Trainer.cpp:62-81) on hand cases. The reference's cv::resize call runs BILINEAR (CV_INTER_NN lands
    in `fx`, :78); cv::resize maps [0, 1] -> [0, .25, .75, 1] when doubling. PARITY UNPINNED (third-party cv::resize)."""
    d = np.array([[0.0, 1.0], [2.0, 3.0]], np.float32)
    out = O.train_rescale_depth(d, 4, 4)
    row = np.array([0, .25, .75, 1], np.float32)
    assert np.array_equal(out[0], row) and np.array_equal(out[3], row + 2)
    assert np.array_equal(out[:, 0], np.array([0, .5, 1.5, 2], np.float32))
    nn = O.train_rescale_depth(d, 4, 4, nearest=True)
    assert np.array_equal(nn, np.repeat(np.repeat(d, 2, 0), 2, 1))
    # aspect mismatch: 2 x 4 depth on a 6 x 8 image fills int(2 * 2.0) = 4 rows, NaN below (:74-76)
    out = O.train_rescale_depth(np.ones((2, 4), np.float32), 6, 8)
    assert np.all(out[:4] == 1) and np.isnan(out[4:]).all()
    # uint16 millimetres, 0 = no measurement; equal size = conversion only (:68-71)
    d16 = np.array([[0, 1000], [1500, 65535]], np.uint16)
    out = O.train_rescale_depth(d16, 2, 2)
    assert np.isnan(out[0, 0]) and out[0, 1] == np.float32(1000) * np.float32(0.001) and out[1, 0] == np.float32(1500) * np.float32(0.001)
    # exact 2x shrink: the mean of each 2 x 2 block; a NaN poisons its block
    big = np.arange(16, dtype=np.float32).reshape(4, 4)
    big[3, 3] = np.nan
    out = O.train_rescale_depth(big, 2, 2)
    assert out[0, 0] == 2.5 and out[0, 1] == 4.5 and out[1, 0] == 10.5 and np.isnan(out[1, 1])
    # a depth whose scaled height exceeds the image: cv::Mat::rowRange throws in the reference
    assert O.train_rescale_depth(np.ones((8, 4), np.float32), 6, 8) is None


def test_ratio_test_definition_against_numpy():
    """orc_match_ratio (the definition of the ratio test that the reference leaves empty) against an independent numpy form"""
    from tod_amd import synth
    desc, pts, off = synth.make_db_ragged([400, 30, 250], seed=5)
    desc[500] = desc[7]
    fr = synth.make_frame(desc, pts, off, 80, frame=2, visible_object=0, flip_p=0.10)
    q = fr["q_desc"]; q[0] = desc[7]
    lut = np.array([bin(i).count("1") for i in range(256)], np.uint32)
    d = lut[q[:, None, :] ^ desc[None, :, :]].sum(-1).astype(np.int64)            # [Q, N]
    order = np.argsort(d * (1 << 20) + np.arange(d.shape[1])[None, :], axis=1)      # (distance, then row) ascending
    for k, radius, ratio in ((1, 40, 0.8), (3, 40, 0.8), (2, 255, 0.6), (2, 35, 1.0)):
        rc, row_ptr, m, xyz = O.match(desc, off, pts, q, k, radius, ratio)
        assert rc == 0
        for qi in range(len(q)):
            rows = order[qi, :max(k, 2)]
            d1, d2 = d[qi, rows[0]], d[qi, rows[1]]
            want = [] if not (np.float32(d1) < np.float32(ratio) * np.float32(d2)) else [int(r) for r in rows[:k] if d[qi, r] <= radius]
            want = want[:next((i for i, r in enumerate(rows[:k]) if d[qi, r] > radius), k)] if want else want
            got = (off[m["imgIdx"][row_ptr[qi]:row_ptr[qi + 1]]].astype(np.int64) + m["trainIdx"][row_ptr[qi]:row_ptr[qi + 1]]).tolist()
            assert got == want, (k, radius, ratio, qi)


def test_verify_2d_definition_recovers_known_poses_and_its_p3p_is_a_p3p():
    """oracle/pnp_oracle.c DEFINES the 2D-only branch (GuessGenerator.cpp:147-152 is a TODO in the reference: parity
    unpinned by construction). Checked here against what it claims to be: the poses of a two-object scene come out within
    the noise of the keypoints, every reported inlier reprojects within the threshold under the reported pose, the rand()
    stream advances by exactly one draw, and with noise-free keypoints the pose is exact to float precision."""
    K = np.array([[525.0, 0, 320.0], [0, 525.0, 240.0], [0, 0, 1]], np.float32)
    for noise, tol_R, tol_t in ((0.002, 0.03, 0.012), (0.0, 2e-4, 2e-4)):
        sc = synth.make_verify_scene(600, n_objects=5, per_object=300, visible=((1, 0.30), (3, 0.22)), matches_per_kp=3, seed=4, noise=noise)
        rng = O.rng_new(1)
        d0 = rng.draws
        rc, poses, (bh, bc) = O.verify_2d(sc["kp_xy"], K, sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], 12, 400, 3.0, rng)
        assert rc == 0 and rng.draws == d0 + 1
        assert sorted(p["object"] for p in poses) == [1, 3]
        for p in poses:
            R_true, t_true = sc["poses"][p["object"]]
            assert np.abs(p["R"] - R_true).max() < tol_R and np.abs(p["t"] - t_true).max() < tol_t, (noise, p["R"], R_true, p["t"], t_true)
            assert np.abs(p["R"] @ p["R"].T - np.eye(3)).max() < 1e-5 and np.linalg.det(p["R"]) > 0.999
            assert len(p["inliers"]) >= 12 and (np.diff(p["inliers"].astype(np.int64)) > 0).all()
            # every inlier keypoint has a match to this object whose model point reprojects within 3 px
            for q in p["inliers"]:
                ms = range(sc["row_ptr"][q], sc["row_ptr"][q + 1])
                errs = []
                for m in ms:
                    if sc["matches"][m]["imgIdx"] != p["object"]:
                        continue
                    Xc = p["R"].astype(np.float64) @ sc["matches_xyz"][m].astype(np.float64) + p["t"].astype(np.float64)
                    errs.append(np.hypot(525.0 * Xc[0] / Xc[2] + 320.0 - sc["kp_xy"][q, 0], 525.0 * Xc[1] / Xc[2] + 240.0 - sc["kp_xy"][q, 1]))
                assert errs and min(errs) < 3.0 + 1e-3
        assert bc[1] >= 100 and bc[3] >= 80 and bh[1] < 400 and bc[0] < 12 and bc[2] < 12


def test_lsh_definition_against_numpy():
    """oracle/lsh_oracle.c (the checker of todhip_set_lsh; FLANN's scheme, own key bits, parity unpinned) against a numpy restatement
    of the same definition, and its structural properties: every table's key bits are distinct, level = key_size admits every row
    (so the result is the exact k-NN), level 0 with one table admits exactly the rows with the query's key."""
    import ctypes as C
    rng = np.random.Generator(np.random.PCG64(11))
    db = rng.integers(0, 256, (3000, 32), dtype=np.uint8)
    q = db[rng.integers(0, 3000, 20)] ^ (np.packbits(rng.random((20, 256)) < 0.05, axis=1, bitorder="little"))
    bits_db = np.unpackbits(db, axis=1, bitorder="little"); bits_q = np.unpackbits(q, axis=1, bitorder="little")
    dist = (bits_db[None, :, :] != bits_q[:, None, :]).sum(2)
    for tables, ks, level in ((3, 10, 1), (1, 6, 0), (5, 12, 2)):
        cand = np.zeros((20, 3000), bool)
        for t in range(tables):
            pos = np.zeros(ks, np.uint8)
            O.lib().orc_lsh_key_bits(C.c_uint32(t), C.c_uint32(ks), pos.ctypes.data_as(C.POINTER(C.c_uint8)))
            assert len(set(pos.tolist())) == ks
            cand |= (bits_db[None, :, pos] != bits_q[:, None, pos]).sum(2) <= level
        keys, n_cand = O.lsh_knn_keys(db, q, 4, tables, ks, level)
        assert np.array_equal(n_cand, cand.sum(1))
        for i in range(20):
            rows = np.flatnonzero(cand[i])
            order = rows[np.lexsort((rows, dist[i, rows]))][:4]
            want = [(int(dist[i, r]) << 32) | int(r) for r in order] + [2 ** 64 - 1] * (4 - len(order))
            assert keys[i].tolist() == want
    keys, n_cand = O.lsh_knn_keys(db, q, 3, 2, 8, 8)
    assert (n_cand == 3000).all() and np.array_equal(keys, O.knn_keys(db, q, 3))
