"""The multi-rank step (tod_amd/sharded.py::ShardedMatcher -- the class bench.py runs on the GPU) on CPU: world_size 2 and
3 over gloo, with the CPU oracle standing in for the per-shard matcher, the merge and the verifier. Several consecutive
steps of B frames per rank, so the double-buffered gather / key / merge buffers of the overlapped choreography are reused;
both exchanges (all-to-all, all-gather); the serial single-stream form as well. Every step's merged result of every rank's
frames must equal the unsharded result, and all ranks must issue their collectives in the same order."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

K, RADIUS, NQ, B, STEPS = 3, 60, 96, 2, 5


def _merge_numpy(keys_mine, off, pts, radius, k):
    """[shard][n][k] keys -> per query: CSR matches with the order (distance asc, global row asc), radius cut, gather."""
    S, n, _ = keys_mine.shape
    row_ptr = np.zeros(n + 1, np.uint32)
    out_m, out_xyz = [], []
    for q in range(n):
        cand = np.sort(keys_mine[:, q, :].reshape(-1).astype(np.uint64))
        cand = cand[cand != np.uint64(0xFFFFFFFFFFFFFFFF)][:k]
        for key in cand:
            d, row = int(key) >> 32, int(key) & 0xFFFFFFFF
            if float(d) > float(radius):
                break
            obj = int(np.searchsorted(off, row, side="right") - 1)
            out_m.append((q, row - int(off[obj]), obj, float(d)))
            out_xyz.append(pts[row])
        row_ptr[q + 1] = len(out_m)
    return row_ptr, out_m, np.array(out_xyz, np.float32).reshape(-1, 3)


def _worker(rank, world, port, ret, exchange, overlap):
    import oracle_lib as O
    from tod_amd import capi, sharded, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        desc, pts, off = synth.make_db_ragged([700, 40, 0, 900, 350, 5, 610], seed=123)
        obj_lo, obj_hi, row_lo, row_hi = sharded.shard_bounds(off, rank, world)
        spans = O.spans(pts, off)
        # frame (step, rank, b): every step shows other frames, so a stale buffer cannot go unnoticed
        frames = {(i, b): synth.make_frame(desc, pts, off, NQ, frame=100 * i + 10 * rank + b,
                                           visible_object=(3, 0, 4, 6)[(i + rank + b) % 4]) for i in range(STEPS) for b in range(B)}

        def match_shard(q_all):
            keys = O.knn_keys(desc[row_lo:row_hi], q_all.numpy(), K) if row_hi > row_lo else \
                np.full((q_all.shape[0], K), np.iinfo(np.uint64).max, np.uint64)
            real = keys != np.uint64(0xFFFFFFFFFFFFFFFF)
            keys[real] += np.uint64(row_lo)
            return torch.from_numpy(keys.view(np.int64))

        def merge(keys_mine):
            return _merge_numpy(keys_mine.numpy().view(np.uint64), off, pts, RADIUS, K)

        ops = sharded.HostOps(dist, match_shard, merge, two_streams=overlap)
        sm = sharded.ShardedMatcher(ops, world, rank, B, NQ, K, exchange=exchange, overlap=overlap)
        assert sm.overlap == overlap

        def q_of(i):
            return torch.from_numpy(np.stack([frames[(i, b)]["q_desc"] for b in range(B)])), None

        sm.begin(STEPS, q_of)
        ok, n_matches, n_poses = True, 0, 0
        for i in range(STEPS):
            out = {}
            sm.step(i, out)
            row_ptr, m, xyz = out["result"]                          # B * NQ queries: frame b owns [b * NQ, (b + 1) * NQ)
            marr = np.array(m, capi.DMATCH_DTYPE) if m else np.zeros(0, capi.DMATCH_DTYPE)
            for b in range(B):
                fr = frames[(i, b)]
                rc, o_rp, o_m, o_xyz = O.match(desc, off, pts, fr["q_desc"], K, RADIUS)
                lo, hi = int(row_ptr[b * NQ]), int(row_ptr[(b + 1) * NQ])
                mine = marr[lo:hi].copy()
                mine["queryIdx"] -= b * NQ
                ok = ok and np.array_equal(row_ptr[b * NQ:(b + 1) * NQ + 1] - row_ptr[b * NQ], o_rp)
                ok = ok and np.array_equal(xyz[lo:hi], o_xyz)
                for f in ("queryIdx", "trainIdx", "imgIdx", "distance"):
                    ok = ok and np.array_equal(mine[f], o_m[f])
                n_matches += hi - lo
                if i == STEPS - 1:                                   # the verifier on the merged matches == on the unsharded ones
                    rng, rng_o = O.rng_new(1), O.rng_new(1)
                    rc, poses, _ = O.verify(fr["kp_xy"], fr["cloud"], (row_ptr[b * NQ:(b + 1) * NQ + 1] - row_ptr[b * NQ]).astype(np.uint32),
                                            mine, xyz[lo:hi], spans, 8, 300, 0.01, rng)
                    rc, o_poses, _ = O.verify(fr["kp_xy"], fr["cloud"], o_rp, o_m, o_xyz, spans, 8, 300, 0.01, rng_o)
                    ok = ok and rng.draws == rng_o.draws and len(poses) == len(o_poses) and all(
                        a["object"] == c["object"] and np.array_equal(a["inliers"], c["inliers"]) and np.array_equal(a["R"], c["R"])
                        for a, c in zip(poses, o_poses))
                    n_poses += len(poses)
        ret[rank] = (bool(ok), n_matches, n_poses, (row_lo, row_hi), list(ops.log))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange,overlap", [(2, "all_to_all", True), (3, "all_to_all", True), (2, "all_gather", True),
                                                    (3, "all_gather", True), (2, "all_to_all", False), (3, "all_gather", False)])
def test_sharded_matcher_steps_equal_unsharded(world, exchange, overlap):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29511 + world + (10 if exchange == "all_to_all" else 0) + (20 if overlap else 0)
    mp.spawn(_worker, args=(world, port, ret, exchange, overlap), nprocs=world, join=True)
    assert len(ret) == world
    rows = sorted(v[3] for v in ret.values())
    assert rows[0][0] == 0 and all(a[1] == b[0] for a, b in zip(rows, rows[1:])) and rows[-1][1] == 2605
    for r in range(world):
        ok, n_matches, n_poses, _, log = ret[r]
        assert ok, "rank %d differs from the unsharded result" % r
        assert n_matches > 20 * STEPS
        assert log == ret[0][4], "rank %d issued its collectives in another order than rank 0" % r
    assert sum(v[2] for v in ret.values()) >= 1          # at least one frame of the last step yields a pose
    # the overlapped form runs one step ahead with the descriptor gather: gather(0), gather(1), exchange(0), gather(2), ...
    log = ret[0][4]
    gathers = [i for i, w in enumerate(log) if w == "all_gather%d" % (B * NQ * 32)]
    assert len(gathers) == STEPS
    if overlap:
        assert gathers[:2] == [0, 1]
    else:
        assert gathers[0] == 0 and gathers[1] > 1


def test_shard_bounds_are_object_aligned_partitions():
    from tod_amd import sharded
    off = np.concatenate([[0], np.cumsum([900, 50, 0, 1200, 700, 5, 333, 2000, 41, 800])])
    for world in (1, 2, 3, 8, 16):
        spans = [sharded.shard_bounds(off, r, world) for r in range(world)]
        assert spans[0][2] == 0 and spans[-1][3] == off[-1]
        for a, b in zip(spans, spans[1:]):
            assert a[1] == b[0] and a[3] == b[2]
