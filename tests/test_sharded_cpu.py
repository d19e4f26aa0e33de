"""The multi-rank step (tod_amd/sharded.py) on CPU: world_size 2 over gloo, with the CPU oracle standing in for
the per-shard matcher and the verifier. Checks the collective choreography, the object-aligned sharding and that
the merged result of every rank's frame equals the unsharded result."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

K, RADIUS, NQ = 3, 60, 120


def _merge_numpy(keys_mine, off, pts, radius, k):
    """[shard][Q][k] keys -> CSR matches with the order (distance asc, global row asc), radius cut, gather."""
    S, Q, _ = keys_mine.shape
    row_ptr = np.zeros(Q + 1, np.uint32)
    out_m, out_xyz = [], []
    for q in range(Q):
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


def _worker(rank, world, port, ret, exchange):
    import oracle_lib as O
    from tod_amd import capi, sharded, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        desc, pts, off = synth.make_db_ragged([700, 40, 0, 900, 350, 5, 610], seed=123)
        fr = synth.make_frame(desc, pts, off, NQ, frame=10 + rank, visible_object=3 if rank == 0 else 0)
        obj_lo, obj_hi, row_lo, row_hi = sharded.shard_bounds(off, rank, world)
        spans = O.spans(pts, off)

        def alloc(shape, dtype_name):
            return torch.empty(shape, dtype=getattr(torch, dtype_name))

        def all_gather(out, inp):
            dist.all_gather_into_tensor(out.view(-1), inp.contiguous().view(-1))

        def all_to_all(out, inp):
            dist.all_to_all_single(out.view(-1), inp.contiguous().view(-1))

        def match_shard(q_all):
            keys = O.knn_keys(desc[row_lo:row_hi], q_all.numpy(), K) if row_hi > row_lo else \
                np.full((q_all.shape[0], K), np.iinfo(np.uint64).max, np.uint64)
            real = keys != np.uint64(0xFFFFFFFFFFFFFFFF)
            keys[real] += np.uint64(row_lo)
            return torch.from_numpy(keys.view(np.int64))

        def merge(keys_mine):
            return _merge_numpy(keys_mine.numpy().view(np.uint64), off, pts, RADIUS, K)

        def verify(matches):
            row_ptr, m, xyz = matches
            marr = np.array(m, capi.DMATCH_DTYPE) if m else np.zeros(0, capi.DMATCH_DTYPE)
            rng = O.rng_new(1)
            rc, poses, _ = O.verify(fr["kp_xy"], fr["cloud"], row_ptr, marr, xyz, spans, 8, 300, 0.01, rng)
            return row_ptr, marr, xyz, poses, rng.draws

        row_ptr, marr, xyz, poses, draws = sharded.sharded_step(dist, world, rank, torch.from_numpy(fr["q_desc"]),
                                                                match_shard, merge, verify, alloc, all_gather,
                                                                all_to_all if exchange == "all_to_all" else None)
        # unsharded reference on this rank's frame
        rc, o_rp, o_m, o_xyz = O.match(desc, off, pts, fr["q_desc"], K, RADIUS)
        rng = O.rng_new(1)
        rc, o_poses, _ = O.verify(fr["kp_xy"], fr["cloud"], o_rp, o_m, o_xyz, spans, 8, 300, 0.01, rng)
        ok = np.array_equal(row_ptr, o_rp) and np.array_equal(xyz, o_xyz) and draws == rng.draws
        for f in ("queryIdx", "trainIdx", "imgIdx", "distance"):
            ok = ok and np.array_equal(marr[f], o_m[f])
        ok = ok and len(poses) == len(o_poses) and all(
            a["object"] == b["object"] and np.array_equal(a["inliers"], b["inliers"]) and np.array_equal(a["R"], b["R"])
            for a, b in zip(poses, o_poses))
        ret[rank] = (bool(ok), len(marr), len(poses), (row_lo, row_hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange", [(2, "all_gather"), (3, "all_gather"), (2, "all_to_all"), (3, "all_to_all")])
def test_sharded_step_equals_unsharded(world, exchange):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29511 + world + (10 if exchange == "all_to_all" else 0)
    mp.spawn(_worker, args=(world, port, ret, exchange), nprocs=world, join=True)
    assert len(ret) == world
    rows = sorted(v[3] for v in ret.values())
    assert rows[0][0] == 0 and all(a[1] == b[0] for a, b in zip(rows, rows[1:])) and rows[-1][1] == 2605
    for r in range(world):
        ok, n_matches, n_poses, _ = ret[r]
        assert ok, "rank %d differs from the unsharded result" % r
        assert n_matches > 20
    assert ret[0][2] == 1      # rank 0's frame shows object 3 with enough matches for a pose


def test_shard_bounds_are_object_aligned_partitions():
    from tod_amd import sharded
    off = np.concatenate([[0], np.cumsum([900, 50, 0, 1200, 700, 5, 333, 2000, 41, 800])])
    for world in (1, 2, 3, 8, 16):
        spans = [sharded.shard_bounds(off, r, world) for r in range(world)]
        assert spans[0][2] == 0 and spans[-1][3] == off[-1]
        for a, b in zip(spans, spans[1:]):
            assert a[1] == b[0] and a[3] == b[2]
