"""Multi-GPU step of the detection hot path.

The descriptor rows of the object DB are cut into object-aligned shards, one per rank; the frames of a
batch are dealt one per rank. One step =

  1. all-gather of the ranks' frame descriptors          (world x Q x 32 B, so every rank holds every frame)
  2. every rank matches ALL frames against ITS shard      -> per-shard top-k keys (distance << 32 | global row)
  3. exchange of those candidate keys: all-gather         (the "final RCCL all-gather of per-shard match
     candidates" of BASELINE.json's north star) or, with the `all_to_all` callable, an all-to-all -- a rank only
     merges ITS OWN frames, so it only needs 1/world of what an all-gather delivers (xGMI is point to point: the
     traffic, not the latency, is what a ring all-gather of F*Q*k*8 B per rank pays for)
  4. every rank merges the world candidate lists of ITS frame with the order (distance asc, global row asc)
     -- identical to the 1-GPU result -- and verifies that frame.

Per-rank work is constant as ranks are added (Q*N distances, one frame verified), the DB is fixed, and
frames/s grows with the rank count. The collectives go through torch.distributed (backend "nccl" == RCCL
over xGMI on ROCm, "gloo" in the CPU tests); the compute is injected as callables so that the choreography
is testable without a GPU.
"""
import numpy as np


def shard_bounds(obj_off, rank, world):
    """Object-aligned contiguous shard of `rank`: objects whose first row falls into the rank's 1/world slice
    of the rows. Mirrors shard_bounds() in csrc/capi.hip. Returns (obj_lo, obj_hi, row_lo, row_hi)."""
    obj_off = np.asarray(obj_off, np.int64)
    n_obj = len(obj_off) - 1
    total = int(obj_off[-1])

    def owner(o):
        if total == 0:
            return 0
        return min(int(obj_off[o]) * world // total, world - 1)

    lo = 0
    while lo < n_obj and owner(lo) < rank:
        lo += 1
    hi = lo
    while hi < n_obj and owner(hi) == rank:
        hi += 1
    return lo, hi, int(obj_off[lo]), int(obj_off[hi])


def sharded_step(dist, world, rank, my_q, match_shard, merge, verify, alloc, all_gather, all_to_all=None):
    """One step. my_q: this rank's frame descriptors [Q, B]. Callables:
         alloc(shape, dtype_name) -> tensor on the compute device
         all_gather(out, inp)     -> dist.all_gather_into_tensor on flat views
         all_to_all(out, inp)     -> dist.all_to_all_single on flat views (optional: step 3 as an all-to-all)
         match_shard(q_all)       -> keys [world*Q, k] int64 of this rank's shard for all frames
         merge(keys_mine)         -> merged matches of this rank's frame from keys [world, Q, k]
         verify(matches)          -> poses of this rank's frame
    """
    Q = my_q.shape[0]
    if world == 1:
        keys = match_shard(my_q)
        return verify(merge(keys.reshape(1, Q, -1)))
    q_all = alloc((world,) + tuple(my_q.shape), "uint8")
    all_gather(q_all, my_q)
    keys = match_shard(q_all.reshape(world * Q, -1))                 # [world*Q, k]
    k = keys.shape[-1]
    if all_to_all is not None:
        mine = alloc((world, Q, k), "int64")                         # chunk j <- shard j's keys of MY frame
        all_to_all(mine, keys)                                       # keys is [frame owner][Q][k]: chunk j -> rank j
        return verify(merge(mine))
    keys_all = alloc((world, world, Q, k), "int64")                  # [shard][frame][Q][k]
    all_gather(keys_all, keys)
    mine = keys_all[:, rank].contiguous()                            # [shard][Q][k] of my frame
    return verify(merge(mine))
