"""Multi-GPU step of the detection hot path.

The descriptor rows of the object DB are cut into object-aligned shards, one per rank; every rank owns B frames of a
step. One step =

  1. all-gather of the ranks' frame descriptors          (world x B x Q x 32 B, so every rank holds every frame)
  2. every rank matches ALL frames against ITS shard      -> per-shard top-k keys (distance << 32 | global row)
  3. exchange of those candidate keys: an all-to-all by default -- a rank only merges ITS OWN frames, so it needs 1/world
     of what an all-gather delivers, and xGMI is point to point: the bytes, not the latency, are what a ring pays for --
     or the literal "final RCCL all-gather of per-shard match candidates" of BASELINE.json's north star (`all_gather`)
  4. every rank merges the world candidate lists of ITS frames with the order (distance asc, global row asc)
     -- identical to the 1-GPU result -- and verifies those frames (frame-parallel, no collective).

Per-rank work is constant as ranks are added (B x Q x N distances, B frames verified), the DB is fixed, and frames/s
grows with the rank count.

`ShardedMatcher` is the ONE implementation of steps 1-4's choreography: bench.py runs it on the GPU (GpuOps: torch
streams/events, RCCL through torch.distributed, libtodhip for the compute) and tests/test_sharded_cpu.py runs the same
class over gloo with a CPU restatement as the compute. With `overlap` the collectives and the merge have a stream of
their own, double buffered, so that the DB pass of step i + 1 follows that of step i without a gap:

  comm stream, identical on every rank:  gather(0), gather(1), exchange(0), merge(0), gather(2), exchange(1), merge(1), ...
  cross-stream edges (events):           gathered(i) -> match(i);   matched(i) -> exchange(i);
                                         exchanged(i - 2) -> match(i)   (match(i) overwrites the keys buffer exchange(i - 2) read)
  by stream order alone:                 gather(i + 2) overwrites q_all[i % 2] after exchange(i), which waited for match(i);
                                         exchange(i + 2) overwrites mine[i % 2] after merge(i)
"""
import contextlib

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


class ShardedMatcher:
    """Steps 1-4 above for batches of B frames per rank. `ops` supplies the backend:

         ops.compute, ops.comm          stream handles (equal when the backend has one stream)
         ops.alloc(shape, dtype_name)   -> tensor on the compute device
         ops.use(stream)                -> context manager: collectives issued inside go to `stream`
         ops.record(stream) -> event;   ops.wait(stream, event)
         ops.all_gather(out, inp);      ops.all_to_all(out, inp)        (flat views, equal splits)
         ops.match_shard(q_all, n, keys_out)          this rank's shard against n queries, on ops.compute
         ops.merge(keys_mine, n_shards, n, out, stream)   merge + radius cut + gather into `out`, on `stream`

       begin(n_steps, q_of): q_of(i) -> (this rank's descriptors of step i as a [B, Q, desc_bytes] tensor, event or None
       after which they are complete). step(i, out) issues step i and returns the stream on which `out` becomes complete.
    """

    def __init__(self, ops, world, rank, frames_per_rank, nq, k, desc_bytes=32, exchange="all_to_all", overlap=True):
        assert exchange in ("all_to_all", "all_gather")
        self.ops, self.world, self.rank = ops, world, rank
        self.B, self.nq, self.k, self.exchange = frames_per_rank, nq, k, exchange
        self.overlap = overlap and ops.comm is not ops.compute
        n = frames_per_rank * nq
        depth = 2 if self.overlap else 1
        self.q_all = [ops.alloc((world, frames_per_rank, nq, desc_bytes), "uint8") for _ in range(depth)]
        self.keys = [ops.alloc((world * n, k), "int64") for _ in range(depth)]          # [frame owner][B*Q][k]
        self.mine = [ops.alloc((world, n, k), "int64") for _ in range(depth)]           # [shard][B*Q][k] of MY frames
        self.keys_all = ([ops.alloc((world, world, n, k), "int64") for _ in range(depth)]  # [shard][frame owner][B*Q][k]
                         if exchange == "all_gather" else None)
        self.n_steps, self.q_of = 0, None
        self.ev_gathered, self.ev_exchanged = {}, {}

    def begin(self, n_steps, q_of):
        self.n_steps, self.q_of = n_steps, q_of
        self.ev_gathered, self.ev_exchanged = {}, {}
        if self.overlap and n_steps > 0:
            self._gather(0)

    def _gather(self, i):
        ops = self.ops
        q, ready = self.q_of(i)
        with ops.use(ops.comm):
            if ready is not None:
                ops.wait(ops.comm, ready)
            ops.all_gather(self.q_all[i % 2], q)
            self.ev_gathered[i] = ops.record(ops.comm)

    def _exchange(self, slot):
        """keys[slot] -> mine[slot]: every shard's candidates for this rank's frames."""
        ops = self.ops
        if self.exchange == "all_to_all":
            ops.all_to_all(self.mine[slot], self.keys[slot])            # chunk j of keys goes to rank j
        else:
            ops.all_gather(self.keys_all[slot], self.keys[slot])
            self.mine[slot].copy_(self.keys_all[slot][:, self.rank])

    def step(self, i, out):
        ops, n = self.ops, self.B * self.nq
        if not self.overlap:
            # program order on the one stream: gather -> match -> exchange -> merge
            q, ready = self.q_of(i)
            with ops.use(ops.compute):
                if ready is not None:
                    ops.wait(ops.compute, ready)
                ops.all_gather(self.q_all[0], q)
                ops.match_shard(self.q_all[0], self.world * n, self.keys[0])
                self._exchange(0)
                ops.merge(self.mine[0], self.world, n, out, ops.compute)
            return ops.compute
        s = i % 2
        if i + 1 < self.n_steps:
            self._gather(i + 1)
        ops.wait(ops.compute, self.ev_gathered.pop(i))
        if i - 2 in self.ev_exchanged:
            ops.wait(ops.compute, self.ev_exchanged.pop(i - 2))
        ops.match_shard(self.q_all[s], self.world * n, self.keys[s])
        matched = ops.record(ops.compute)
        with ops.use(ops.comm):
            ops.wait(ops.comm, matched)
            self._exchange(s)
            self.ev_exchanged[i] = ops.record(ops.comm)
            ops.merge(self.mine[s], self.world, n, out, ops.comm)
        return ops.comm


class GpuOps:
    """ShardedMatcher backend on the GPU: torch streams/events, torch.distributed collectives (backend "nccl" is RCCL
    over xGMI; "gloo" stages through the host -- rehearsals on a one-GPU box only), libtodhip for the compute. `out` is a
    dict of torch tensors: counts [n] i32, matches [n*k, 4] i32, xyz [n*k, 3] f32 (todhip_match_device's outputs)."""

    def __init__(self, ctx, compute_stream, comm_stream, backend, k, radius):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ctx, self.compute, self.comm = ctx, compute_stream, comm_stream
        self.backend, self.k, self.radius = backend, k, radius
        self.world = dist.get_world_size()

    def alloc(self, shape, dtype_name):
        return self.torch.empty(shape, dtype=getattr(self.torch, dtype_name), device="cuda")

    def use(self, stream):
        return self.torch.cuda.stream(stream)

    def record(self, stream):
        ev = self.torch.cuda.Event()
        ev.record(stream)
        return ev

    def wait(self, stream, event):
        stream.wait_event(event)

    def all_gather(self, out, inp):
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(out.view(-1), inp.contiguous().view(-1))
        else:
            parts = [self.torch.empty(inp.numel(), dtype=inp.dtype) for _ in range(self.world)]
            self.torch.cuda.current_stream().synchronize()
            self.dist.all_gather(parts, inp.contiguous().view(-1).cpu())
            out.view(-1).copy_(self.torch.cat(parts).to(out.device))

    def all_to_all(self, out, inp):
        if self.backend == "nccl":
            self.dist.all_to_all_single(out.view(-1), inp.contiguous().view(-1))
        else:
            o = self.torch.empty(out.numel(), dtype=out.dtype)
            self.torch.cuda.current_stream().synchronize()
            self.dist.all_to_all_single(o, inp.contiguous().view(-1).cpu())
            out.view(-1).copy_(o.to(out.device))

    def match_shard(self, q_all, n, keys_out):
        self.ctx.match_shard_device(q_all.data_ptr(), n, self.k, self.radius, keys_out.data_ptr())

    def merge(self, keys_mine, n_shards, n, out, stream):
        args = (keys_mine.data_ptr(), n_shards, n, self.k, self.radius, out["counts"].data_ptr(), out["matches"].data_ptr(),
                out["xyz"].data_ptr())
        if stream is self.compute:
            self.ctx.merge_shards_device(*args)
        else:       # the merge only reads the context's immutable tables: it may run beside the next DB pass
            self.ctx.merge_shards_device_on(stream.cuda_stream, *args)


class HostOps:
    """ShardedMatcher backend without a GPU (tests/test_sharded_cpu.py): CPU tensors, gloo collectives, one synchronous
    "stream" per name so that the overlapped choreography (buffer rotation, issue order) runs exactly as on the GPU; the
    two compute calls are injected."""

    def __init__(self, dist, match_shard, merge, two_streams=True):
        import torch
        self.torch, self.dist = torch, dist
        self.compute = "compute"
        self.comm = "comm" if two_streams else self.compute
        self._match_shard, self._merge = match_shard, merge
        self.log = []                                   # (what, stream) in issue order: the tests check it is rank independent

    def alloc(self, shape, dtype_name):
        return self.torch.zeros(shape, dtype=getattr(self.torch, dtype_name))

    def use(self, stream):
        return contextlib.nullcontext()

    def record(self, stream):
        return None

    def wait(self, stream, event):
        pass

    def all_gather(self, out, inp):
        self.log.append("all_gather%d" % inp.numel())
        self.dist.all_gather_into_tensor(out.view(-1), inp.contiguous().view(-1))

    def all_to_all(self, out, inp):
        self.log.append("all_to_all%d" % inp.numel())
        self.dist.all_to_all_single(out.view(-1), inp.contiguous().view(-1))

    def match_shard(self, q_all, n, keys_out):
        keys_out.copy_(self._match_shard(q_all.reshape(n, -1)))

    def merge(self, keys_mine, n_shards, n, out, stream):
        out["result"] = self._merge(keys_mine.clone())
