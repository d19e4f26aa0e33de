"""Child process of tests/test_sharded_gpu.py (started before anything touches the GPU): torch.distributed on the "nccl"
backend (= RCCL on ROCm) with ONE rank, tod_amd/sharded.py::ShardedMatcher over GpuOps -- the class and the backend bench.py runs
on several GPUs -- for several consecutive steps, overlapped and serial, both exchanges; every step's merged matches must equal
todhip_match_device on the whole DB. `stress`: torch.cuda._sleep on the streams in front of the collectives and of the DB pass, so
that a consumer that does not wait for its producer's event reads stale buffers and the comparison fails."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29641")
os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

from tod_amd import capi, sharded, synth

K, RADIUS, NQ, B, STEPS = 2, 45, 300, 3, 6
stress = len(sys.argv) > 1 and sys.argv[1] in ("stress", "broken")
broken = len(sys.argv) > 1 and sys.argv[1] == "broken"     # the detector's own check: no cross-stream edge at all must be noticed

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

desc, pts, off = synth.make_db_ragged([2500, 40, 0, 1900, 3500, 5, 2610], seed=321)
compute, comm = torch.cuda.Stream(), torch.cuda.Stream()
ctx = capi.Context(0, compute.cuda_stream)                  # this rank's shard: with one rank, every row
ctx.db_load(desc, pts, off, 0, 1)
ref = capi.Context(0)                                       # the unsharded matcher, on a stream of its own
ref.db_load(desc, pts, off)
frames = {(i, b): synth.make_frame(desc, pts, off, NQ, frame=100 * i + b, visible_object=(0, 3, 4, 6)[(i + b) % 4])
          for i in range(STEPS) for b in range(B)}
q_dev = [torch.from_numpy(np.stack([frames[(i, b)]["q_desc"] for b in range(B)])).cuda() for i in range(STEPS)]
torch.cuda.synchronize()
n = B * NQ


class SleepyOps(sharded.GpuOps):
    """The producers are late: whoever reads q_all / keys / mine without its event edge reads the previous step's data."""
    SPIN = 3_000_000                                         # ~1.5 ms of s_sleep on the stream

    def all_gather(self, out, inp):
        torch.cuda._sleep(self.SPIN)                         # (inside ops.use(stream): the current stream is the collective's)
        super().all_gather(out, inp)

    def all_to_all(self, out, inp):
        torch.cuda._sleep(self.SPIN)
        super().all_to_all(out, inp)

    def match_shard(self, q_all, nn, keys_out):
        with torch.cuda.stream(self.compute):
            torch.cuda._sleep(self.SPIN)
        super().match_shard(q_all, nn, keys_out)

    def wait(self, stream, event):
        if not broken:
            super().wait(stream, event)


def new_out():
    return dict(counts=torch.zeros(n, dtype=torch.int32, device="cuda"), matches=torch.zeros((n * K, 4), dtype=torch.int32, device="cuda"),
                xyz=torch.zeros((n * K, 3), dtype=torch.float32, device="cuda"))


# the reference result of every step, computed before anything else runs
want = []
for i in range(STEPS):
    o = new_out()
    ref.match_device(q_dev[i].data_ptr(), n, K, RADIUS, o["counts"].data_ptr(), o["matches"].data_ptr(), o["xyz"].data_ptr())
    ref.synchronize()
    want.append({key: v.cpu().numpy().copy() for key, v in o.items()})
assert sum(int(w["counts"].sum()) for w in want) > STEPS * B * 50

n_checked = 0
for exchange in ("all_to_all", "all_gather"):
    for overlap in (True, False):
        Ops = SleepyOps if stress else sharded.GpuOps
        ops = Ops(ctx, compute, comm, "nccl", K, RADIUS)
        sm = sharded.ShardedMatcher(ops, 1, 0, B, NQ, K, exchange=exchange, overlap=overlap)
        assert sm.overlap == overlap
        outs = [new_out() for _ in range(STEPS)]
        sm.begin(STEPS, lambda i: (q_dev[i], None))
        done = []
        for i in range(STEPS):
            s = sm.step(i, outs[i])
            assert s is (comm if overlap else compute)
            ev = torch.cuda.Event()
            ev.record(s)
            done.append(ev)
        for i in range(STEPS):
            done[i].synchronize()
            got = {key: v.cpu().numpy() for key, v in outs[i].items()}
            assert np.array_equal(got["counts"], want[i]["counts"]), (exchange, overlap, i, "counts")
            for q in range(n):                               # the fixed-stride slots beyond counts[q] are unspecified
                c = int(got["counts"][q])
                assert np.array_equal(got["matches"][q * K:q * K + c], want[i]["matches"][q * K:q * K + c]), (exchange, overlap, i, q)
                assert np.array_equal(got["xyz"][q * K:q * K + c], want[i]["xyz"][q * K:q * K + c]), (exchange, overlap, i, q)
            n_checked += 1
        torch.cuda.synchronize()
dist.destroy_process_group()
ctx.close(); ref.close()
print("ok: %d steps checked (%s)" % (n_checked, "stress" if stress else "plain"))
