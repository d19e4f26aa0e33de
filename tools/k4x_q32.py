"""The memory-bound regime alone, for the profiler: 16 queries per pass over a 40M-row DB (1.28 GB), kernel hamming_topk_mfma_q32,
exactly bench.py's hbm_regime launch; a dozen launches. Prints the live launch time (HIP events on the context's stream)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi
rows, Q, k, radius = 40_000_000, 16, 2, 35
rng = np.random.Generator(np.random.PCG64(77))
desc = rng.integers(0, 256, size=(rows, 32), dtype=np.uint8)
pts = np.zeros((rows, 3), np.float32)
off = (np.arange(rows // 5000 + 1, dtype=np.uint64) * 5000).astype(np.uint32)
ctx = capi.Context(0); ctx.db_load(desc, pts, off); ctx.set_matcher_engine("mfma")
qrows = rng.choice(rows, Q, replace=False)
q = desc[qrows] ^ np.packbits(rng.random((Q, 256)) < 0.08, axis=1, bitorder="little")
del desc, pts
d_q = torch.from_numpy(np.ascontiguousarray(q)).cuda()
cnt = torch.zeros(Q, dtype=torch.int32, device="cuda"); mm = torch.zeros((Q * k, 4), dtype=torch.int32, device="cuda"); xx = torch.zeros((Q * k, 3), device="cuda")
call = lambda: ctx.match_device(d_q.data_ptr(), Q, k, radius, cnt.data_ptr(), mm.data_ptr(), xx.data_ptr())
for _ in range(2): call()
ctx.synchronize(); ctx.set_kernel_timing(True); c0 = ctx.counters()
for _ in range(12): call()
ctx.synchronize(); c1 = ctx.counters()
ms = (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / (c1.n_match_kernel_launches - c0.n_match_kernel_launches)
print("q32: %d queries x %d rows: %.4f ms per launch live = %.0f GB/s of %d algorithmic bytes" % (Q, rows, ms, rows * 32 / (ms * 1e-3) / 1e9, rows * 32 + Q * (32 + k * 8)))
