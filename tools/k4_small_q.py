"""The memory-bound regime of the exact Hamming search (BASELINE.json: "achieved HBM GB/s on BF-matcher"): a few queries per
pass over a DB far larger than the 256 MiB Infinity Cache (40M rows = 1.28 GB), so every row comes from HBM and is used by
only Q queries. Reports the DB pass in GB/s against the HBM roof (8 TB/s spec, and the device-to-device copy rate measured
here), for both engines; results of the two engines must be identical, and 4 queries are checked against the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
import numpy as np, torch
from tod_amd import capi, synth
N_OBJ = int(os.environ.get("OBJECTS", "8000"))                      # x 5000 rows
rng = np.random.Generator(np.random.PCG64(77))
n = N_OBJ * 5000
t = time.time()
desc = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
pts = np.zeros((n, 3), np.float32); off = (np.arange(N_OBJ + 1, dtype=np.uint64) * 5000).astype(np.uint32)
print("DB: %d rows = %.2f GB (generated in %.1f s)" % (n, n * 32 / 1e9, time.time() - t), flush=True)
ctx = capi.Context(0); ctx.db_load(desc, pts, off)
# copy-rate reference: read + write of the same number of bytes
a = torch.empty(n * 32, dtype=torch.uint8, device='cuda'); b = torch.empty_like(a)
for _ in range(2): b.copy_(a)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): b.copy_(a)
torch.cuda.synchronize(); copy_s = (time.perf_counter() - t) / 5
print("device copy of %.2f GB: %.3f ms = %.2f TB/s read + write (%.2f TB/s each way)" % (n * 32 / 1e9, copy_s * 1e3, 2 * n * 32 / copy_s / 1e12, n * 32 / copy_s / 1e12), flush=True)
del a, b
K, R = 2, 35
qrows = rng.choice(n, 64, replace=False)
flips = np.packbits(rng.random((64, 256)) < 0.08, axis=1, bitorder="little")
qall = desc[qrows] ^ flips
res = {}
for Q in (1, 4, 8, 16, 32, 64):
    q = np.ascontiguousarray(qall[:Q]); d_q = torch.from_numpy(q).cuda()
    for eng in ("valu", "mfma"):
        ctx.set_matcher_engine(eng)
        d_c = torch.zeros(Q, dtype=torch.int32, device='cuda'); d_m = torch.zeros((Q * K, 4), dtype=torch.int32, device='cuda'); d_x = torch.zeros((Q * K, 3), device='cuda')
        call = lambda: ctx.match_device(d_q.data_ptr(), Q, K, R, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
        for _ in range(2): call()
        ctx.synchronize(); ctx.set_kernel_timing(True); c0 = ctx.counters()
        for _ in range(5): call()
        ctx.synchronize(); c1 = ctx.counters(); ctx.set_kernel_timing(False)
        ms = (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / 5
        res[(Q, eng)] = (d_c.cpu().numpy().copy(), d_m.cpu().numpy().copy())
        gbs = n * 32 / (ms * 1e-3) / 1e9
        print("Q=%2d %s: DB pass %.3f ms = %.0f GB/s = %.2f of the 8 TB/s HBM roof, %.2f of the measured one-way copy rate" %
              (Q, eng, ms, gbs, gbs / 8000.0, gbs / (n * 32 / copy_s / 1e9)), flush=True)
    same = np.array_equal(res[(Q, "valu")][0], res[(Q, "mfma")][0]) and np.array_equal(res[(Q, "valu")][1], res[(Q, "mfma")][1])
    print("      engines identical: %s; matches %d" % (same, int(res[(Q, "mfma")][0].sum())), flush=True)
import oracle_lib as O
keys = O.knn_keys(desc, qall[:4], K)
cnt, m = res[(4, "mfma")]
m = m.reshape(4, K, 4)
ok = True
for qi in range(4):
    want = [(int(kk) >> 32, int(kk) & 0xFFFFFFFF) for kk in keys[qi] if (int(kk) >> 32) <= R]
    got = [(int(np.float32(m[qi, j, 3:4].view(np.float32)[0])), int(m[qi, j, 2]) * 5000 + int(m[qi, j, 1])) for j in range(cnt[qi])]
    ok = ok and got == want
print("4 queries against the CPU oracle over all %d rows: %s" % (n, "identical" if ok else "MISMATCH"))
