"""The matcher's elimination schedules on descriptors with CONTROLLED correlation (a stand-in for real rBRIEF statistics,
which this image has no data for): bit j of a descriptor = sign(a * <u, L_j> + sqrt(1 - a^2) * noise) with an 8-dim latent
u per descriptor -- per-bit mean 0.5, non-match distance 128 with a standard deviation set by `a` (independent bits: 8).
For each correlation level: distance statistics, and the DB pass of 16 x 1000 queries x 1M rows at radius 35 under each
schedule (TODHIP_K4_MODE, one process per mode because the override is read once)."""
import sys, os, subprocess, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np

def gen(n, a, rng, L):
    out = np.empty((n, 32), np.uint8)
    for i0 in range(0, n, 100000):
        m = min(100000, n - i0)
        x = a * (rng.normal(0, 1, (m, L.shape[0])).astype(np.float32) @ L) + np.float32(np.sqrt(1 - a * a)) * rng.normal(0, 1, (m, 256)).astype(np.float32)
        out[i0:i0 + m] = np.packbits(x > 0, axis=1)
    return out

def child(a, mode):
    import torch
    from tod_amd import capi
    rng = np.random.Generator(np.random.PCG64(77))
    L = rng.normal(0, 1, (8, 256)).astype(np.float32); L /= np.linalg.norm(L, axis=0, keepdims=True)
    db = gen(1000000, a, rng, L)
    q = gen(16000, a, rng, L)
    planted = rng.choice(16000, 4800, replace=False)                     # 30 % of the queries: a DB row with 8 % of the bits flipped
    rows = rng.integers(0, 1000000, 4800)
    q[planted] = db[rows] ^ np.packbits(rng.random((4800, 256)) < 0.08, axis=1)
    if mode == 2:
        s = np.unpackbits(db[:3000], axis=1).astype(np.int16); t = np.unpackbits(q[~np.isin(np.arange(16000), planted)][:300], axis=1).astype(np.int16)
        d = (t[:, None, :] != s[None, :, :]).sum(-1)
        d96 = (t[:, None, :96] != s[None, :, :96]).sum(-1); d128 = (t[:, None, :128] != s[None, :, :128]).sum(-1)
        print("a=%.2f: non-match distance %.1f +- %.1f; per pair P(d96 < 36) = %.4f, P(d128 < 36) = %.5f" %
              (a, d.mean(), d.std(), (d96 < 36).mean(), (d128 < 36).mean()), flush=True)
    ctx = capi.Context(0)
    pts = np.zeros((1000000, 3), np.float32); off = (np.arange(201) * 5000).astype(np.uint32)
    ctx.db_load(db, pts, off)
    n, k = 16000, 2
    d_q = torch.from_numpy(q).cuda()
    d_c = torch.empty(n, dtype=torch.int32, device='cuda'); d_m = torch.empty((n * k, 4), dtype=torch.int32, device='cuda'); d_x = torch.empty((n * k, 3), device='cuda')
    for _ in range(2): ctx.match_device(d_q.data_ptr(), n, k, 35, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
    ctx.synchronize(); ctx.set_kernel_timing(True); c0 = ctx.counters()
    for _ in range(5): ctx.match_device(d_q.data_ptr(), n, k, 35, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
    ctx.synchronize(); c1 = ctx.counters()
    k4 = (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / (c1.n_match_kernel_launches - c0.n_match_kernel_launches)
    print("a=%.2f schedule %d: K4 %.3f ms per 16 frames (%.4f ms per frame); queries with a match %d, checksum %d" %
          (a, mode, k4, k4 / 16, int((d_c.cpu().numpy() > 0).sum()), int(d_m.cpu().numpy().astype(np.int64).sum())), flush=True)

if __name__ == "__main__":
    if len(sys.argv) == 3:
        child(float(sys.argv[1]), int(sys.argv[2]))
    else:
        for a in (0.0, 0.6, 0.8, 0.9):
            for mode in (2, 1, 0):
                env = dict(os.environ, TODHIP_K4_MODE=str(mode))
                subprocess.run([sys.executable, __file__, str(a), str(mode)], env=env, check=True)
