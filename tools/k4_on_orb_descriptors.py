"""How much does the partial-distance elimination prune on descriptors that are NOT independent random bits?
DB = 1M rBRIEF descriptors computed by this repo's ORB on 1000 synthetic images (rectangles + noise, the 8(d) image
generator), queries = ORB descriptors of 16 further images. Compares the matcher's time with the synthetic-bit DB."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
ctx = capi.Context(0)
B = 50
def orb_descs(first, count):
    out = []
    for i0 in range(first, first + count, B):
        imgs = np.stack([synth.make_image(1000 + i) for i in range(i0, i0 + B)])
        d = torch.from_numpy(imgs).cuda()
        kp = torch.empty((B, 1000, 2), device='cuda'); aux = torch.empty((B, 1000, 4), device='cuda'); de = torch.zeros((B, 1000, 32), dtype=torch.uint8, device='cuda')
        n = ctx.orb_batch_device(d.data_ptr(), B, 480 * 640, 480, 640, 640, 1000, 3, 1.2, kp.data_ptr(), aux.data_ptr(), de.data_ptr(), 1000)
        assert min(n) == 1000
        out.append(de.cpu().numpy().reshape(-1, 32))
    return np.concatenate(out)
t = time.time()
db = orb_descs(0, 1000)
q = orb_descs(5000, 50)[:16000]
print("1M ORB descriptors from 1000 images in %.1f s (incl. image synthesis)" % (time.time() - t), flush=True)
bits = np.unpackbits(db[:200000], axis=1)
print("mean bit value %.3f; per-bit mean range [%.3f, %.3f]" % (bits.mean(), bits.mean(0).min(), bits.mean(0).max()))
sample = np.unpackbits(db[:2000], axis=1).astype(np.int32); qs = np.unpackbits(q[:200], axis=1).astype(np.int32)
d = (qs[:, None, :] != sample[None, :, :]).sum(-1)
print("Hamming distance query-vs-DB sample: mean %.1f, std %.1f, P(d <= 35) = %.2e, P(d96 < 36) = %.3f" %
      (d.mean(), d.std(), (d <= 35).mean(), ((qs[:, None, :96] != sample[None, :, :96]).sum(-1) < 36).mean()))
pts = np.zeros((len(db), 3), np.float32); off = (np.arange(201) * 5000).astype(np.uint32)
for name, dbx in (("ORB descriptors", db), ("independent random bits", synth.make_db(200)[0])):
    ctx.db_load(dbx, pts, off)
    n, k = 16000, 2
    d_q = torch.from_numpy(q).cuda()
    d_c = torch.empty(n, dtype=torch.int32, device='cuda'); d_m = torch.empty((n * k, 4), dtype=torch.int32, device='cuda'); d_x = torch.empty((n * k, 3), device='cuda')
    for radius in (35, 55):
        for _ in range(2): ctx.match_device(d_q.data_ptr(), n, k, radius, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
        ctx.synchronize(); ctx.set_kernel_timing(True); c0 = ctx.counters()
        for _ in range(5): ctx.match_device(d_q.data_ptr(), n, k, radius, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
        ctx.synchronize(); c1 = ctx.counters(); ctx.set_kernel_timing(False)
        k4 = (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / (c1.n_match_kernel_launches - c0.n_match_kernel_launches)
        print("%s, radius %d: K4 %.3f ms per 16 x 1000 queries x 1M rows (%.4f ms per frame); queries with a match: %d" %
              (name, radius, k4, k4 / 16, int((d_c.cpu().numpy() > 0).sum())), flush=True)
