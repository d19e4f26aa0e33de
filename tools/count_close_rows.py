import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tod_amd import capi, scenes
tex = scenes.make_textures(200)
ctx = capi.Context(0)
desc, pts, off = scenes.train_db(ctx, tex, rows_per_object=5000)
bt = scenes.make_detection_batches(tex, 1, 16)[0]
imgs = bt["images"].cpu().numpy().reshape(16, 480, 640)
kp, aux, de = ctx.orb(imgs[0], 1000, 3, 1.2)
lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int16, device="cuda")
D = torch.from_numpy(desc).cuda()
Q = torch.from_numpy(de[:200]).cuda()
cnt35 = torch.zeros(len(Q), dtype=torch.int64, device="cuda"); cnt20 = torch.zeros_like(cnt35); cnt10 = torch.zeros_like(cnt35)
for r0 in range(0, len(D), 50000):
    x = (Q[:, None, :] ^ D[None, r0:r0 + 50000, :]).long()
    d = lut[x].sum(-1)
    cnt35 += (d <= 35).sum(1); cnt20 += (d <= 20).sum(1); cnt10 += (d <= 10).sum(1)
print("rows within 35 / 20 / 10 bits per query (200 queries): median %d / %d / %d, mean %.0f / %.0f / %.0f, max %d" % (
    cnt35.median(), cnt20.median(), cnt10.median(), cnt35.float().mean(), cnt20.float().mean(), cnt10.float().mean(), cnt35.max()))
