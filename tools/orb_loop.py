import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import torch
from tod_amd import capi, synth
img = synth.make_image(1); ctx = capi.Context(0)
d_img = torch.from_numpy(img).cuda(); d_kp = torch.empty((1000, 2), device='cuda'); d_aux = torch.empty((1000, 4), device='cuda'); d_desc = torch.empty((1000, 32), dtype=torch.uint8, device='cuda')
for _ in range(20):
    n = ctx.orb_device(d_img.data_ptr(), 480, 640, 640, 1000, 3, 1.2, d_kp.data_ptr(), d_aux.data_ptr(), d_desc.data_ptr(), 1000)
print(n)
