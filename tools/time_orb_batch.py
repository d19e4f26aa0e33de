"""todhip_orb_batch_device on the bench's 16-frame step (640x480, ORB-1000, 3 levels), alone."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
B = int(os.environ.get("B", "16"))
ctx = capi.Context(0)
if os.environ.get("SCENES"):                              # the chained block's rendered detection views instead of the 8(d) image
    from tod_amd import scenes
    d = scenes.make_detection_batches(scenes.make_textures(200), 1, B)[0]["images"]
else:
    d = torch.from_numpy(np.stack([synth.make_image(f % 8) for f in range(B)])).cuda()
kp = torch.empty((B, 1000, 2), device='cuda'); aux = torch.empty((B, 1000, 4), device='cuda'); desc = torch.empty((B, 1000, 32), dtype=torch.uint8, device='cuda')
def run(): return ctx.orb_batch_device(d.data_ptr(), B, 480 * 640, 480, 640, 640, 1000, 3, 1.2, kp.data_ptr(), aux.data_ptr(), desc.data_ptr(), 1000)
for _ in range(3): run()
t = time.perf_counter(); n = 20
for _ in range(n): r = run()
dt = (time.perf_counter() - t) / n
print("orb_batch_device %d frames: %.3f ms per batch (%.1f us per frame), %d keypoints" % (B, dt * 1e3, dt * 1e6 / B, sum(r)))
