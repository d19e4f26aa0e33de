"""What slows the matcher's DB pass when other work shares the GPU? K4 loop on one stream; on a second stream (high priority,
own thread) one kind of noise at a time: memory sweeps of several sizes, a cache-resident arithmetic kernel, many tiny launches."""
import sys, os, time, threading
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
desc, pts, off = synth.make_db(200)
B, nq, k = 16, 1000, 2
q = np.concatenate([synth.make_frame(desc, pts, off, nq, frame=f, visible_object=(17 * f + 3) % 200)["q_desc"] for f in range(B)])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
ctx = capi.Context(0, stream.cuda_stream)
ctx.db_load(desc, pts, off)
d_q = torch.from_numpy(q).cuda(); n = B * nq
d_c = torch.empty(n, dtype=torch.int32, device='cuda'); d_m = torch.empty((n * k, 4), dtype=torch.int32, device='cuda'); d_x = torch.empty((n * k, 3), device='cuda')
junk = torch.empty(1 << 28, dtype=torch.float32, device='cuda')
small = torch.ones(1 << 18, device='cuda')
def run(): ctx.match_device(d_q.data_ptr(), n, k, 35, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
noise = {
    "none": None,
    "fill 32 MB, back to back": lambda: junk[: 32 << 18].fill_(1.0),
    "fill 256 MB, back to back": lambda: junk[: 256 << 18].fill_(1.0),
    "read 256 MB, back to back": lambda: junk[: 256 << 18].sum(),
    "arithmetic on 1 MB (cache resident), back to back": lambda: small.mul_(1.0000001).add_(1e-9),
    "tiny launches (4 KB fill), back to back": lambda: junk[:1024].fill_(2.0),
}
for name, fn in noise.items():
    stop = [False]
    def worker():
        s2 = torch.cuda.Stream(priority=-1)
        with torch.cuda.stream(s2):
            while not stop[0]:
                for _ in range(20): fn()
                s2.synchronize()
    t = threading.Thread(target=worker)
    if fn: t.start(); time.sleep(0.05)
    for _ in range(2): run()
    ctx.synchronize(); ctx.set_kernel_timing(True); c0 = ctx.counters()
    for _ in range(15): run()
    ctx.synchronize(); c1 = ctx.counters(); ctx.set_kernel_timing(False)
    stop[0] = True
    if fn: t.join()
    print("noise: %-52s K4 %.3f ms per launch" % (name, (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / (c1.n_match_kernel_launches - c0.n_match_kernel_launches)), flush=True)
