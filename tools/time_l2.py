"""BASELINE configs[3] shape: 1000 SIFT-like float queries vs a 500k-row float DB, k = 2 (tod_amd/csrc/l2.hip)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, synth
n_obj = int(sys.argv[1]) if len(sys.argv) > 1 else 100
desc, pts, off = synth.make_sift_db(n_obj)
q, truth = synth.make_sift_queries(desc, 1000, frame=1)
ctx = capi.Context(0)
ctx.db_load(desc, pts, off)
nq, k = 1000, 2
d_q = torch.from_numpy(q).cuda(); d_c = torch.empty(nq, dtype=torch.int32, device='cuda')
d_m = torch.empty((nq * k, 4), dtype=torch.int32, device='cuda'); d_x = torch.empty((nq * k, 3), device='cuda')
def run(): ctx.match_l2_device(d_q.data_ptr(), nq, k, 400.0, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
for _ in range(3): run()
ctx.synchronize(); t = time.perf_counter()
n = 20
for _ in range(n): run()
ctx.synchronize(); dt = (time.perf_counter() - t) / n
flop = 2.0 * nq * desc.shape[0] * 128
print("match_l2_device Q=%d N=%d k=%d: %.3f ms per call; one GEMM pass = %.1f GFLOP, two passes at %.1f TFLOP/s overall"
      % (nq, desc.shape[0], k, dt * 1e3, flop / 1e9, 2 * flop / dt / 1e12))
cnt = d_c.cpu().numpy(); m = d_m.cpu().numpy().view(capi.DMATCH_DTYPE).reshape(nq, k)
planted = truth >= 0
glob = off[m["imgIdx"][:, 0]].astype(np.int64) + m["trainIdx"][:, 0]
print("queries with a match within radius: %d; planted found as nearest: %.3f" % ((cnt > 0).sum(), (glob[planted] == truth[planted]).mean()))
