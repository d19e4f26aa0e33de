"""The matcher's DB pass on the data it will see: the bench launch shape (32 000 queries x 1M rows, k = 2, radius 35) on the `chained`
block's trained DB (this library's ORB descriptors of rendered views: biased, correlated bits, self-similar textures) with the ORB
descriptors of 32 rendered detection views as queries, beside the same launch on SURVEY 8(d)'s independent bits. Prints ms per launch
(HIP events on the context's stream) and the fraction of the fp4 MFMA roof; with the diagnostics build of tools/k4x_walks.sh loaded
(TODHIP_LIB_PATH), also the fraction of 32 x 32 accumulator blocks whose 16 rows were walked."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
from tod_amd import capi, scenes, synth
K, R, nq, F = 2, 35, 1000, 32
lib = capi.lib()
has_walks = hasattr(lib, "todhip_debug_k4x_walks")
out = {}

def timed(ctx, d_q, n, tag):
    d_c = torch.zeros(n, dtype=torch.int32, device='cuda'); d_m = torch.zeros((n * K, 4), dtype=torch.int32, device='cuda'); d_x = torch.zeros((n * K, 3), device='cuda')
    call = lambda: ctx.match_device(d_q.data_ptr(), n, K, R, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr())
    for _ in range(6): call(); ctx.synchronize()        # (a fresh context walks the block forms from its launches' reports: settled after three)
    if has_walks: lib.todhip_debug_k4x_walks(None, 1)
    ctx.set_kernel_timing(True); c0 = ctx.counters()
    for _ in range(6): call()
    ctx.synchronize(); c1 = ctx.counters(); ctx.set_kernel_timing(False)
    ms = (c1.sum_match_kernel_ms - c0.sum_match_kernel_ms) / (c1.n_match_kernel_launches - c0.n_match_kernel_launches)
    rows = ctx.db_info()["total_rows"]
    pf = 512.0 * n * rows / (ms * 1e-3) / 1e15
    res = dict(ms_per_launch=ms, PFLOPs=pf, frac_of_10PF=pf / 10.0, queries=n, rows=int(rows), matches=int(d_c.sum().item()))
    if has_walks:
        w = (C.c_ulonglong * 2)()
        lib.todhip_debug_k4x_walks(w, 0)
        res.update(blocks_tested=int(w[0]), blocks_walked=int(w[1]), walk_fraction=w[1] / max(w[0], 1))
    print(tag, json.dumps(res), flush=True)
    out[tag] = res

# ---- the chained block's DB and queries
tex = scenes.make_textures(200)
ctx = capi.Context(0)
desc, pts, off = scenes.train_db(ctx, tex, rows_per_object=5000)
ctx.db_load(desc, pts, off); ctx.set_matcher_engine("mfma")
bts = scenes.make_detection_batches(tex, 2, 16)
de = torch.zeros((F, nq, 32), dtype=torch.uint8, device='cuda')
kp = torch.zeros((16, nq, 2), device='cuda'); aux = torch.zeros((16, nq, 4), device='cuda')
for b, bt in enumerate(bts):
    ctx.orb_batch_device(bt["images"].data_ptr(), 16, 480 * 640, 480, 640, 640, nq, 3, 1.2, kp.data_ptr(), aux.data_ptr(), de[16 * b:].data_ptr(), nq)
ctx.synchronize()
timed(ctx, de, F * nq, "chained_db")
ctx.close()
if os.environ.get("ONLY_CHAINED"): sys.exit(0)
# ---- independent bits, planted queries
desc, pts, off = synth.make_db(200)
ctx = capi.Context(0); ctx.db_load(desc, pts, off); ctx.set_matcher_engine("mfma")
q = np.concatenate([synth.make_frame(desc, pts, off, nq, frame=f, visible_object=(17 * f + 3) % 200)["q_desc"] for f in range(F)])
timed(ctx, torch.from_numpy(q).cuda(), F * nq, "independent_bits")
json.dump(out, open(os.path.join(os.path.dirname(__file__), '..', 'gpurun_out', 'k4x_on_chained_db%s.json' % ("_walks" if has_walks else "")), "w"), indent=1)
