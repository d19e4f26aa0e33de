"""Where the clique gate of the true object's hypotheses spends its cycles on the data-chained frames (the 400-470-match objects of
bench `chained`): per hypothesis |F|, which LDS pass it takes, steps, clique size, cycles by phase (eval_kernel's dbg mode)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np, torch
import oracle_lib as O
from tod_amd import capi, scenes
from test_verify_gpu import _clusters_of
n_obj, nq, k, radius = 200, 1000, 2, 35
tex = scenes.make_textures(n_obj)
ctx = capi.Context(0)
desc, pts, off = scenes.train_db(ctx, tex, rows_per_object=5000)
spans = ctx.db_load(desc, pts, off)
bt = scenes.make_detection_batches(tex, 1, 16)[0]
kp = torch.zeros((16, nq, 2), device='cuda'); aux = torch.zeros((16, nq, 4), device='cuda'); de = torch.zeros((16, nq, 32), dtype=torch.uint8, device='cuda')
ctx.orb_batch_device(bt["images"].data_ptr(), 16, 480 * 640, 480, 640, 640, nq, 3, 1.2, kp.data_ptr(), aux.data_ptr(), de.data_ptr(), nq)
ctx.synchronize()
depth = bt["depth"].cpu().numpy().reshape(16, 480, 640)
fx, fy, cx, cy = scenes.K[0, 0], scenes.K[1, 1], scenes.K[0, 2], scenes.K[1, 2]
for f in [int(a) for a in sys.argv[1:]] or [0, 1, 2]:
    kpf = kp[f].cpu().numpy(); q = de[f].cpu().numpy()
    row_ptr, m, xyz = ctx.match(q, k, radius)
    z = depth[f].astype(np.float32)
    u, v = np.meshgrid(np.arange(640, dtype=np.float32), np.arange(480, dtype=np.float32))
    cloud = np.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], -1).astype(np.float32)
    cloud[~(z > 0)] = np.nan
    cl = _clusters_of(dict(kp_xy=kpf, cloud=cloud, row_ptr=row_ptr, matches=m, matches_xyz=xyz))
    obj = max(cl, key=lambda o: len(cl[o][2]))
    t, qq, qi = cl[obj]
    oc = O.Cluster(t, qq, qi); oc.fill(kpf, float(spans[obj]), 0.01)
    rng = O.rng_new(1)
    triples = np.array([oc.draw(rng) for _ in range(8)], np.uint32)
    stride = 2 + 2 * len(qi) + 16
    t0 = time.perf_counter()
    counts, dbg = ctx.test_consensus(t, qq, kpf[qi], float(spans[obj]), 0.01, triples, stop_level=0, dbg_stride=stride)
    print("frame %d object %d n=%d counts %s (%.2f ms for 8 hypotheses incl. copies)" % (f, obj, len(qi), counts.tolist(), 1e3 * (time.perf_counter() - t0)), flush=True)
    for i in range(4):
        print("  it", i, "cnt", dbg[i, 0], "m", dbg[i, 1], "cycles flist/adjc/search", dbg[i, -6:-3], "steps", dbg[i, -3], "q", dbg[i, -2],
              "| isect/sort/colour", dbg[i, -12:-9], "colour>64 cycles / vertices / all coloured", dbg[i, -9:-6], flush=True)
