"""Where a hypothesis evaluation spends its cycles on the bench scene (clock64 stamps of eval_kernel's dbg mode)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import oracle_lib as O
from tod_amd import capi, synth
from test_verify_gpu import _clusters_of
desc, pts, off = synth.make_db(200)
fr = synth.make_frame(desc, pts, off, 1000, frame=1, visible_object=20)
ctx = capi.Context(0)
spans = ctx.db_load(desc, pts, off)
row_ptr, m, xyz = ctx.match(fr["q_desc"], 2, 35)
sc = dict(kp_xy=fr["kp_xy"], cloud=fr["cloud"], row_ptr=row_ptr, matches=m, matches_xyz=xyz)
cl = _clusters_of(sc)
print({o: len(v[2]) for o, v in cl.items() if len(v[2]) >= 3})
t, q, qi = cl[20]
oc = O.Cluster(t, q, qi); oc.fill(fr["kp_xy"], float(spans[20]), 0.01)
rng = O.rng_new(1)
triples = np.array([oc.draw(rng) for _ in range(64)], np.uint32)
stride = 2 + 2 * len(qi) + 16
counts, dbg = ctx.test_consensus(t, q, fr["kp_xy"][qi], float(spans[20]), 0.01, triples, stop_level=0, dbg_stride=stride)
print("n", len(qi), "counts", counts[:16])
for i in range(10):
    print("it", i, "cnt", dbg[i, 0], "m", dbg[i, 1], "cycles flist/adjc/search", dbg[i, -6:-3], "steps", dbg[i, -3], "q", dbg[i, -2],
          "| isect/sort/colour cycles", dbg[i, -12:-9], "colour cycles on lists > 64 / their vertices / all coloured vertices", dbg[i, -9:-6])
