import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
import oracle_lib as O
from tod_amd import capi, synth
from test_verify_gpu import _clusters_of
sc = synth.make_verify_scene(300, visible=((1, 0.30),), seed=300)
t, q, qi = _clusters_of(sc)[1]
cl = O.Cluster(t, q, qi); cl.fill(sc["kp_xy"], float(sc["spans"][1]), 0.01)
rng = O.rng_new(1)
triples = np.array([cl.draw(rng) for _ in range(64)], np.uint32)
ctx = capi.Context(0)
lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 1
counts, dbg = ctx.test_consensus(t, q, sc["kp_xy"][qi], float(sc["spans"][1]), 0.01, triples, stop_level=lvl, dbg_stride=600)
print('level', lvl, 'returned; counts', counts[:16])
samp = cl.bits(1); deg = np.array([[bin(int(x)).count('1') for x in row] for row in samp]).sum(1)
bad = 0
for i, tr in enumerate(triples):
    inl, gc, gs = cl.consensus(tr)
    # oracle F: members of (P + samples) with sample degree >= 7
    P = set(np.flatnonzero((np.unpackbits(cl.bits(0)[tr[0]].view(np.uint8), bitorder='little') & np.unpackbits(cl.bits(0)[tr[1]].view(np.uint8), bitorder='little') & np.unpackbits(cl.bits(0)[tr[2]].view(np.uint8), bitorder='little'))).tolist()) | set(int(x) for x in tr)
    F = sorted(v for v in P if deg[v] >= 7)
    m = int(dbg[i, 1]); gF = dbg[i, 2:2 + 2 * m:2].tolist()
    ok = (int(dbg[i, 0]) == len(P)) and (m == 0 or gF == F)
    if not ok or i < 3:
        print(i, tr, "gpu cnt", dbg[i, 0], "m", m, "counts", counts[i], "| oracle |P|", len(P), "|F|", len(F), "oracle consensus", len(inl), gc, gs, "OK" if ok else "MISMATCH")
        if not ok: print("   gpu F", gF[:20], "\n   orc F", F[:20]); bad += 1
print("mismatches", bad)

if lvl == 0:
    for i in range(6):
        print("it", i, "cnt", dbg[i,0], "m", dbg[i,1], "cycles flist/adjc/search", dbg[i,-6:-3], "steps", dbg[i,-3], "q", dbg[i,-2])
