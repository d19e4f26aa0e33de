"""todhip_db_load (host attachments -> object-aligned device shard, spans on the device) for 1M and 10M rows."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
import numpy as np
from tod_amd import capi, synth
for n_obj in (200, 2000):
    desc, pts, off = synth.make_db(n_obj)
    ctx = capi.Context(0)
    ctx.db_load(desc, pts, off); ctx.synchronize()
    t=time.perf_counter()
    for _ in range(3): sp = ctx.db_load(desc, pts, off)
    ctx.synchronize()
    print("db_load %d objects x 5000 rows (%.0f MB): %.1f ms" % (n_obj, desc.nbytes/1e6, (time.perf_counter()-t)/3*1e3))
    ctx.close()
