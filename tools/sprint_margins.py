"""One frame through todhip_verify_device with the sprint's first look-ahead forced (TODHIP_SPRINT_MARGIN, read once per process: a
child per value): the stream's end, and with it sprint_kernel's stop-and-resume path, lands at different points of the frame's rounds.
Every run must end at the oracle's generator position with the oracle's poses."""
import os, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..')); sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tests'))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    import oracle_lib as O
    from tod_amd import capi, synth
    from test_verify_gpu import _pack_scene
    k, nq = 3, 400
    vis = [((1, 0.45),), ((6, 0.40), (2, 0.04)), (), ((3, 0.30), (5, 0.30)), ((7, 0.5),), ((0, 0.03), (4, 0.03), (6, 0.03)), ((2, 0.35),),
           ((1, 0.2), (3, 0.2), (5, 0.2)), ((4, 0.6),), ((0, 0.05),)]
    ctx = capi.Context(0)
    bad = 0
    for i, v in enumerate(vis):
        sc = synth.make_verify_scene(nq, visible=v, seed=640 + i, matches_per_kp=3, n_objects=8)
        c, m, x = _pack_scene(sc, k)
        d_kp = torch.from_numpy(sc["kp_xy"].astype(np.float32)).cuda(); d_cloud = torch.from_numpy(sc["cloud"].astype(np.float32)).cuda()
        d_c = torch.from_numpy(c).cuda(); d_m = torch.from_numpy(m).cuda(); d_x = torch.from_numpy(x).cuda()
        for seed in (3, 4, 5, 6, 7):
            r = capi.rng_new(seed)
            got = ctx.verify_device(d_kp.data_ptr(), nq, d_cloud.data_ptr(), 480, 640, d_c.data_ptr(), d_m.data_ptr(), d_x.data_ptr(), k, sc["spans"], 8, 600, 0.01, r)
            ro = O.rng_new(seed)
            rc, want, _ = O.verify(sc["kp_xy"], sc["cloud"], sc["row_ptr"], sc["matches"], sc["matches_xyz"], sc["spans"], 8, 600, 0.01, ro)
            ok = rc == 0 and r.draws == ro.draws and len(got) == len(want) and all(a["object"] == b["object"] and np.array_equal(a["inliers"], b["inliers"]) for a, b in zip(got, want))
            if not ok:
                bad += 1
                print("  scene %d seed %d: draws %d (oracle %d), poses %d (oracle %d), sprint launches %d" % (i, seed, r.draws, ro.draws, len(got), len(want), ctx.counters().last_sprint_launches), flush=True)
    print("margin %s: %d of %d (scene, seed) pairs differ from the oracle" % (os.environ.get("TODHIP_SPRINT_MARGIN", "default"), bad, len(vis) * 5))
    sys.exit(0)
for mg in os.environ.get("MARGINS", "default 300 1000 3000 10000 30000 100000 200000").split():
    env = dict(os.environ)
    if mg != "default": env["TODHIP_SPRINT_MARGIN"] = mg
    p = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=600)
    print(p.stdout.strip() or p.stderr[-400:], flush=True)
