"""Cuts tests/golden/chained_frames.npz out of gpurun_out/chained_frames.npz (written on the GPU box by tools/chained_objects.py:
this library's own ORB keypoints and matcher output on rendered views against the 200-object trained DB). A fixture is data:
keypoints, fixed-stride match lists and spans of a few frames; the cloud is the constant-depth plane the frames were rendered on
(back-projected in the test). The frames are kept if the CPU oracle replays them (it is the checker of the GPU test)."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O

src = np.load(os.path.join(ROOT, "gpurun_out", "chained_frames.npz"))
frames = [int(a) for a in sys.argv[1:]] or [8, 10, 13]
kp, counts, matches, xyz = src["kp"][frames], src["counts"][frames], src["matches"][frames], src["xyz"][frames]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "chained_frames.npz"), kp=kp.astype(np.float32), counts=counts.astype(np.uint32),
                    matches=matches.astype(np.int32), xyz=xyz.astype(np.float32), spans=src["spans"], K=src["K"], Z=src["Z"],
                    frames=np.asarray(frames), objects=src["objects"][frames])
print("wrote", os.path.getsize(os.path.join(ROOT, "tests", "golden", "chained_frames.npz")), "bytes")
