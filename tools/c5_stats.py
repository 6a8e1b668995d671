"""Work counters of the pruned 1-NN search on a 1 M-point map (development aid; needs a library built with -DGORIO_STATS)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gorio = importlib.import_module("go-rio_amd")
synth = gorio.synth
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_scans = max(6, m // 16384)
tx, tl = synth.local_map(m, seed=synth.BASE_SEED + 77, n_scans=n_scans)
pose = np.eye(4)
pose[0, 3] = 0.8 * 7
sx, sl = synth.radar_scan(16384, seed=5, sensor_pose=pose)
lib = gorio.load_library()
g = gorio.ApdGicp(corr_dist_threshold=2.0, search=1, max_iterations=1, optimizer=0, rotation_epsilon=0.0, transformation_epsilon=0.0)
g.setInputTarget(tx, tl)
g.setInputSource(sx, sl)
g.calculateCovariances()
out = (C.c_ulonglong * 8)()
T = pose.copy()
names = ["rounds in heavy waves", "heavy waves (> 64 rounds)", "max rounds of a wave", "lane0 items", "eval rounds", "waves", "tiles needed", "tiles past coarse"]
for it in range(6):
    lib.gorio_debug_search_stats(out, 1)
    g.set_params(max_iterations=it + 1)
    r = g.align(pose.astype(np.float32))
    lib.gorio_debug_search_stats(out, 1)
    w = max(out[5], 1)
    print("align with", it + 1, "iterations:", {names[k]: round(out[k] / w, 2) for k in range(3, 8)}, "waves", out[5], {names[k]: out[k] for k in range(3)})
