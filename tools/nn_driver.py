"""One batched align (64 pairs, 20 fixed Gauss-Newton iterations) on the C4 or C5 shape, for rocprofv3 runs that look at the
correspondence search alone (development aid).  usage: python3 tools/nn_driver.py c4|c5 [iterations] [repeats]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gorio = importlib.import_module("go-rio_amd")
synth = gorio.synth
shape = sys.argv[1] if len(sys.argv) > 1 else "c4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
params = dict(corr_dist_threshold=2.0, search=1, max_iterations=iters, optimizer=0, rotation_epsilon=0.0, transformation_epsilon=0.0)
objs, guesses = [], []
if shape == "c4":
    for q in range(64):
        sx, sl, tx, tl, _ = synth.scan_pair(16384, 16384, seed=synth.BASE_SEED + 3 + q)
        o = gorio.ApdGicp(**params)
        o.setInputTarget(tx, tl)
        o.setInputSource(sx, sl)
        objs.append(o)
        guesses.append(np.eye(4, dtype=np.float32))
else:
    m = 1000000
    n_scans = max(6, m // 16384)
    tx, tl = synth.local_map(m, seed=synth.BASE_SEED + 77, n_scans=n_scans)
    for q in range(64):
        pose = np.eye(4)
        pose[0, 3] = 0.8 * ((q * 7) % n_scans)
        sx, sl = synth.radar_scan(16384, seed=synth.BASE_SEED + 3 + q, sensor_pose=pose)
        o = gorio.ApdGicp(**params)
        if q == 0:
            o.setInputTarget(tx, tl)
        else:
            o.setInputTargetShared(objs[0])
        o.setInputSource(sx, sl)
        objs.append(o)
        guesses.append(pose.astype(np.float32))
guesses = np.stack(guesses)
objs[0].setProfiling(True)
for _ in range(reps + 1):
    gorio.align_batch(objs, guesses)
st, cn = objs[0].getStageTimes()
print(shape, "nn ms/launch", round(1e3 * st[1] / max(cn[1], 1), 4), "launches", cn[1])
