"""Work counters of the pruned 1-NN search on the C5 shape: 64 scans against a shared 1 M-point map (development aid; needs a
library built with -DGORIO_STATS)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gorio = importlib.import_module("go-rio_amd")
synth = gorio.synth
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
n_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n_scans = max(6, m // 16384)
tx, tl = synth.local_map(m, seed=synth.BASE_SEED + 77, n_scans=n_scans)
lib = gorio.load_library()
params = dict(corr_dist_threshold=2.0, search=1, max_iterations=1, optimizer=0, rotation_epsilon=0.0, transformation_epsilon=0.0)
objs, guesses = [], []
for q in range(n_pairs):
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
out = (C.c_ulonglong * 24)()
names = ["rounds in heavy waves", "heavy waves (> 64 rounds)", "max rounds of a wave", "lane0 items", "eval rounds", "waves", "tiles needed", "tiles past coarse"]
prev = None
for it in (1, 2, 3, 6, 10):
    for o in objs:
        o.set_params(max_iterations=it)
    lib.gorio_debug_search_stats(out, 1)
    res = gorio.align_batch(objs, guesses)
    lib.gorio_debug_search_stats(out, 1)
    cur = [int(out[k]) for k in range(8)]
    w = max(cur[5], 1)
    unmatched = 0
    for o in objs[:8]:
        c, _ = o.getCorrespondences()
        unmatched += int((c < 0).sum())
    print("align_batch with", it, "iterations (cumulative):", {names[k]: round(cur[k] / w, 2) for k in range(3, 8)}, "waves", cur[5], {names[k]: cur[k] for k in range(3)},
          "unmatched share (first 8 scans)", round(unmatched / (8 * 16384), 3))
