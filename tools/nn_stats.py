"""Work counters and per-phase cycles of nn_search_pruned_kernel (development aid; needs a library built with -DGORIO_STATS, given
through GORIO_AMD_LIB).  usage: python tools/nn_stats.py c4|c5 [iterations]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gorio = importlib.import_module("go-rio_amd")
synth = gorio.synth
shape = sys.argv[1] if len(sys.argv) > 1 else "c4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = gorio.load_library()
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
out = (C.c_ulonglong * 24)()
gorio.align_batch(objs, guesses)  # warm-up (index, covariances)
for its in (1, iters):
    for o in objs:
        o.set_params(max_iterations=its)
    lib.gorio_debug_search_stats(out, 1)
    objs[0].setProfiling(True)
    gorio.align_batch(objs, guesses)
    st, cn = objs[0].getStageTimes()
    lib.gorio_debug_search_stats(out, 1)
    v = [int(out[k]) for k in range(24)]
    w = max(v[5], 1)
    names = {1: "groups tested", 0: "tiles needed", 7: "fine tests", 3: "items", 4: "rounds", 6: "flushes", 2: "blocks passed"}
    ph = ["preamble", "advance+coarse", "fine test", "staging", "item list", "rounds", "winners", "epilogue"]
    tot = sum(v[8:16])
    print(shape, "iterations", its, "waves", v[5], "nn ms/launch", round(1e3 * st[1] / max(cn[1], 1), 4))
    print("  per wave:", {names[k]: round(v[k] / w, 2) for k in names})
    print("  cycles per wave:", {ph[k]: int(v[8 + k] / w) for k in range(8)}, "total", int(tot / w))
    print("  waves by lifetime (< 32k, < 64k, < 128k, ... cycles):", v[16:24])
    print("  share:", {ph[k]: round(v[8 + k] / max(tot, 1), 3) for k in range(8)})
