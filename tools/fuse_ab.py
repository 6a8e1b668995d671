"""Stage times of a C4-shaped batch align with the optimiser step fused into the linearisation launch or run as its own launch
(gorio_apd_debug_set_schedule): development aid."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gorio = importlib.import_module("go-rio_amd")
synth = gorio.synth
pairs = [synth.scan_pair(16384, 16384, seed=synth.BASE_SEED + 3 + q) for q in range(64)]
for fuse in (True, False, True, False):
    objs = []
    for sx, sl, tx, tl, _ in pairs:
        o = gorio.ApdGicp(corr_dist_threshold=2.0, search=1, max_iterations=20, optimizer=0, rotation_epsilon=0.0, transformation_epsilon=0.0)
        o.setInputTarget(tx, tl)
        o.setInputSource(sx, sl)
        o.debugSetSchedule(fuse_step=fuse, plan_search=True)
        objs.append(o)
    gorio.align_batch(objs)
    objs[0].setProfiling(True)
    gorio.align_batch(objs)
    s, c = objs[0].getStageTimes()
    print("fuse", fuse, "ms:", {k: round(1e3 * v, 3) for k, v in zip(("knn", "nn", "linearize", "solve", "index"), s[:5])}, c[:5])
