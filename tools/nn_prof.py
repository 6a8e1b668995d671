import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, sys, time
import numpy as np
gorio = importlib.import_module("go-rio_amd")
synth = gorio.synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
m = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
sx, sl, tx, tl, T = synth.scan_pair(n, m, seed=5) if m <= 20000 else (*synth.radar_scan(n, seed=5), *synth.local_map(m, seed=6), synth.gt_transform())
for mode in (0, 1):
    g = gorio.ApdGicp(corr_dist_threshold=2.0, search=mode, max_iterations=20, optimizer=0, rotation_epsilon=0.0, transformation_epsilon=0.0)
    g.setInputTarget(tx, tl); g.setInputSource(sx, sl)
    g.align()
    g.setInputTarget(tx, tl); g.setInputSource(sx, sl)
    g.setProfiling(True)
    t = time.perf_counter(); r = g.align(); dt = time.perf_counter() - t
    s, c = g.getStageTimes()
    print("mode", mode, "align ms", round(dt * 1e3, 3), "stages ms", [round(x * 1e3, 3) for x in s], c, "nn us/launch", round(s[1] / max(c[1], 1) * 1e6, 1))
