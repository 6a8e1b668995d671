import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, time
import numpy as np
gorio = importlib.import_module("go-rio_amd")
synth = gorio.synth
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
sx, sl = synth.radar_scan(16384, seed=5)
t0 = time.perf_counter(); tx, tl = synth.local_map(m, seed=6); print("map generated", tx.shape, round(time.perf_counter() - t0, 1), "s")
g = gorio.ApdGicp(corr_dist_threshold=2.0, search=1, max_iterations=10, optimizer=0, rotation_epsilon=0.0, transformation_epsilon=0.0)
for rep in range(2):
    g.setInputTarget(tx, tl); g.setInputSource(sx, sl)
    g.setProfiling(True)
    t = time.perf_counter(); r = g.align(); dt = time.perf_counter() - t
    s, c = g.getStageTimes()
    print("align ms", round(dt * 1e3, 2), "stages ms [knn+index, nn, lin, solve]", [round(x * 1e3, 3) for x in s], c, "nn us/launch", round(s[1] / max(c[1], 1) * 1e6, 1))
