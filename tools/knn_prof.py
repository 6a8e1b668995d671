import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, sys, time, ctypes as C
import numpy as np
gorio = importlib.import_module("go-rio_amd")
synth = gorio.synth
npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pairs = [synth.scan_pair(16384, 16384, seed=100 + q) for q in range(npairs)]
objs = [gorio.ApdGicp(corr_dist_threshold=2.0, search=1, max_iterations=iters, optimizer=0, rotation_epsilon=0.0, transformation_epsilon=0.0) for _ in range(npairs)]
lib = gorio.apd.load_library()
have_stats = hasattr(lib, "gorio_debug_search_stats")
for rep in range(2):
    for o, (sx, sl, tx, tl, T) in zip(objs, pairs):
        o.setInputTarget(tx, tl); o.setInputSource(sx, sl)
    objs[0].setProfiling(True)
    if have_stats:
        st = (C.c_ulonglong * 8)(); lib.gorio_debug_search_stats(st, 1)
    gorio.align_batch(objs)
    s, c = objs[0].getStageTimes()
print("stages ms", [round(x * 1e3, 3) for x in s], c)
if have_stats:
    st = (C.c_ulonglong * 8)(); lib.gorio_debug_search_stats(st, 0)
    v = list(st)
    print("knn: waves", v[0], "coarse tiles/wave", v[1] / max(v[0], 1), "evaluated tiles/wave", v[2] / max(v[0], 1), "insert rounds/wave", v[3] / max(v[0], 1), "lanes needing a tile", v[4] / max(v[2], 1))
    print("nn: waves", v[5], "needed (union) tiles/wave", v[6] / max(v[5], 1), "coarse-passed tiles/wave", v[7] / max(v[5], 1), "work-list rounds/wave", v[4] / max(v[5], 1), "items of lane 0 per wave", v[3] / max(v[5], 1))
