import ctypes as C, importlib, os, sys
import numpy as np
sys.path.insert(0, "/root/repo")
gorio = importlib.import_module("go-rio_amd"); synth = gorio.synth
lib = gorio.load_library()
out = (C.c_ulonglong * 24)()
objs=[]
for q in range(8):
    sx, sl, tx, tl, _ = synth.scan_pair(16384, 16384, seed=synth.BASE_SEED + 3 + q)
    o = gorio.ApdGicp(corr_dist_threshold=2.0, search=1); o.setInputTarget(tx, tl); o.setInputSource(sx, sl); objs.append(o)
lib.gorio_debug_search_stats(out, 1)
for o in objs: o.calculateCovariances()
lib.gorio_debug_search_stats(out, 1)
v=[int(x) for x in out]
print("collect waves", v[0], "tiles per wave", v[1]/max(v[0],1), "needed (lane,tile) per wave", v[2]/max(v[0],1), "=> per lane", v[2]/max(v[0],1)/64)
