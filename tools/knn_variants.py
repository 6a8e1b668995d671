"""Times the covariance stage (index build + k-NN + covariance) of the C4 batch for every libgorio_amd variant given on the command
line (development tool: variants are built by hand into tools/variants/*.so)."""
import importlib, os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 or (len(sys.argv) == 2 and not sys.argv[1].endswith(".so")):
    for lib in sys.argv[1:]:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), os.path.abspath(lib) if lib.endswith(".so") else lib], capture_output=True, text=True)
        print(os.path.basename(lib), out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:])
    sys.exit(0)
gorio = importlib.import_module("go-rio_amd")
if len(sys.argv) == 2:
    gorio.LIB_PATH = sys.argv[1]
    importlib.import_module("go-rio_amd.apd").__dict__["_lib"] = None
import numpy as np
synth = gorio.synth
npairs = 64
pairs = [synth.scan_pair(16384, 16384, seed=100 + q) for q in range(npairs)]
objs = [gorio.ApdGicp(corr_dist_threshold=2.0, search=1, max_iterations=2, optimizer=0, rotation_epsilon=0.0, transformation_epsilon=0.0) for _ in range(npairs)]
best = None
for rep in range(4):
    for o, (sx, sl, tx, tl, T) in zip(objs, pairs):
        o.setInputTarget(tx, tl); o.setInputSource(sx, sl)
    objs[0].setProfiling(True)
    gorio.align_batch(objs)
    s, c = objs[0].getStageTimes()
    if rep:
        best = s if best is None else [min(a, b) for a, b in zip(best, s)]
print(json.dumps({"knn_cov_ms": round(best[0] * 1e3, 3), "index_ms": round(best[4] * 1e3, 3), "nn_ms_per": round(best[1] * 1e3 / max(c[1], 1), 4)}))
