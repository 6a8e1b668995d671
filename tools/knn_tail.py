"""Where the slow waves of knn_kth_kernel spend their time, on the C3 clouds (16 k scan, 100 k local map), with a -DGORIO_STATS build whose
knn_kth_kernel carries extra counters (development aid; that instrumentation is not in the tree, see profiles/r03/experiments.md):
rows [0..5] waves of >= 262144 cycles, [6..11] the others: waves, evaluated tiles, executed insertions, tiles passing the coarse test,
tiles taken candidate-per-lane, cycles; [12..15] maxima over 1024 wave classes (summed by the debug call)."""
import ctypes as C, importlib, sys
sys.path.insert(0, ".")
gorio = importlib.import_module("go-rio_amd"); synth = gorio.synth
lib = gorio.load_library()
out = (C.c_ulonglong * 24)()
for name, (x, l) in (("scan 16384", synth.radar_scan(16384, seed=synth.BASE_SEED + 2)), ("local map 100000", synth.local_map(100000, seed=synth.BASE_SEED + 3))):
    o = gorio.ApdGicp(corr_dist_threshold=2.0, search=1)
    o.setInputTarget(x, l)
    o.setInputSource(x[:64], l[:64])
    lib.gorio_debug_search_stats(out, 1)
    o.calculateCovariances()
    lib.gorio_debug_search_stats(out, 1)
    v = [int(t) for t in out]
    for tag, o0 in (("slow waves (>= 256 k cycles)", 0), ("other waves", 6)):
        w = max(1, v[o0])
        print(name, tag, ": waves", v[o0], "tiles", round(v[o0 + 1] / w, 1), "insertions", round(v[o0 + 2] / w, 1), "coarse-passing tiles", round(v[o0 + 3] / w, 1),
              "candidate-per-lane tiles", round(v[o0 + 4] / w, 1), "cycles", v[o0 + 5] // w, flush=True)
