"""Work counters and per-phase cycles of nn_search_pruned_kernel from a library built with -DGORIO_STATS (given through GORIO_AMD_LIB):
what the exact pruned search really executes -- (query, tile) items = 32 distance evaluations each, rounds, tiles, fine tests -- per
launch of the workload's 20-iteration align.  With --merge the per-launch distance evaluations go into profiles/kernel_counters.json,
where bench.py turns them into the useful-work roofline (evaluations x 8 flop / live time / FP32 peak).

    make -C go-rio_amd/csrc OUT=../../tools/variants/stats.so EXTRA=-DGORIO_STATS
    GORIO_AMD_LIB=$PWD/tools/variants/stats.so python tools/search_work.py c3|c4|c5 [iterations] [--merge]"""
import json
import ctypes as C
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
gorio = importlib.import_module("go-rio_amd")
synth = gorio.synth
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
merge = "--merge" in sys.argv
shape = argv[0] if len(argv) > 0 else "c4"
iters = int(argv[1]) if len(argv) > 1 else 20
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
elif shape == "c3":
    sx, sl = synth.radar_scan(16384, seed=synth.BASE_SEED + 3)
    tx, tl = synth.local_map(100000, seed=synth.BASE_SEED + 4)
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
    last = dict(launches=cn[1], items=v[3], rounds=v[4], waves=v[5], tiles=v[0], fine_tests=v[7], flushes=v[6], nn_ms_per_launch_instrumented=1e3 * st[1] / max(cn[1], 1))
if merge:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    path = os.path.join(root, "profiles", "kernel_counters.json")
    try:
        doc = json.load(open(path))
    except Exception:
        doc = {}
    sha = bench.source_sha16()
    if doc.get("_source_sha16") != sha:
        doc = {"_source_sha16": sha, "_note": "", "kernels": {}}
    e = doc.setdefault("kernels", {}).setdefault(f"nn_search_pruned_kernel:{shape}", {})
    per = last["launches"]
    e["distance_evaluations"] = 32.0 * last["items"] / per
    e["search_items"] = last["items"] / per
    e["search_rounds"] = last["rounds"] / per
    e["search_work_units"] = last["waves"] / per  # workgroup-waves per launch (query waves x parts of the plan)
    e["search_tiles_needed"] = last["tiles"] / per
    e["search_fine_tests"] = last["fine_tests"] / per
    json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
    print("merged into", path, {k: round(x, 1) for k, x in e.items() if k.startswith(("distance", "search"))})
