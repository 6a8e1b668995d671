import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, sys, time
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.set_device(0); x = torch.zeros(10, device="cuda")
gorio = importlib.import_module("go-rio_amd")
import cProfile, pstats
wins = [gorio.synth.imu_window(seed=100 + q) for q in range(64)]
gorio.ugpm_preint_batch(wins)
t = time.perf_counter(); gorio.ugpm_preint_batch(wins); print("wall ms", (time.perf_counter() - t) * 1e3)
pr = cProfile.Profile(); pr.enable(); gorio.ugpm_preint_batch(wins); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
