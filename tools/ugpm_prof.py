import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib, time
import numpy as np
gorio = importlib.import_module("go-rio_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
wins = [gorio.synth.imu_window(seed=100 + q) for q in range(n)]
gorio.ugpm_preint_batch(wins)
t = time.perf_counter(); gorio.ugpm_preint_batch(wins); dt = time.perf_counter() - t
s, c = gorio.ugpm_stage_times()
print("windows", n, "wall ms", dt * 1e3, "windows/s", n / dt)
print("stage s [lpm, gram+cross, corr, lm, infer]:", [round(x * 1e3, 3) for x in s], c)
