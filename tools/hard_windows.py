"""GPU against the oracle on windows with very fast rotation, with both solvers' traces (GORIO_UGPM_LMTRACE / UGPMO_LMTRACE)."""
import importlib, os, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
os.environ["GORIO_UGPM_LMTRACE"] = "1"; os.environ["UGPMO_LMTRACE"] = "1"
from scipy.spatial.transform import Rotation as Rot
gorio = importlib.import_module("go-rio_amd"); synth = gorio.synth
import oracle; from oracle import ugpm as u
oracle.build()
from test_ugpm_gpu import _spinning
for a, f, gv in [(10.0, 5.0, 1e-4), (20.0, 5.0, 1e-4)]:  # LPM-initialised cost already differs in the 7th / 3rd digit at equal x
    w = synth.imu_window(seed=320, duration=1.0, omega_fn=_spinning(a, f), gyr_var=gv)
    sys.stderr.write(f"==== amp {a} f {f}\n"); sys.stderr.flush()
    ro, do = u.preintegrate(w)
    rg, dg = gorio.ugpm_preint_batch([w], return_diag=True)
    A, B = rg[0][0], ro[0]
    rot = np.linalg.norm(Rot.from_matrix(B["delta_R"].T @ A["delta_R"]).as_rotvec())
    print(a, f, "iters", dg[0]["iters_rot"], do["iters_rot"], "cost", dg[0]["cost_rot"], do["cost_rot"], "rot", rot, "pos", np.linalg.norm(A["delta_p"] - B["delta_p"]),
          "cov rel", np.abs(A["cov"] - B["cov"]).max() / np.abs(B["cov"]).max(), flush=True)
