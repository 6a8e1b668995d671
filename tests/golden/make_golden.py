"""Regenerates the golden fixtures in this directory from the CPU ORACLE on seeded synthetic inputs.

These vectors are NOT reference outputs: the reference ships no vectors for this path and cannot be compiled here (DESIGN.md
section 2).  They freeze the oracle's answers so that a later edit of the oracle (or of the synthetic generators) cannot drift
silently, and they give the GPU tests a data file to compare against that does not require running the O(n^2) oracle.

    python tests/golden/make_golden.py
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    synth = importlib.import_module("go-rio_amd.synth")
    import oracle
    from oracle import apd, ugpm

    oracle.build()
    # APD-GICP: 1500 x 1700 pair, shipped launch parameters, LM
    sx, sl, tx, tl, T = synth.scan_pair(1500, 1700, seed=424242)
    p = apd.launch_params()
    cs, ct = apd.calculate_covariances(sx, p), apd.calculate_covariances(tx, p)
    pose = np.eye(4)
    pose[:3, :3] = synth.rpy_to_matrix([0.1, -0.1, 1.0])
    pose[:3, 3] = [0.2, -0.05, 0.01]
    err, H, b, corr, sqd, _ = apd.linearize(pose, sx, sl, tx, tl, cs, ct, p)
    r = apd.align(np.eye(4), sx, sl, tx, tl, cs, ct, p)
    idx, _ = apd.knn_self(sx, 20)
    np.savez_compressed(os.path.join(HERE, "apd_pair_1500x1700.npz"), seed=424242, pose=pose, error=err, H=H, b=b, corr=corr, sqd=sqd,
                        knn_src_first64=idx[:64], cov_src_first64=cs[:64], T_final=r["T"], H_final=r["H"], converged=r["converged"],
                        nr_iterations=r["nr_iterations"], n_linearize=r["n_linearize"])
    # UGPM: C2 windows (200 Hz / 20 Hz ego-velocity)
    out = {}
    for name, hz in (("c2_200hz", 200.0), ("c2_20hz", 20.0)):
        win = synth.imu_window(seed=424243, vel_hz=hz)
        res, d = ugpm.preintegrate(win)
        m = res[0]
        out[name] = dict(seed=424243, vel_hz=hz, diag={k: (float(v) if not isinstance(v, int) else v) for k, v in d.items()},
                         **{k: np.asarray(v).tolist() for k, v in m.items()})
    json.dump(out, open(os.path.join(HERE, "ugpm_c2_windows.json"), "w"), indent=1)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
