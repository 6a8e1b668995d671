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
    make_prep_golden(synth, apd, ugpm)
    make_chunked_golden(synth, ugpm)
    make_sor_golden(synth, apd)
    print("golden fixtures written to", HERE)


def chunked_inputs(synth):
    """A 1.6 s request in chunks of 0.5 s (opt.quantum, preint.h:1584-1702): stamps in every chunk, one on a chunk boundary."""
    win = synth.imu_window(seed=424250, duration=1.6)
    s = win["start_t"]
    return win, 0.5, [s + 0.2, s + 0.5, s + 0.95, s + 1.2, win["end_t"]]


def make_chunked_golden(synth, ugpm):
    win, quantum, q = chunked_inputs(synth)
    res, d = ugpm.preintegrate_chunked(win, quantum, infer_t=q)
    out = dict(quantum=quantum, diag={k: (float(v) if not isinstance(v, int) else v) for k, v in d.items()},
               records=[{k: np.asarray(v).tolist() for k, v in m.items()} for m in res[0]])
    json.dump(out, open(os.path.join(HERE, "ugpm_chunked.json"), "w"), indent=1)


def make_sor_golden(synth, apd):
    """pcl::StatisticalOutlierRemoval (the nodelet's default filter) on the scan of prep_inputs(): mask and per-point mean distances."""
    d = prep_inputs(synth)
    keep, dist = apd.statistical_outlier_mask(d["scan"], 20, 1.0)
    keep30, _ = apd.statistical_outlier_mask(d["scan"], 30, 1.2)
    np.savez_compressed(os.path.join(HERE, "prep_sor.npz"), keep_20_10=keep, dist_20=dist, keep_30_12=keep30)


def prep_inputs(synth):
    """Seeded inputs of the preprocessing / submap / LPM fixtures (shared with tests/test_golden.py, which regenerates them)."""
    rng = np.random.default_rng(424244)
    scan, _ = synth.radar_scan(3000, seed=424244)
    # REVE targets: x y z intensity doppler, a moving platform with some movers and weak returns (test_prep_gpu._radar_targets)
    txyz, _ = synth.radar_scan(1500, seed=424245)
    r = np.linalg.norm(txyz, axis=1, keepdims=True)
    dop = -(txyz / r) @ np.array([4.1, 0.4, -0.05]) + rng.normal(0, 0.05, 1500)
    dop[:40] += rng.uniform(-4, 4, 40)
    targets = np.concatenate([txyz, rng.uniform(-5, 30, 1500)[:, None], dop[:, None]], axis=1).astype(np.float32)
    samples = rng.integers(0, 1000, (3, 5)).astype(np.uint32)
    # submap: four keyframes moved into the newest frame
    frames, rel = [], []
    odoms = [synth.gt_transform([0.6 * k, 0.05 * k, 0.01 * k], [0.1 * k, -0.05 * k, 1.2 * k]) for k in range(5)]
    for k in range(4):
        xyz, lab = synth.radar_scan(900 + 31 * k, seed=424250 + k, sensor_pose=odoms[k])
        frames.append((xyz, lab))
        rel.append(np.linalg.inv(odoms[4]) @ odoms[k])
    win = synth.imu_window(seed=424246, vel_hz=20.0)
    q = [win["start_t"], win["start_t"] + 0.35, win["end_t"]]
    return dict(scan=scan, targets=targets, samples=samples, frames=frames, rel=rel, win=win, q=q, lpm_kw=dict(gyr_bias=[0.002, -0.001, 0.003], vel_bias=[0.02, -0.01, 0.0]))


def make_prep_golden(synth, apd, ugpm):
    """Oracle outputs of the rows SURVEY 8f added (preprocessing, submap assembly) and of the LPM output type (a8), on small seeded inputs:
    nothing else guards those restatements against silent drift."""
    d = prep_inputs(synth)
    lab, nc = apd.dbscan_labels(d["scan"])
    keep = apd.radius_outlier_mask(d["scan"], 2.0, 2)
    reve = apd.reve_estimate(d["targets"], d["samples"])
    xs, ls = apd.submap_assemble(d["frames"], d["rel"], 0.4)
    lpm, _ = ugpm.preintegrate(d["win"], infer_t=d["q"], type=0, **d["lpm_kw"])
    np.savez_compressed(os.path.join(HERE, "prep_submap_lpm.npz"), dbscan_labels=lab, dbscan_clusters=nc, outlier_keep=keep,
                        reve_v=reve["v_r"], reve_sigma=reve["sigma_v_r"], reve_inlier=reve["inlier"], reve_n_valid=reve["n_valid"], reve_success=reve["success"],
                        submap_xyz=xs, submap_label=ls,
                        lpm_delta_R=np.stack([m["delta_R"] for m in lpm]), lpm_delta_p=np.stack([m["delta_p"] for m in lpm]), lpm_cov=np.stack([m["cov"] for m in lpm]),
                        lpm_dt=np.array([m["dt"] for m in lpm]))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "chunked":  # only the chunked-mode fixture (the others stay byte-identical)
    sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
    import importlib

    import oracle
    from oracle import ugpm as _u

    oracle.build()
    make_chunked_golden(importlib.import_module("go-rio_amd.synth"), _u)
    sys.exit(0)

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sor":  # only the statistical-outlier fixture
    import oracle
    from oracle import apd as _a

    oracle.build()
    make_sor_golden(importlib.import_module("go-rio_amd.synth"), _a)
    sys.exit(0)

if __name__ == "__main__":
    main()
