"""Golden fixtures (tests/golden/, made by tests/golden/make_golden.py from the oracle): the oracle must still reproduce them on CPU,
and the HIP path must reproduce them on the GPU without running the oracle."""
import importlib
import json
import os

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

synth = importlib.import_module("go-rio_amd.synth")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _apd_inputs():
    g = np.load(os.path.join(GOLD, "apd_pair_1500x1700.npz"))
    return g, synth.scan_pair(1500, 1700, seed=int(g["seed"]))


def test_oracle_reproduces_apd_golden(oracle_apd):
    g, (sx, sl, tx, tl, _) = _apd_inputs()
    p = oracle_apd.launch_params()
    cs, ct = oracle_apd.calculate_covariances(sx, p), oracle_apd.calculate_covariances(tx, p)
    err, H, b, corr, sqd, _ = oracle_apd.linearize(g["pose"], sx, sl, tx, tl, cs, ct, p)
    assert np.array_equal(corr, g["corr"]) and np.array_equal(sqd, g["sqd"])
    assert np.allclose(H, g["H"], rtol=1e-12) and np.allclose(b, g["b"], rtol=1e-11, atol=1e-9) and err == pytest.approx(float(g["error"]), rel=1e-12)
    r = oracle_apd.align(np.eye(4), sx, sl, tx, tl, cs, ct, p)
    assert np.array_equal(r["T"], g["T_final"]) and r["n_linearize"] == int(g["n_linearize"])


def test_oracle_reproduces_ugpm_golden():
    import oracle
    from oracle import ugpm

    oracle.build()
    gold = json.load(open(os.path.join(GOLD, "ugpm_c2_windows.json")))
    for name, g in gold.items():
        res, d = ugpm.preintegrate(synth.imu_window(seed=g["seed"], vel_hz=g["vel_hz"]))
        assert np.allclose(res[0]["delta_R"], g["delta_R"], atol=1e-12) and np.allclose(res[0]["delta_p"], g["delta_p"], atol=1e-12)
        assert d["nb_state"] == g["diag"]["nb_state"]


@pytest.mark.gpu
def test_gpu_matches_apd_golden(gpu, gorio, pose_err):
    g, (sx, sl, tx, tl, _) = _apd_inputs()
    a = gorio.ApdGicp(corr_dist_threshold=2.0, transformation_epsilon=0.1, keep_knn_indices=1)
    a.setInputTarget(tx, tl)
    a.setInputSource(sx, sl)
    err, H, b = a.linearize(g["pose"])
    corr, sqd = a.getCorrespondences()
    assert np.array_equal(corr, g["corr"]) and np.array_equal(sqd, g["sqd"])  # bit-exact indices
    assert np.abs(H - g["H"]).max() / np.abs(g["H"]).max() < 1e-9 and np.abs(b - g["b"]).max() / np.abs(g["b"]).max() < 1e-9
    assert np.array_equal(a.getKnnIndices(0)[:64], g["knn_src_first64"])
    r = a.align()
    te, re = pose_err(g["T_final"], r["T"])
    assert te < 1e-4 and re < 1e-4 and r["n_linearize"] == int(g["n_linearize"]) and r["converged"] == bool(g["converged"])


@pytest.mark.gpu
def test_gpu_matches_ugpm_golden(gpu, gorio):
    gold = json.load(open(os.path.join(GOLD, "ugpm_c2_windows.json")))
    for name, g in gold.items():
        m = gorio.ugpm_preint_batch([synth.imu_window(seed=g["seed"], vel_hz=g["vel_hz"])])[0][0]
        rot = np.linalg.norm(Rot.from_matrix(np.array(g["delta_R"]).T @ m["delta_R"]).as_rotvec())
        assert rot < 1e-4 and np.linalg.norm(m["delta_p"] - np.array(g["delta_p"])) < 1e-4
        assert np.allclose(m["cov"], g["cov"], rtol=1e-3, atol=1e-3 * np.abs(g["cov"]).max())


# ---------------------------------------------------------------- chunked pre-integration (f4, preint.h:1584-1702)

def _chunked_golden():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return mg.chunked_inputs(synth), json.load(open(os.path.join(GOLD, "ugpm_chunked.json")))


def test_oracle_reproduces_chunked_golden():
    import oracle
    from oracle import ugpm

    oracle.build()
    (win, quantum, q), gold = _chunked_golden()
    res, d = ugpm.preintegrate_chunked(win, quantum, infer_t=q)
    assert d["iters_rot"] == gold["diag"]["iters_rot"] and d["iters_vel"] == gold["diag"]["iters_vel"]
    for m, g in zip(res[0], gold["records"]):
        for k in g:
            assert np.allclose(m[k], g[k], rtol=1e-9, atol=1e-12), k


@pytest.mark.gpu
def test_gpu_matches_chunked_golden(gpu, gorio):
    (win, quantum, q), gold = _chunked_golden()
    res, d = gorio.ugpm_preint_batch([win], infer_t=[q], quantum=quantum, return_diag=True)
    assert d[0]["iters_rot"] == gold["diag"]["iters_rot"] and d[0]["iters_vel"] == gold["diag"]["iters_vel"]
    assert len(res[0]) == len(gold["records"]) == 5
    for m, g in zip(res[0], gold["records"]):
        rot = np.linalg.norm(Rot.from_matrix(np.array(g["delta_R"]).T @ m["delta_R"]).as_rotvec())
        assert rot < 1e-4 and np.linalg.norm(m["delta_p"] - np.array(g["delta_p"])) < 1e-4 and m["dt"] == pytest.approx(g["dt"], abs=1e-12)
        sg = np.sqrt(np.diag(np.array(g["cov"])))
        assert np.allclose(np.sqrt(np.diag(m["cov"])), sg, rtol=2e-3)


# ---------------------------------------------------------------- preprocessing (f3), submap assembly (f4), LPM output type (a8)

def _prep_golden():
    import importlib.util

    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    return np.load(os.path.join(GOLD, "prep_submap_lpm.npz")), mg.prep_inputs(synth)


def test_oracle_reproduces_prep_submap_lpm_golden(oracle_apd):
    """The restatements of DBSCAN_simple.h, pcl::RadiusOutlierRemoval, the REVE estimator, the submap assembly and the LPM integrator still
    give the frozen answers (labels, masks, voxel membership: exact; floating-point results: to rounding)."""
    import oracle
    from oracle import ugpm

    oracle.build()
    g, d = _prep_golden()
    lab, nc = oracle_apd.dbscan_labels(d["scan"])
    assert nc == int(g["dbscan_clusters"]) and np.array_equal(lab, g["dbscan_labels"])
    assert np.array_equal(oracle_apd.radius_outlier_mask(d["scan"], 2.0, 2), g["outlier_keep"])
    reve = oracle_apd.reve_estimate(d["targets"], d["samples"])
    assert reve["success"] == bool(g["reve_success"]) and reve["n_valid"] == int(g["reve_n_valid"]) and np.array_equal(reve["inlier"], g["reve_inlier"])
    assert np.allclose(reve["v_r"], g["reve_v"], rtol=1e-12) and np.allclose(reve["sigma_v_r"], g["reve_sigma"], rtol=1e-10)
    xs, ls = oracle_apd.submap_assemble(d["frames"], d["rel"], 0.4)
    assert np.array_equal(xs, g["submap_xyz"]) and np.array_equal(ls, g["submap_label"])
    lpm, _ = ugpm.preintegrate(d["win"], infer_t=d["q"], type=0, **d["lpm_kw"])
    for k, m in enumerate(lpm):
        assert np.allclose(m["delta_R"], g["lpm_delta_R"][k], atol=1e-13) and np.allclose(m["delta_p"], g["lpm_delta_p"][k], atol=1e-13)
        assert np.allclose(m["cov"], g["lpm_cov"][k], rtol=1e-10, atol=1e-18) and m["dt"] == pytest.approx(float(g["lpm_dt"][k]))


@pytest.mark.gpu
def test_gpu_matches_prep_submap_lpm_golden(gpu, gorio):
    """The HIP path against the same frozen answers, without running the oracle."""
    g, d = _prep_golden()
    lab, nc = gorio.prep.dbscan_labels(d["scan"])
    assert nc == int(g["dbscan_clusters"]) and np.array_equal(lab, g["dbscan_labels"])
    assert np.array_equal(gorio.prep.radius_outlier_mask(d["scan"], 2.0, 2), g["outlier_keep"])
    reve = gorio.prep.ego_velocity(d["targets"], d["samples"])
    assert reve["success"] == bool(g["reve_success"]) and reve["n_valid"] == int(g["reve_n_valid"]) and np.array_equal(reve["inlier"], g["reve_inlier"])
    assert np.allclose(reve["v_r"], g["reve_v"], rtol=1e-10, atol=1e-12) and np.allclose(reve["sigma_v_r"], g["reve_sigma"], rtol=1e-9, atol=1e-14)
    a = gorio.ApdGicp(corr_dist_threshold=2.0)
    n = a.setInputTargetSubmap(d["frames"], d["rel"], voxel_leaf=0.4)
    xg, lg = a.getTargetPoints()
    assert n == g["submap_xyz"].shape[0] and np.array_equal(xg, g["submap_xyz"]) and np.array_equal(lg, g["submap_label"])
    lpm = gorio.ugpm_preint_batch([d["win"]], infer_t=[d["q"]], type=gorio.ugpm.LPM, **d["lpm_kw"])[0]
    for k, m in enumerate(lpm):
        rot = np.linalg.norm(Rot.from_matrix(g["lpm_delta_R"][k].T @ m["delta_R"]).as_rotvec())
        assert rot < 1e-10 and np.linalg.norm(m["delta_p"] - g["lpm_delta_p"][k]) < 1e-10
        assert np.allclose(m["cov"], g["lpm_cov"][k], rtol=1e-8, atol=1e-9 * np.abs(g["lpm_cov"][k]).max())


def test_oracle_reproduces_sor_golden(oracle_apd):
    _, d = _prep_golden()
    g = np.load(os.path.join(GOLD, "prep_sor.npz"))
    keep, dist = oracle_apd.statistical_outlier_mask(d["scan"], 20, 1.0)
    assert np.array_equal(keep, g["keep_20_10"]) and np.array_equal(dist, g["dist_20"])
    assert np.array_equal(oracle_apd.statistical_outlier_mask(d["scan"], 30, 1.2)[0], g["keep_30_12"])


@pytest.mark.gpu
def test_gpu_matches_sor_golden(gpu, gorio):
    _, d = _prep_golden()
    g = np.load(os.path.join(GOLD, "prep_sor.npz"))
    keep, dist = gorio.prep.statistical_outlier_mask(d["scan"], 20, 1.0, return_distances=True)
    assert np.array_equal(keep, g["keep_20_10"]) and np.array_equal(dist, g["dist_20"])
    assert np.array_equal(gorio.prep.statistical_outlier_mask(d["scan"], 30, 1.2), g["keep_30_12"])

