"""Randomised soak of the parity claims (development aid, not a test): random cloud sizes / seeds / batch sizes for the registration
(correspondences and k-NN lists bit-exact, final pose 1e-4) and random windows for the pre-integration (1e-4), GPU against the oracle.
usage: python tools/soak.py [seconds]"""
import importlib, sys, time
import numpy as np
from scipy.spatial.transform import Rotation as Rot
sys.path.insert(0, ".")
gorio = importlib.import_module("go-rio_amd"); synth = gorio.synth
import oracle
from oracle import apd as oapd, ugpm as ougpm
oracle.build()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(20251005)
t0 = time.time()
n_pairs = n_batches = n_windows = 0
while time.time() - t0 < budget * 0.6:
    count = int(rng.choice([1, 3, 8, 16, 24]))
    pairs = [synth.scan_pair(int(rng.integers(64, 5000)), int(rng.integers(64, 6000)), seed=int(rng.integers(1, 1 << 30))) for _ in range(count)]
    opt = int(rng.integers(0, 2))
    kw = dict(corr_dist_threshold=2.0, transformation_epsilon=0.05, optimizer=opt, search=int(rng.integers(0, 2)), keep_knn_indices=1)
    objs = []
    for pr in pairs:
        o = gorio.ApdGicp(**kw); o.setInputTarget(pr[2], pr[3]); o.setInputSource(pr[0], pr[1]); objs.append(o)
    res = gorio.align_batch(objs)
    for q in rng.choice(count, size=min(count, 3), replace=False):
        pr, o = pairs[q], objs[q]
        p = oapd.launch_params(optimizer=opt, transformation_epsilon=0.05)
        cs, ct = oapd.calculate_covariances(pr[0], p), oapd.calculate_covariances(pr[2], p)
        ro = oapd.align(np.eye(4), pr[0], pr[1], pr[2], pr[3], cs, ct, p)
        dT = np.linalg.inv(np.asarray(ro["T"], np.float64)) @ np.asarray(res[q]["T"], np.float64)
        te, re = np.linalg.norm(dT[:3, 3]), np.linalg.norm(Rot.from_matrix(dT[:3, :3]).as_rotvec())
        assert te < 1e-4 and re < 1e-4 and res[q]["n_linearize"] == ro["n_linearize"], (count, q, te, re, res[q]["n_linearize"], ro["n_linearize"])
        idx, _ = oapd.knn_self(pr[0], 20)
        assert np.array_equal(o.getKnnIndices(0), idx), ("knn", count, q)
        n_pairs += 1
    n_batches += 1
while time.time() - t0 < budget:
    count = int(rng.choice([1, 5, 16, 24]))
    wins = [synth.imu_window(seed=int(rng.integers(1, 1 << 30)), duration=float(rng.uniform(0.5, 1.8)), vel_hz=float(rng.choice([200.0, 100.0, 20.0]))) for _ in range(count)]
    res, diag = gorio.ugpm_preint_batch(wins, return_diag=True)
    for q in rng.choice(count, size=min(count, 2), replace=False):
        ro, do = ougpm.preintegrate(wins[q])
        a, b = res[q][0], ro[0]
        rot = np.linalg.norm(Rot.from_matrix(b["delta_R"].T @ a["delta_R"]).as_rotvec())
        assert rot < 1e-4 and np.linalg.norm(a["delta_p"] - b["delta_p"]) < 1e-4 and diag[q]["iters_rot"] == do["iters_rot"] and diag[q]["iters_vel"] == do["iters_vel"], (count, q, rot, diag[q], do)
        assert np.allclose(a["cov"], b["cov"], rtol=1e-3, atol=1e-3 * np.abs(b["cov"]).max())
        n_windows += 1
# chunked requests, LPM output type, preprocessing filters: a few random cases each
n_extra = 0
t1 = time.time()
while time.time() - t1 < min(40.0, 0.25 * budget):
    w = synth.imu_window(seed=int(rng.integers(1, 1 << 30)), duration=float(rng.uniform(1.0, 2.2)))
    quantum = float(rng.uniform(0.35, 0.9))
    q = sorted(float(w["start_t"] + rng.uniform(0.05, 1.0) * (w["end_t"] - w["start_t"])) for _ in range(int(rng.integers(1, 4)))) + [w["end_t"]]
    typ = int(rng.integers(0, 2))
    ro, _ = ougpm.preintegrate_chunked(w, quantum, infer_t=q, type=typ)
    rg = gorio.ugpm_preint_batch([w], infer_t=[q], quantum=quantum, type=typ)
    for a, b in zip(rg[0], ro[0]):
        rot = np.linalg.norm(Rot.from_matrix(b["delta_R"].T @ a["delta_R"]).as_rotvec())
        assert rot < 1e-4 and np.linalg.norm(a["delta_p"] - b["delta_p"]) < 1e-4 and abs(a["dt"] - b["dt"]) < 1e-12, ("chunked", typ, quantum, rot)
    xyz, _ = synth.radar_scan(int(rng.integers(200, 9000)), seed=int(rng.integers(1, 1 << 30)))
    mk = int(rng.integers(1, 32))
    if len(xyz) > mk:
        keep, dist = gorio.prep.statistical_outlier_mask(xyz, mk, float(rng.uniform(0.0, 2.0)), return_distances=True)
        ok, od = oapd.statistical_outlier_mask(xyz, mk, 1.0)
        assert np.array_equal(dist, od), ("sor", len(xyz), mk)
    r, mn = float(rng.uniform(0.5, 4.0)), int(rng.integers(1, 8))
    assert np.array_equal(gorio.prep.radius_outlier_mask(xyz, r, mn), oapd.radius_outlier_mask(xyz, r, mn)), ("radius", len(xyz), r, mn)
    n_extra += 1
print("extra cases (chunked request + statistical + radius filter each):", n_extra)
print("soak ok:", n_batches, "registration batches,", n_pairs, "pairs and", n_windows, "windows checked against the oracle in", round(time.time() - t0), "s")
