"""Preprocessing that feeds the hot path: the DBSCAN cluster labels written to normal_x
(preprocessing_nodelet_ntu.cpp:518-568, DBSCAN_simple.h) -- GPU radius searches + the reference's queue, against the CPU restatement.
Labels are small integers stored as floats: the comparison is exact."""
import importlib
import os

import numpy as np
import pytest

synth = importlib.import_module("go-rio_amd.synth")


def _blobs(seed, n_blobs=9, per=120, spread=0.35, noise=150):
    rng = np.random.default_rng(seed)
    pts = []
    for b in range(n_blobs):
        c = np.array([rng.uniform(3, 90), rng.uniform(-30, 30), rng.uniform(-1, 3)])
        pts.append(c + rng.normal(0, spread * (1 + c[0] / 60), (per + 15 * b, 3)))
    pts.append(np.stack([rng.uniform(1, 100, noise), rng.uniform(-40, 40, noise), rng.uniform(-2, 6, noise)], axis=1))
    xyz = np.concatenate(pts).astype(np.float32)
    return xyz[rng.permutation(len(xyz))]  # DBSCAN_simple's result depends on the point order: shuffle it


def test_oracle_dbscan_properties(oracle_apd):
    """CPU: separated blobs get one label each, ranked by centroid distance (label 1 = nearest cluster); sparse noise gets 0; clusters
    below the size window are dropped."""
    xyz = _blobs(1)
    lab, nc = oracle_apd.dbscan_labels(xyz)
    assert nc >= 7 and lab.max() == nc and (lab == 0).sum() > 50
    cent = [np.linalg.norm(xyz[lab == k].mean(axis=0)) for k in range(1, nc + 1)]
    assert np.all(np.diff(cent) > 0)  # rank order = distance order (PREP:555-568)
    sizes = np.bincount(lab.astype(int))[1:]
    assert sizes.min() >= 20  # setMinClusterSize(20)
    lab2, nc2 = oracle_apd.dbscan_labels(xyz, min_cluster=200)
    assert nc2 < nc


def _float_radius_case(eps=0.9):
    """A cloud whose clustering depends on HOW the expansion radius of DBSCAN_simple.h:65-67 is evaluated: (std::sqrt(float) - 1) / 100 in
    float and only `+ eps_` in double (the reference) against the whole expression in double.  Returns (xyz, index of the probe point,
    whether the reference's arithmetic claims it).  A blob of 30 points sits behind the expanded point c = (N, 0, 0); the probe t = (N, y, 0)
    lies at a float squared distance y * y equal to the SMALLER of the two candidate squared radii (they differ by one ulp), so exactly
    the evaluation with the larger radius claims it (the search is `d < r2`); the seed radius of DBS:36-39 never reaches it."""
    f32 = np.float32
    for k in range(2000):
        N = f32(7.0) + f32(k) * f32(0.00390625)
        ef = f32(f32(N - f32(1.0)) / f32(100.0))
        r_f = float(ef) + eps
        r_d = (float(N) - 1.0) / 100.0 + eps
        r2f, r2d = f32(r_f * r_f), f32(r_d * r_d)
        if r2f == r2d:
            continue
        target = min(r2f, r2d)
        y = f32(np.sqrt(float(target)))
        for dy in (0, 1, -1, 2, -2):
            yy = np.nextafter(y, f32(np.inf if dy > 0 else -np.inf)) if abs(dy) == 1 else y
            if abs(dy) == 2:
                yy = np.nextafter(np.nextafter(y, f32(np.inf if dy > 0 else -np.inf)), f32(np.inf if dy > 0 else -np.inf))
            if f32(yy * yy) == target:
                rng = np.random.default_rng(12)
                blob = np.stack([np.full(30, float(N)) + rng.uniform(-0.05, 0.05, 30), rng.uniform(-0.30, -0.15, 30), rng.uniform(-0.05, 0.05, 30)], axis=1)
                xyz = np.concatenate([blob, [[float(N), 0.0, 0.0]], [[float(N), float(yy), 0.0]]]).astype(np.float32)
                return xyz, len(xyz) - 1, bool(r2f > r2d)
    raise AssertionError("no norm with differing float / double radii found")


def test_oracle_dbscan_expansion_radius_is_float(oracle_apd):
    """DBS:65-67 evaluates (sqrt - 1) / 100 in float: the probe point is claimed exactly when the float evaluation says so."""
    xyz, probe, claimed = _float_radius_case()
    lab, nc = oracle_apd.dbscan_labels(xyz)
    assert nc == 1 and np.all(lab[:-1] == 1)
    assert (lab[probe] == 1) == claimed


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["blobs", "blobs2", "radar4k", "radar16k", "tiny", "float_radius"])
def test_dbscan_labels_match_oracle(gpu, gorio, oracle_apd, case):
    if case == "blobs":
        xyz = _blobs(2)
    elif case == "blobs2":
        xyz = _blobs(3, n_blobs=14, per=60, spread=0.6, noise=400)  # touching clusters: shared border points, order dependence
    elif case == "radar4k":
        xyz, _ = synth.radar_scan(4000, seed=610)
    elif case == "radar16k":
        xyz, _ = synth.radar_scan(16384, seed=611)
    elif case == "float_radius":
        xyz = _float_radius_case()[0]
    else:
        xyz = _blobs(4, n_blobs=1, per=25, noise=5)
    lab_o, nc_o = oracle_apd.dbscan_labels(xyz)
    lab_g, nc_g = gorio.prep.dbscan_labels(xyz)
    assert nc_g == nc_o
    assert np.array_equal(lab_g, lab_o)


@pytest.mark.gpu
def test_dbscan_labels_on_real_scan_and_other_parameters(gpu, gorio, oracle_apd):
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_lidar_pair.npz"))
    xyz = g["a_0"][:, :3].copy()
    for kw in (dict(), dict(eps=0.4, min_pts=5, min_cluster=10), dict(eps=1.5, min_pts=30, max_cluster=800)):
        lab_o, nc_o = oracle_apd.dbscan_labels(xyz, **kw)
        lab_g, nc_g = gorio.prep.dbscan_labels(xyz, eps=kw.get("eps", 0.9), core_min_pts=kw.get("min_pts", 10), min_cluster_size=kw.get("min_cluster", 20),
                                               max_cluster_size=kw.get("max_cluster", 25000))
        assert nc_g == nc_o and np.array_equal(lab_g, lab_o), kw


@pytest.mark.gpu
def test_labels_feed_the_registration(gpu, gorio):
    """The labels are what APD:271-273 compares: a pair labelled on the GPU registers like the same pair labelled by the generator's
    object ids would (same pose to 1e-3: the cluster weight is a 1/N term of the LM acceptance error only)."""
    sx, _, tx, _, T = synth.scan_pair(6000, 6000, seed=620)
    ls, _ = gorio.prep.dbscan_labels(sx)
    lt, _ = gorio.prep.dbscan_labels(tx)
    a = gorio.ApdGicp(corr_dist_threshold=2.0, transformation_epsilon=0.01)
    a.setInputTarget(tx, lt)
    a.setInputSource(sx, ls)
    b = gorio.ApdGicp(corr_dist_threshold=2.0, transformation_epsilon=0.01)
    b.setInputTarget(tx, None)
    b.setInputSource(sx, None)
    ra, rb = a.align(), b.align()
    assert ra["converged"] and np.allclose(ra["T"], rb["T"], atol=1e-3)


# ---------------------------------------------------------------------------------------------- REVE Doppler ego-velocity

def _radar_targets(seed, n=4000, v_true=(5.2, -0.3, 0.1), noise=0.05, movers=200):
    rng = np.random.default_rng(seed)
    xyz, _ = synth.radar_scan(n, seed=700 + seed)
    r = np.linalg.norm(xyz, axis=1, keepdims=True)
    dop = -(xyz / r) @ np.asarray(v_true) + rng.normal(0, noise, n)  # the estimator negates the measured Doppler (REVE:87)
    dop[:movers] += rng.uniform(-4, 4, movers)  # moving objects
    inten = rng.uniform(-5, 30, n)  # some below min_db
    t = np.concatenate([xyz, inten[:, None], dop[:, None]], axis=1).astype(np.float32)
    return t, rng


def test_oracle_reve_recovers_the_ego_velocity(oracle_apd):
    t, rng = _radar_targets(1)
    res0 = oracle_apd.reve_estimate(t, np.zeros((0, 5), np.uint32))
    samples = rng.integers(0, res0["n_valid"], (3, 5))
    res = oracle_apd.reve_estimate(t, samples)
    assert res["success"] and not res["zero_velocity"]
    assert np.allclose(res["v_r"], [5.2, -0.3, 0.1], atol=0.02) and np.all(res["sigma_v_r"] < 0.05)
    assert res["inlier"].sum() > 0.8 * res["n_valid"] and res["n_valid"] < len(t)  # the intensity / angle gates removed some targets
    still, _ = _radar_targets(2, v_true=(0, 0, 0), noise=0.01, movers=50)
    z = oracle_apd.reve_estimate(still, samples)
    assert z["success"] and z["zero_velocity"] and np.all(z["v_r"] == 0) and np.allclose(z["sigma_v_r"], [1e-3, 3.2e-3, 1e-2], rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["moving", "standstill", "many_outliers", "no_ransac", "few_targets"])
def test_reve_ego_velocity_matches_oracle(gpu, gorio, oracle_apd, case):
    kw, ok = {}, {}
    if case == "moving":
        t, rng = _radar_targets(3)
    elif case == "standstill":
        t, rng = _radar_targets(4, v_true=(0, 0, 0), noise=0.01, movers=30)
    elif case == "many_outliers":
        t, rng = _radar_targets(5, movers=1500)  # > 5 % outliers: the reference then counts every target as an inlier (REVE:215-220)
    elif case == "no_ransac":
        t, rng = _radar_targets(6)
        kw = dict(use_ransac=0)
    else:
        t, rng = _radar_targets(7, n=40, movers=3)
    cfg_o, cfg_g = oracle_apd.reve_default_config(**kw), gorio.prep.reve_default_config(**kw)
    assert gorio.prep.reve_ransac_iterations(cfg_g) == 3  # uint(log(0.005) / log(1 - 0.95^5)), REVEH:138-141
    nv = gorio.prep.ego_velocity(t, [], cfg_g)["n_valid"]  # n_iter = 0: the gates only
    assert nv == oracle_apd.reve_estimate(t, np.zeros((0, 5), np.uint32), cfg_o)["n_valid"]
    samples = rng.integers(0, max(nv, 1), (3, 5)).astype(np.uint32)
    ro = oracle_apd.reve_estimate(t, samples, cfg_o)
    rg = gorio.prep.ego_velocity(t, samples, cfg_g)
    assert rg["success"] == ro["success"] and rg["zero_velocity"] == ro["zero_velocity"] and rg["n_valid"] == ro["n_valid"]
    assert np.array_equal(rg["inlier"], ro["inlier"]) and np.array_equal(rg["outlier"], ro["outlier"])  # the published clouds: exact
    assert np.allclose(rg["v_r"], ro["v_r"], rtol=1e-10, atol=1e-12) and np.allclose(rg["sigma_v_r"], ro["sigma_v_r"], rtol=1e-9, atol=1e-14)
    if case == "many_outliers":
        assert ro["inlier"].sum() == ro["n_valid"] and ro["outlier"].sum() == 0
    if case == "moving":
        assert np.allclose(rg["v_r"], [5.2, -0.3, 0.1], atol=0.02)


# ------------------------------------------------------------------------------------------------ radius outlier removal

def test_oracle_radius_outlier_mask_against_kdtree_counts(oracle_apd):
    """CPU: the restatement against an independent neighbour count (scipy cKDTree on the float coordinates; pairs within 2e-5 m of the
    radius are excluded from the comparison: the tree measures in double, the restatement on FLANN's float distances)."""
    from scipy.spatial import cKDTree

    xyz, _ = synth.radar_scan(6000, seed=synth.BASE_SEED + 61)
    tree = cKDTree(xyz.astype(np.float64))
    for radius, min_pts in ((2.0, 2), (2.0, 5), (0.8, 1)):
        keep = oracle_apd.radius_outlier_mask(xyz, radius, min_pts)
        lo = np.array([len(v) for v in tree.query_ball_point(xyz.astype(np.float64), radius - 2e-5)])
        hi = np.array([len(v) for v in tree.query_ball_point(xyz.astype(np.float64), radius + 2e-5)])
        sure = lo == hi
        assert sure.mean() > 0.95
        assert np.array_equal(keep[sure], lo[sure] > min_pts)
        assert 0 < keep.sum() < len(xyz)


@pytest.mark.gpu
@pytest.mark.parametrize("radius,min_pts", [(2.0, 2), (2.0, 5), (0.8, 1), (5.0, 40)])
def test_radius_outlier_mask_matches_oracle(gpu, gorio, oracle_apd, radius, min_pts):
    """pcl::RadiusOutlierRemoval with the launch files' parameters (radius 2, 1 - 5 neighbours) and two others: the GPU mask equals the
    CPU restatement exactly (both count float L2_Simple squared distances <= radius^2)."""
    for seed, n in ((62, 16384), (63, 3000), (64, 257)):
        xyz, _ = synth.radar_scan(n, seed=synth.BASE_SEED + seed)
        keep = gorio.prep.radius_outlier_mask(xyz, radius, min_pts)
        assert np.array_equal(keep, oracle_apd.radius_outlier_mask(xyz, radius, min_pts))


@pytest.mark.gpu
def test_radius_outlier_mask_counts_duplicates_and_isolated_points(gpu, gorio, oracle_apd):
    xyz = np.array([[0, 0, 0], [0, 0, 0], [0, 0, 0], [10, 0, 0], [10.5, 0, 0], [50, 50, 5]], np.float32)
    xyz = np.concatenate([xyz, np.array([[100 + 0.1 * i, -20, 1] for i in range(40)], np.float32)])
    keep = gorio.prep.radius_outlier_mask(xyz, 1.0, 2)
    assert np.array_equal(keep, oracle_apd.radius_outlier_mask(xyz, 1.0, 2))
    assert keep[:3].all() and not keep[3] and not keep[4] and not keep[5]  # three coincident points count each other; a pair is one short


# ------------------------------------------------------------------------------------------------ statistical outlier removal (the nodelet's default filter)

def test_oracle_statistical_outlier_mask_against_kdtree(oracle_apd):
    """CPU: the restatement against an independent computation (scipy cKDTree in double on the float coordinates): per-point mean
    neighbour distances to 1e-5, and the same mask wherever a point is not within 1e-5 of the threshold."""
    from scipy.spatial import cKDTree

    xyz, _ = synth.radar_scan(6000, seed=synth.BASE_SEED + 65)
    tree = cKDTree(xyz.astype(np.float64))
    for mean_k, mul in ((20, 1.0), (30, 1.2), (5, 0.5)):
        keep, dist = oracle_apd.statistical_outlier_mask(xyz, mean_k, mul)
        dd, _ = tree.query(xyz.astype(np.float64), k=mean_k + 1)
        ref = dd[:, 1:].mean(axis=1)
        assert np.abs(ref - dist).max() < 1e-5 * max(1.0, ref.max())
        thr = ref.mean() + mul * ref.std(ddof=1)
        sure = np.abs(ref - thr) > 1e-5 * max(1.0, thr)
        assert sure.mean() > 0.99 and np.array_equal(keep[sure], (ref <= thr)[sure])
        assert 0 < keep.sum() < len(xyz)
    with pytest.raises(ValueError):
        oracle_apd.statistical_outlier_mask(xyz[:10], 20, 1.0)  # fewer points than mean_k + 1


@pytest.mark.gpu
@pytest.mark.parametrize("mean_k,mul", [(20, 1.0), (30, 1.2), (1, 0.0), (31, 2.0)])
def test_statistical_outlier_mask_matches_oracle(gpu, gorio, oracle_apd, mean_k, mul):
    """pcl::StatisticalOutlierRemoval with the nodelet's defaults (20, 1.0), the launch files' values (30, 1.2) and the ends of the
    supported range: per-point mean neighbour distances and the mask equal the CPU restatement exactly (the same float squared
    distances, sorted, double square roots summed in the same order, the statistics in PCL's order on the host)."""
    for seed, n in ((66, 16384), (67, 3000), (68, 257)):
        xyz, _ = synth.radar_scan(n, seed=synth.BASE_SEED + seed)
        keep, dist = gorio.prep.statistical_outlier_mask(xyz, mean_k, mul, return_distances=True)
        okeep, odist = oracle_apd.statistical_outlier_mask(xyz, mean_k, mul)
        assert np.array_equal(dist, odist)
        assert np.array_equal(keep, okeep)


@pytest.mark.gpu
def test_statistical_outlier_mask_duplicates_small_clouds_and_errors(gpu, gorio, oracle_apd):
    rng = np.random.default_rng(9)
    blob = rng.normal(0.0, 0.3, (60, 3)).astype(np.float32)
    xyz = np.concatenate([blob, blob[:7], np.array([[30, 30, 3], [-40, 5, 0]], np.float32)])  # duplicates (zero distances beside the query's own) and two far points
    keep, dist = gorio.prep.statistical_outlier_mask(xyz, 8, 1.0, return_distances=True)
    okeep, odist = oracle_apd.statistical_outlier_mask(xyz, 8, 1.0)
    assert np.array_equal(dist, odist) and np.array_equal(keep, okeep)
    assert not keep[-1] and not keep[-2] and keep[:60].all()
    tiny = xyz[:9]
    k2, d2 = gorio.prep.statistical_outlier_mask(tiny, 8, 1.0, return_distances=True)  # n = mean_k + 1: every other point is a neighbour
    o2, od2 = oracle_apd.statistical_outlier_mask(tiny, 8, 1.0)
    assert np.array_equal(d2, od2) and np.array_equal(k2, o2)
    with pytest.raises(gorio.GorioError):
        gorio.prep.statistical_outlier_mask(tiny, 9, 1.0)  # fewer points than mean_k + 1
    with pytest.raises(gorio.GorioError):
        gorio.prep.statistical_outlier_mask(xyz, 32, 1.0)  # mean_k + 1 > 32 list slots


# ------------------------------------------------------------------------------------------------ voxel-grid downsampling

@pytest.mark.gpu
@pytest.mark.parametrize("leaf", [0.1, 0.5, 2.0])
def test_voxel_downsample_matches_oracle(gpu, gorio, oracle_apd, leaf):
    """pcl::VoxelGrid of one scan (the preprocessing nodelet's downsample step, leaf 0.1 in the launch files): the device voxel grid
    against the CPU restatement of PCL's centroid rule -- same voxels, same order, centroids to float rounding."""
    xyz, _ = synth.radar_scan(16384, seed=synth.BASE_SEED + 71)
    out = gorio.prep.voxel_downsample(xyz, leaf)
    ref, _ = oracle_apd.submap_assemble([(xyz, np.zeros(len(xyz), np.float32))], [np.eye(4)], leaf)
    assert out.shape == ref.shape and out.shape[0] < len(xyz)
    assert np.abs(out - ref).max() < 1e-6 * max(1.0, np.abs(ref).max())


# ------------------------------------------------------------------------------------------------ the chain that feeds the registration

@pytest.mark.gpu
def test_preprocessing_chain_then_registration_matches_oracle(gpu, gorio, oracle_apd, pose_err):
    """What the preprocessing nodelet does to a scan before the registration sees it (PREP:503-568: VoxelGrid -> RadiusOutlierRemoval
    -> DBSCAN labels into normal_x), on the GPU and through the CPU restatements, for two scans of one scene; then APD-GICP on the two
    results (the labels feed its cluster weight, APD:271-273).  Every stage must agree: same voxels, same mask, same labels, and the
    registration within 1e-4 of the oracle's on the GPU-made inputs."""
    sx, _, tx, _, _ = synth.scan_pair(16384, 16384, seed=synth.BASE_SEED + 81)
    made = []
    for raw in (sx, tx):
        v_g = gorio.prep.voxel_downsample(raw, 0.3)
        v_o, _ = oracle_apd.submap_assemble([(raw, np.zeros(len(raw), np.float32))], [np.eye(4)], 0.3)
        assert v_g.shape == v_o.shape and np.abs(v_g - v_o).max() < 1e-5
        k_g = gorio.prep.radius_outlier_mask(v_g, 2.0, 2)
        assert np.array_equal(k_g, oracle_apd.radius_outlier_mask(v_g, 2.0, 2))
        pts = np.ascontiguousarray(v_g[k_g])
        l_g, nc = gorio.prep.dbscan_labels(pts)
        l_o, nc_o = oracle_apd.dbscan_labels(pts)
        assert nc == nc_o and np.array_equal(l_g, l_o)
        made.append((pts, l_g))
    (a, la), (b, lb) = made
    assert 3000 < len(a) < 16384 and la.max() >= 1
    p = oracle_apd.launch_params()
    ca, cb = oracle_apd.calculate_covariances(a, p), oracle_apd.calculate_covariances(b, p)
    ro = oracle_apd.align(np.eye(4), a, la, b, lb, ca, cb, p)
    g = gorio.ApdGicp(corr_dist_threshold=2.0, transformation_epsilon=0.1, search=1)
    g.setInputTarget(b, lb)
    g.setInputSource(a, la)
    r = g.align()
    te, re = pose_err(ro["T"], r["T"])
    assert te < 1e-4 and re < 1e-4 and r["converged"] == ro["converged"] and r["n_linearize"] == ro["n_linearize"]


# ------------------------------------------------------------------------------------------------ ego-velocity -> pre-integration

@pytest.mark.gpu
def test_ego_velocity_samples_feed_the_preintegration(gpu, gorio, oracle_apd):
    """The other feeder of the hot path: every radar scan's Doppler ego-velocity becomes one `vel` sample of the pre-integration window
    (radar_graph_slam_nodelet.cpp:274-280, 481-495).  A 20 Hz train of synthetic scans along the IMU generator's velocity profile is
    estimated on the GPU and by the CPU restatement with the same RANSAC draws; the GPU-made samples then go through UGPM on both
    sides.  Every estimate agrees to 1e-10, the pre-integrated window to the usual gates."""
    win = synth.imu_window(seed=synth.BASE_SEED + 91, vel_hz=20.0)
    cfg_g, cfg_o = gorio.prep.reve_default_config(), oracle_apd.reve_default_config()
    vel = []
    for k, t_k in enumerate(win["vel_t"]):
        v_true = synth.vel_true(np.array([t_k]))[0]
        targets, rng = _radar_targets(100 + k, n=1500, v_true=tuple(v_true), noise=0.03, movers=40)
        nv = gorio.prep.ego_velocity(targets, [], cfg_g)["n_valid"]
        samples = rng.integers(0, max(nv, 1), (3, 5)).astype(np.uint32)
        rg, ro = gorio.prep.ego_velocity(targets, samples, cfg_g), oracle_apd.reve_estimate(targets, samples, cfg_o)
        assert rg["success"] and ro["success"] and np.allclose(rg["v_r"], ro["v_r"], rtol=1e-10, atol=1e-12)
        assert abs(rg["v_r"][0] - v_true[0]) < 0.05  # forward speed; lateral / vertical speed are weakly observable in this field of view
        vel.append(rg["v_r"])
    import oracle
    from oracle import ugpm as oracle_ugpm

    oracle.build()
    win = dict(win, vel=np.ascontiguousarray(np.stack(vel)))
    mg = gorio.ugpm_preint_batch([win], state_freq=20.0)[0][0]
    mo = oracle_ugpm.preintegrate(win, state_freq=20.0)[0][0]
    dR = mo["delta_R"].T @ mg["delta_R"]
    ang = float(np.arccos(np.clip((np.trace(dR) - 1.0) / 2.0, -1.0, 1.0)))
    assert ang < 1e-6 and np.linalg.norm(mg["delta_p"] - mo["delta_p"]) < 1e-6
