"""GPU parity tests of the APD-GICP path: HIP (through the C ABI) vs the CPU oracle on identical seeded inputs.

Gates: correspondence and k-NN indices BIT-EXACT; H, b, error relative error <= 1e-9; final transform within
1e-4 m / 1e-4 rad; plus size-independent properties at the BASELINE sizes (16k x 16k).
"""
import importlib

import numpy as np
import pytest

synth = importlib.import_module("go-rio_amd.synth")
pytestmark = pytest.mark.gpu

H_RTOL = 1e-9


def rel(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300))


def make(gorio, sx, sl, tx, tl, **params):
    params.setdefault("keep_knn_indices", 1)  # the parity hook behind getKnnIndices (off by default: 80 B of stores per point)
    g = gorio.ApdGicp(**params)
    g.setInputTarget(tx, tl)
    g.setInputSource(sx, sl)
    return g


def _pose():
    T = np.eye(4)
    T[:3, :3] = synth.rpy_to_matrix([0.1, -0.1, 1.0])
    T[:3, 3] = [0.2, -0.05, 0.01]
    return T


@pytest.fixture(scope="module")
def pair2k():
    return synth.scan_pair(2000, 2300, seed=21)  # ragged sizes: not multiples of 16 / 64 / 256


@pytest.mark.parametrize("n", [20, 21, 255, 257, 1000, 2300])
def test_knn_indices_bit_exact(gpu, gorio, oracle_apd, n):
    xyz, lab = synth.radar_scan(n, seed=100 + n)
    g = make(gorio, xyz, lab, xyz, lab)
    g.calculateCovariances()
    idx = g.getKnnIndices(0)
    idx_o, _ = oracle_apd.knn_self(xyz, 20)
    assert np.array_equal(idx, idx_o)


@pytest.mark.parametrize("search", [0, 1])
@pytest.mark.parametrize("k", [5, 20, 27, 32])
def test_knn_other_k(gpu, gorio, oracle_apd, k, search):
    xyz, lab = synth.radar_scan(700, seed=7)
    g = make(gorio, xyz, lab, xyz, lab, k_correspondences=k, search=search)
    g.calculateCovariances()
    idx_o, _ = oracle_apd.knn_self(xyz, k)
    assert np.array_equal(g.getKnnIndices(1), idx_o)
    cov_o = oracle_apd.covariances_from_knn(xyz, idx_o, oracle_apd.REG_PLANE)
    assert np.allclose(g.getTargetCovariances(), cov_o, rtol=0, atol=1e-9)


def test_knn_with_duplicate_points_ties_lowest_index(gpu, gorio, oracle_apd):
    """Exact duplicates and a regular lattice produce many equal distances: ties must go to the lowest index."""
    gx, gy = np.meshgrid(np.arange(12, dtype=np.float32), np.arange(12, dtype=np.float32))
    lattice = np.stack([gx.ravel(), gy.ravel(), np.zeros(144, np.float32)], axis=1)
    xyz = np.concatenate([lattice, lattice[:40]])  # 40 exact duplicates
    g = make(gorio, xyz, None, xyz, None, regularization=0)
    g.calculateCovariances()
    idx_o, _ = oracle_apd.knn_self(xyz, 20)
    assert np.array_equal(g.getKnnIndices(0), idx_o)


@pytest.mark.parametrize("reg", ["PLANE", "NONE", "MIN_EIG", "NORMALIZED_MIN_EIG", "FROBENIUS"])
def test_covariances_match_oracle(gpu, gorio, oracle_apd, pair2k, reg):
    sx, sl = pair2k[0], pair2k[1]
    code = getattr(oracle_apd, "REG_" + reg)
    g = make(gorio, sx, sl, sx, sl, regularization=code)
    g.calculateCovariances()
    idx_o, _ = oracle_apd.knn_self(sx, 20)
    cov_o = oracle_apd.covariances_from_knn(sx, idx_o, code)
    cov = g.getSourceCovariances()
    assert cov.shape == cov_o.shape
    if reg in ("NONE", "FROBENIUS"):
        assert np.allclose(cov, cov_o, rtol=1e-12, atol=1e-15)
    else:
        # eigenvector based: error scales with 1 / (relative eigen gap of the raw covariance)
        raw = oracle_apd.covariances_from_knn(sx, idx_o, oracle_apd.REG_NONE)[:, :3, :3]
        w = np.linalg.eigvalsh(raw)
        gap = np.minimum(w[:, 1] - w[:, 0], w[:, 2] - w[:, 1]) / np.maximum(w[:, 2], 1e-300)
        err = np.abs(cov - cov_o).reshape(len(cov), -1).max(axis=1)
        assert np.all(err <= 1e-13 / np.maximum(gap, 1e-12) + 1e-12), float(err.max())
        assert np.median(err) < 1e-13


def test_linearize_matches_oracle(gpu, gorio, oracle_apd, pair2k):
    sx, sl, tx, tl, _ = pair2k
    p = oracle_apd.launch_params()
    cs = oracle_apd.calculate_covariances(sx, p)
    ct = oracle_apd.calculate_covariances(tx, p)
    g = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0, transformation_epsilon=0.1)
    T = _pose()
    err, H, b = g.linearize(T)
    err_o, H_o, b_o, corr_o, sqd_o, maha_o = oracle_apd.linearize(T, sx, sl, tx, tl, cs, ct, p)
    corr, sqd = g.getCorrespondences()
    assert np.array_equal(corr, corr_o)  # bit-exact indices, including the -1 rejections
    assert np.array_equal(sqd, sqd_o)
    assert (corr < 0).any() and (corr >= 0).sum() > 500
    assert rel(H, H_o) < H_RTOL and rel(b, b_o) < H_RTOL and abs(err - err_o) / err_o < H_RTOL
    maha = g.getMahalanobis()
    m = corr >= 0
    assert np.allclose(maha[m], maha_o[m], rtol=1e-9, atol=1e-12)
    assert np.all(maha[~m] == 0)
    # compute_error at other poses re-uses the stale correspondences (APD:310-346)
    gw = oracle_apd.geo_weights(cs)
    for dz in (0.0, 0.3):
        T2 = T.copy()
        T2[2, 3] += dz
        e2 = g.compute_error(T2)
        e2_o = oracle_apd.compute_error(T2, sx, sl, tx, tl, gw, p, corr_o, maha_o)
        assert abs(e2 - e2_o) / e2_o < H_RTOL


def test_linearize_with_oracle_covariances_injected(gpu, gorio, oracle_apd, pair2k):
    """setSourceCovariances / setTargetCovariances (APD:138-145): identical covariances in => H, b to rounding."""
    sx, sl, tx, tl, _ = pair2k
    p = oracle_apd.launch_params()
    cs = oracle_apd.calculate_covariances(sx, p)
    ct = oracle_apd.calculate_covariances(tx, p)
    g = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0)
    g.setSourceCovariances(cs)
    g.setTargetCovariances(ct)
    iu = np.triu_indices(4)
    assert np.array_equal(g.getSourceCovariances()[:, iu[0], iu[1]], cs[:, iu[0], iu[1]])  # the upper triangle is what is kept
    err, H, b = g.linearize(_pose())
    err_o, H_o, b_o, *_ = oracle_apd.linearize(_pose(), sx, sl, tx, tl, cs, ct, p)
    assert rel(H, H_o) < 1e-12 and rel(b, b_o) < 1e-11 and abs(err - err_o) / err_o < 1e-12


def test_identical_clouds_identity(gpu, gorio):
    xyz, lab = synth.radar_scan(1500, seed=4)
    g = make(gorio, xyz, lab, xyz, lab, corr_dist_threshold=2.0)
    err, H, b = g.linearize(np.eye(4))
    corr, sqd = g.getCorrespondences()
    assert np.array_equal(corr, np.arange(1500)) and np.all(sqd == 0)
    assert err == 0.0 and np.all(b == 0)
    assert np.all(np.linalg.eigvalsh(H) > 0)


def test_all_rejected(gpu, gorio):
    """Every correspondence beyond the gate: H = b = 0, error = 0 (APD:183-187)."""
    xyz, lab = synth.radar_scan(300, seed=4)
    far = xyz + np.float32(1000.0)
    g = make(gorio, xyz, lab, far, lab, corr_dist_threshold=2.0)
    err, H, b = g.linearize(np.eye(4))
    corr, _ = g.getCorrespondences()
    assert np.all(corr == -1) and err == 0 and not H.any() and not b.any()


def test_default_threshold_accepts_everything(gpu, gorio, oracle_apd):
    """corr_dist_threshold_ defaults to FLT_MAX (APD:23): no rejection."""
    sx, sl, tx, tl, _ = synth.scan_pair(500, 600, seed=2)
    g = make(gorio, sx, sl, tx, tl)
    g.linearize(np.eye(4))
    corr, _ = g.getCorrespondences()
    assert np.all(corr >= 0)
    p = oracle_apd.default_params()
    cs = oracle_apd.calculate_covariances(sx, p)
    ct = oracle_apd.calculate_covariances(tx, p)
    corr_o, _, _ = oracle_apd.update_correspondences(np.eye(4), sx, tx, cs, ct, p)
    assert np.array_equal(corr, corr_o)


@pytest.mark.parametrize("optimizer", ["LM", "GN"])
def test_align_matches_oracle_c1(gpu, gorio, oracle_apd, pose_err, optimizer):
    """BASELINE config C1: 5k x 5k pair, shipped launch parameters."""
    sx, sl, tx, tl, Tgt = synth.scan_pair(5000, 5000, seed=20250704)
    opt = oracle_apd.OPT_LM if optimizer == "LM" else oracle_apd.OPT_GN
    p = oracle_apd.launch_params(optimizer=opt)
    cs = oracle_apd.calculate_covariances(sx, p)
    ct = oracle_apd.calculate_covariances(tx, p)
    ro = oracle_apd.align(np.eye(4), sx, sl, tx, tl, cs, ct, p)
    g = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0, transformation_epsilon=0.1, optimizer=opt)
    r = g.align()
    te, re = pose_err(ro["T"], r["T"])
    assert te < 1e-4 and re < 1e-4, (te, re)
    assert r["converged"] == ro["converged"] and r["nr_iterations"] == ro["nr_iterations"] and r["n_linearize"] == ro["n_linearize"]
    assert rel(r["H"], ro["H"]) < 1e-6
    assert g.hasConverged() == ro["converged"]


def test_align_tight_epsilon_iteration_trace(gpu, gorio, oracle_apd, pose_err):
    """Tight epsilons -> many iterations; the final pose still agrees to 1e-4 and the iteration count is identical."""
    sx, sl, tx, tl, _ = synth.scan_pair(3000, 3000, seed=5, noise_scale=0.1)
    p = oracle_apd.launch_params(transformation_epsilon=1e-3)
    cs = oracle_apd.calculate_covariances(sx, p)
    ct = oracle_apd.calculate_covariances(tx, p)
    ro = oracle_apd.align(np.eye(4), sx, sl, tx, tl, cs, ct, p)
    g = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0, transformation_epsilon=1e-3)
    r = g.align()
    te, re = pose_err(ro["T"], r["T"])
    assert te < 1e-4 and re < 1e-4
    assert r["n_linearize"] == ro["n_linearize"] and r["converged"] == ro["converged"]


def test_swap_source_and_target(gpu, gorio, oracle_apd, pose_err):
    """swapSourceAndTarget (APD:89-98) keeps the covariances with their clouds: backward alignment == oracle backward."""
    sx, sl, tx, tl, _ = synth.scan_pair(1500, 1700, seed=8)
    p = oracle_apd.launch_params()
    cs = oracle_apd.calculate_covariances(sx, p)
    ct = oracle_apd.calculate_covariances(tx, p)
    g = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0, transformation_epsilon=0.1)
    r_f = g.align()
    g.swapSourceAndTarget()
    assert len(g.getSourceCovariances()) == 1700  # covariances travelled with the clouds, not recomputed
    r_b = g.align()
    ro_b = oracle_apd.align(np.eye(4), tx, tl, sx, sl, ct, cs, p)
    te, re = pose_err(ro_b["T"], r_b["T"])
    assert te < 1e-4 and re < 1e-4
    ro_f = oracle_apd.align(np.eye(4), sx, sl, tx, tl, cs, ct, p)
    te, re = pose_err(ro_f["T"], r_f["T"])
    assert te < 1e-4 and re < 1e-4


def test_set_input_invalidates_only_that_cloud(gpu, gorio):
    """setInputTarget clears only the target covariances (APD:133); clearSource drops the source ones (APD:101-105)."""
    sx, sl, tx, tl, _ = synth.scan_pair(400, 500, seed=8)
    g = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0)
    assert len(g.getSourceCovariances()) == 0 and len(g.getTargetCovariances()) == 0  # stale until first use
    g.calculateCovariances()
    assert len(g.getSourceCovariances()) == 400 and len(g.getTargetCovariances()) == 500
    g.setInputTarget(tx[:450], tl[:450])
    assert len(g.getSourceCovariances()) == 400 and len(g.getTargetCovariances()) == 0
    g.clearSource()
    assert len(g.getSourceCovariances()) == 0
    with pytest.raises(gorio.GorioError):
        g.align()  # no source any more


def test_error_conventions(gpu, gorio):
    g = gorio.ApdGicp()
    with pytest.raises(gorio.GorioError):
        g.align()  # nothing set
    xyz, lab = synth.radar_scan(10, seed=1)
    g.setInputSource(xyz, lab)
    g.setInputTarget(xyz, lab)
    with pytest.raises(gorio.GorioError):
        g.align()  # fewer points than k: undefined in the reference (APD:366-369), refused here
    with pytest.raises(gorio.GorioError):
        g.set_params(k_correspondences=33)
    with pytest.raises(gorio.GorioError):
        g.set_params(regularization=7)  # the reference abort()s (APD:389-391)
    with pytest.raises(gorio.GorioError):
        g.compute_error(np.eye(4))  # no correspondences yet


def test_batch_equals_single(gpu, gorio, pose_err):
    """align_batch advances independent pairs in lock-step; every pair must equal its own single align bit for bit."""
    pairs = [synth.scan_pair(900 + 37 * q, 1000 + 91 * q, seed=40 + q) for q in range(5)]
    objs = [make(gorio, *pr[:4], corr_dist_threshold=2.0, transformation_epsilon=0.05) for pr in pairs]
    singles = [o.align() for o in objs]
    objs2 = [make(gorio, *pr[:4], corr_dist_threshold=2.0, transformation_epsilon=0.05) for pr in pairs]
    batch = gorio.align_batch(objs2)
    for s, b in zip(singles, batch):
        assert np.array_equal(s["T"], b["T"]) and s["n_linearize"] == b["n_linearize"] and s["converged"] == b["converged"]
    assert len({b["n_linearize"] for b in batch}) > 1  # the pairs really stop at different iterations


@pytest.mark.parametrize("count", [24, 20])
def test_ragged_batch_with_xcd_placement_equals_single(gpu, gorio, count):
    """24 pairs (>= 16, a multiple of 8): linearize_kernel and the k-NN selection kernels renumber their workgroups so that a pair / a
    cloud runs on one XCD (xcd_grid_pos, apd_device.h); 20 pairs: the plain numbering.  Clouds of very different sizes (workgroups past
    the end of a small cloud leave at once), both optimisers: every pair must equal its own single align bit for bit, covariances too."""
    pairs = [synth.scan_pair(300 + 211 * (q % 7), 2500 - 173 * (q % 11), seed=700 + q) for q in range(count)]
    for opt in (0, 1):  # Gauss-Newton (the step fused into linearize_kernel), Levenberg-Marquardt
        kw = dict(corr_dist_threshold=2.0, transformation_epsilon=0.05, optimizer=opt)
        singles, covs = [], []
        for pr in pairs[:: max(1, count // 6)]:
            o = make(gorio, *pr[:4], **kw)
            singles.append(o.align())
            covs.append(o.getSourceCovariances())
        objs = [make(gorio, *pr[:4], **kw) for pr in pairs]
        batch = gorio.align_batch(objs)
        for k, q in enumerate(range(0, count, max(1, count // 6))):
            assert np.array_equal(singles[k]["T"], batch[q]["T"]) and singles[k]["n_linearize"] == batch[q]["n_linearize"]
            assert np.array_equal(covs[k], objs[q].getSourceCovariances())


def test_transform_source_and_fitness(gpu, gorio, oracle_apd):
    sx, sl, tx, tl, T = synth.scan_pair(800, 900, seed=12)
    g = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0)
    out = g.transformSource(T.astype(np.float32))
    import ctypes as C

    q = np.zeros(3, np.float32)
    Td = T.astype(np.float32).astype(np.float64)
    for i in (0, 17, 799):
        oracle_apd.lib().apdo_transform_point_f(Td.ctypes.data_as(C.POINTER(C.c_double)), sx[i].ctypes.data_as(C.POINTER(C.c_float)), q.ctypes.data_as(C.POINTER(C.c_float)))
        assert np.array_equal(out[i], q)
    score, inl = g.getFitnessScore(T.astype(np.float32))
    p = oracle_apd.default_params()
    z = np.zeros((900, 4, 4))
    corr_o, sqd_o, _ = oracle_apd.update_correspondences(Td, sx, tx, np.zeros((800, 4, 4)), z, p)
    assert score == pytest.approx(float(np.mean(sqd_o.astype(np.float64))), rel=1e-12)
    # inlier fraction: the nodelet hard-codes max_correspondence_dist = 0.5 m (SMO:677-685), independent of corr_dist_threshold
    want = float(np.mean(sqd_o.astype(np.float64) < 0.5 * 0.5))
    assert 0.0 < want < 1.0
    assert inl == pytest.approx(want, rel=1e-12)
    assert g.getFitnessScore(T.astype(np.float32), inlier_dist=0.0)[1] == inl  # <= 0 selects the nodelet's constant
    g.set_params(corr_dist_threshold=float(np.finfo(np.float32).max))
    assert g.getFitnessScore(T.astype(np.float32))[1] == inl  # the registration gate plays no part
    _, inl2 = g.getFitnessScore(T.astype(np.float32), inlier_dist=1.5)
    assert inl2 == pytest.approx(float(np.mean(sqd_o.astype(np.float64) < 2.25)), rel=1e-12) and inl2 > inl


def test_fitness_score_pruned_equals_brute_force(gpu, gorio):
    """getFitnessScore / the inlier fraction through the pruned search: the same numbers as the exhaustive search, for an
    unlimited and for a tight max_range (pcl compares the SQUARED distance with max_range)."""
    sx, sl, tx, tl, T = synth.scan_pair(5000, 4500, seed=13)
    Tf = T.astype(np.float32)
    gb = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0, search=0)
    gp = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0, search=1)
    for max_range in (None, 0.5, 9.0, 0.01):
        for inlier_dist in (0.5, 0.05, 3.0):  # inlier bound below and above max_range: the search bound is the larger of the two
            a = gb.getFitnessScore(Tf, inlier_dist=inlier_dist) if max_range is None else gb.getFitnessScore(Tf, max_range, inlier_dist)
            b = gp.getFitnessScore(Tf, inlier_dist=inlier_dist) if max_range is None else gp.getFitnessScore(Tf, max_range, inlier_dist)
            assert a == b, (max_range, inlier_dist, a, b)
    assert gp.getFitnessScore(Tf, 0.5)[0] < gp.getFitnessScore(Tf, 9.0)[0]


def test_16k_properties(gpu, gorio, pose_err):
    """BASELINE size (16 384 x 16 384): size-independent properties instead of an O(n^2) oracle run.
    (1) identity on identical clouds; (2) a rigidly moved copy is recovered to 1e-4; (3) H is symmetric PSD;
    (4) the error never increases over accepted LM steps (monotone by construction of LSQ:156-170)."""
    xyz, lab = synth.radar_scan(16384, seed=99)
    g = make(gorio, xyz, lab, xyz, lab, corr_dist_threshold=2.0, transformation_epsilon=1e-4, rotation_epsilon=1e-5)
    err, H, b = g.linearize(np.eye(4))
    corr, sqd = g.getCorrespondences()
    assert np.array_equal(corr, np.arange(16384)) and err == 0 and not b.any()
    assert np.allclose(H, H.T) and np.all(np.linalg.eigvalsh(H) > 0)
    T = synth.gt_transform([0.3, -0.2, 0.05], [0.2, 0.1, 1.0])
    moved = (xyz.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    g2 = make(gorio, xyz, lab, moved, lab, corr_dist_threshold=2.0, transformation_epsilon=1e-5, rotation_epsilon=1e-6)
    r = g2.align()
    te, re = pose_err(T, r["T"])
    assert r["converged"] and te < 1e-4 and re < 1e-4, (te, re)


# ---------------------------------------------------------------------------------------------- exact pruned search

@pytest.mark.parametrize("n,m", [(20, 33), (500, 700), (2000, 2300), (5000, 4097)])
def test_pruned_search_equals_brute_force_and_oracle(gpu, gorio, oracle_apd, n, m):
    """GORIO_SEARCH_PRUNED returns the same correspondences (bit-exact, ties included), k-NN lists, covariances, H and b."""
    sx, sl, tx, tl, _ = synth.scan_pair(n, m, seed=70 + n)
    T = _pose()
    gb = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0, search=0)
    gp = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0, search=1)
    eb, Hb, bb = gb.linearize(T)
    ep, Hp, bp = gp.linearize(T)
    cb, sb = gb.getCorrespondences()
    cp, sp = gp.getCorrespondences()
    assert np.array_equal(cb, cp)
    assert np.array_equal(sb[cb >= 0], sp[cp >= 0]) and np.all(np.isinf(sp[cp < 0]))  # rejected points: distance not searched for
    assert np.array_equal(gb.getKnnIndices(0), gp.getKnnIndices(0)) and np.array_equal(gb.getKnnIndices(1), gp.getKnnIndices(1))
    assert np.array_equal(gb.getSourceCovariances(), gp.getSourceCovariances())
    assert rel(Hp, Hb) < 1e-13 and rel(bp, bb) < 1e-12 and abs(ep - eb) <= 1e-13 * abs(eb)
    idx_o, _ = oracle_apd.knn_self(sx, 20)
    assert np.array_equal(gp.getKnnIndices(0), idx_o)


def test_pruned_search_ties_and_duplicates(gpu, gorio, oracle_apd):
    gx, gy = np.meshgrid(np.arange(20, dtype=np.float32), np.arange(20, dtype=np.float32))
    lattice = np.stack([gx.ravel(), gy.ravel(), np.zeros(400, np.float32)], axis=1)
    xyz = np.concatenate([lattice, lattice[:100]])  # exact duplicates + a lattice full of equal distances
    g = make(gorio, xyz, None, xyz, None, regularization=0, search=1, corr_dist_threshold=2.0)
    g.calculateCovariances()
    idx_o, _ = oracle_apd.knn_self(xyz, 20)
    assert np.array_equal(g.getKnnIndices(0), idx_o)
    shifted = lattice + np.float32(0.5)  # every query is equidistant from 4 lattice points
    g2 = make(gorio, shifted, None, xyz, None, search=1, corr_dist_threshold=2.0)
    g2.linearize(np.eye(4))
    corr, _ = g2.getCorrespondences()
    p = oracle_apd.launch_params()
    z = np.zeros((500, 4, 4))
    corr_o, _, _ = oracle_apd.update_correspondences(np.eye(4), shifted, xyz, np.zeros((400, 4, 4)), z, p)
    assert np.array_equal(corr, corr_o)


def test_pruned_search_default_threshold(gpu, gorio):
    """corr_dist_threshold_ = FLT_MAX (APD:23): nothing is gated, the bound comes from the search itself."""
    sx, sl, tx, tl, _ = synth.scan_pair(1500, 1700, seed=3)
    gb = make(gorio, sx, sl, tx, tl, search=0)
    gp = make(gorio, sx, sl, tx, tl, search=1)
    gb.linearize(np.eye(4))
    gp.linearize(np.eye(4))
    cb, sb = gb.getCorrespondences()
    cp, sp = gp.getCorrespondences()
    assert np.array_equal(cb, cp) and np.array_equal(sb, sp) and np.all(cp >= 0)


def test_pruned_align_identical_to_brute_force(gpu, gorio):
    sx, sl, tx, tl, _ = synth.scan_pair(5000, 5000, seed=20250704)
    rb = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0, transformation_epsilon=0.1, search=0).align()
    rp = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0, transformation_epsilon=0.1, search=1).align()
    assert rb["n_linearize"] == rp["n_linearize"] and rb["converged"] == rp["converged"]
    assert np.allclose(rb["T"], rp["T"], rtol=0, atol=1e-6)
    pairs = [synth.scan_pair(900 + 37 * q, 1000 + 91 * q, seed=40 + q) for q in range(4)]
    ob = [make(gorio, *pr[:4], corr_dist_threshold=2.0, transformation_epsilon=0.05, search=0) for pr in pairs]
    op = [make(gorio, *pr[:4], corr_dist_threshold=2.0, transformation_epsilon=0.05, search=1) for pr in pairs]
    for a, b in zip(gorio.align_batch(ob), gorio.align_batch(op)):
        assert a["n_linearize"] == b["n_linearize"] and np.allclose(a["T"], b["T"], rtol=0, atol=1e-6)


def test_pruned_16k_and_map(gpu, gorio):
    """BASELINE sizes: 16k x 16k and 16k x 100k (C3): pruned == brute force on every correspondence."""
    sx, sl = synth.radar_scan(16384, seed=5)
    tx, tl = synth.local_map(100000, seed=6)
    T = synth.gt_transform([0.1, 0.05, 0.0], [0.0, 0.0, 0.5])
    gb = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0, search=0)
    gp = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0, search=1)
    eb, Hb, bb = gb.linearize(T)
    ep, Hp, bp = gp.linearize(T)
    cb, _ = gb.getCorrespondences()
    cp, _ = gp.getCorrespondences()
    assert np.array_equal(cb, cp) and (cb >= 0).sum() > 8000
    assert rel(Hp, Hb) < 1e-12
    assert np.array_equal(gb.getKnnIndices(1), gp.getKnnIndices(1))


@pytest.mark.gpu
def test_device_inputs_single_and_batched_match_host_inputs(gpu, gorio):
    """gorio_apd_set_*_device and gorio_apd_set_clouds_device_batch copy HBM-resident SoA buffers: same poses and correspondences
    as the host-pointer path (what bench.py feeds the timed loop with)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")  # the runtime the library itself is linked against (torch ships its own copy)
    bufs = []

    def dev(a):
        a = np.ascontiguousarray(a, np.float32)
        ptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(ptr), C.c_size_t(a.nbytes)) == 0
        assert hip.hipMemcpy(ptr, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0  # hipMemcpyHostToDevice
        bufs.append(ptr)
        return ptr.value

    pairs = [gorio.synth.scan_pair(3000, 3300, seed=40 + q) for q in range(3)]
    kw = dict(corr_dist_threshold=2.0, search=1, max_iterations=12)
    ref = []
    for sx, sl, tx, tl, _ in pairs:
        g = gorio.ApdGicp(**kw)
        g.setInputTarget(tx, tl)
        g.setInputSource(sx, sl)
        g.align()
        ref.append((g.getFinalTransformation().copy(), g.getCorrespondences()[0].copy()))
    srcs, tgts = [], []
    for sx, sl, tx, tl, _ in pairs:
        srcs.append(([dev(sx[:, 0]), dev(sx[:, 1]), dev(sx[:, 2]), dev(sl)], sx.shape[0]))
        tgts.append(([dev(tx[:, 0]), dev(tx[:, 1]), dev(tx[:, 2]), dev(tl)], tx.shape[0]))
    assert hip.hipDeviceSynchronize() == 0
    # single calls
    g = gorio.ApdGicp(**kw)
    g.setInputTargetDevice(*tgts[0][0], tgts[0][1])
    g.setInputSourceDevice(*srcs[0][0], srcs[0][1])
    g.align()
    assert np.array_equal(g.getFinalTransformation(), ref[0][0])
    assert np.array_equal(g.getCorrespondences()[0], ref[0][1])
    # batched call, twice (the second time the buffers of every handle already exist)
    objs = [gorio.ApdGicp(**kw) for _ in pairs]
    inputs = gorio.DeviceInputs(objs, sources=srcs, targets=tgts)
    for _ in range(2):
        inputs.apply()
        gorio.align_batch(objs)
        for o, (T, corr) in zip(objs, ref):
            assert np.array_equal(o.getFinalTransformation(), T)
            assert np.array_equal(o.getCorrespondences()[0], corr)
    for ptr in bufs:
        hip.hipFree(ptr)


def test_knn_hook_is_optional_and_changes_nothing(gpu, gorio):
    """keep_knn_indices only adds the stores of the parity hook: covariances are bit-identical with and without, and without it
    getKnnIndices is refused (GORIO_ERR_STATE) instead of returning stale data."""
    xyz, lab = synth.radar_scan(3000, seed=31)
    for search in (0, 1):
        a = make(gorio, xyz, lab, xyz, lab, search=search, keep_knn_indices=1)
        b = make(gorio, xyz, lab, xyz, lab, search=search, keep_knn_indices=0)
        a.calculateCovariances()
        b.calculateCovariances()
        assert np.array_equal(a.getSourceCovariances(), b.getSourceCovariances())
        assert a.getKnnIndices(0).shape == (3000, 20)
        with pytest.raises(gorio.GorioError):
            b.getKnnIndices(0)


def test_knn_select_kernel_falls_back_on_massive_ties(gpu, gorio, oracle_apd):
    """More candidates at the k-th distance than the selection kernel buffers (60 copies of one point, a lattice of equal spacings):
    those waves are redone by the insertion kernel; lists stay bit-exact, ties to the lowest index."""
    rng = np.random.default_rng(3)
    base = rng.uniform(-5, 5, (700, 3)).astype(np.float32)
    xyz = np.concatenate([base, np.repeat(base[:5], 60, axis=0)])  # 5 points x 61 copies: 60 zero distances each
    gx, gy, gz = np.meshgrid(np.arange(8, dtype=np.float32), np.arange(8, dtype=np.float32), np.arange(8, dtype=np.float32))
    xyz = np.concatenate([xyz, np.stack([gx.ravel(), gy.ravel(), gz.ravel()], axis=1) + np.float32(20.0)])  # 26 neighbours within sqrt(3)
    idx_o, _ = oracle_apd.knn_self(xyz, 20)
    for search in (0, 1):
        g = make(gorio, xyz, None, xyz, None, regularization=0, search=search)
        g.calculateCovariances()
        assert np.array_equal(g.getKnnIndices(0), idx_o), search


def test_shared_target_equals_private_copies(gpu, gorio):
    """gorio_apd_set_target_shared: N scans against ONE device-resident 100k map (one upload, one index, one k-NN pass) give bit for
    bit what N handles with private copies of the map give; detaching one handle leaves the others on the shared map."""
    tx, tl = synth.local_map(100000, seed=synth.BASE_SEED + 7)
    scans = [synth.radar_scan(4000 + 100 * q, seed=200 + q) for q in range(6)]
    kw = dict(corr_dist_threshold=2.0, search=1, transformation_epsilon=0.05)
    priv = []
    for sx, sl in scans:
        g = make(gorio, sx, sl, tx, tl, **kw)
        priv.append(g)
    rp = gorio.align_batch(priv)
    owner = gorio.ApdGicp(keep_knn_indices=1, **kw)
    owner.setInputTarget(tx, tl)
    shared = []
    for sx, sl in scans:
        g = gorio.ApdGicp(keep_knn_indices=1, **kw)
        g.setInputTargetShared(owner)
        g.setInputSource(sx, sl)
        shared.append(g)
    rs = gorio.align_batch(shared)
    for a, b in zip(rp, rs):
        assert np.array_equal(a["T"], b["T"]) and np.array_equal(a["H"], b["H"]) and a["n_linearize"] == b["n_linearize"]
    assert len(owner.getTargetCovariances()) == 100000  # the owner sees the covariances the batch computed once
    assert np.array_equal(shared[3].getTargetCovariances(), priv[3].getTargetCovariances())
    assert np.array_equal(shared[0].getKnnIndices(1)[:500], priv[0].getKnnIndices(1)[:500])
    # detach: a new target on one sharer does not touch the others
    shared[1].setInputTarget(tx[:50000], tl[:50000])
    r1 = shared[1].align()
    r2 = shared[2].align()
    assert np.array_equal(r2["T"], rp[2]["T"]) and not np.array_equal(r1["T"], rp[1]["T"])
    # single-handle calls on a shared target
    e_s = shared[4].linearize(np.eye(4))
    e_p = priv[4].linearize(np.eye(4))
    assert e_s[0] == e_p[0] and np.array_equal(e_s[1], e_p[1])


def test_batch_validation(gpu, gorio):
    sx, sl, tx, tl, _ = synth.scan_pair(600, 700, seed=9)
    a = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0)
    b = make(gorio, sx, sl, tx, tl, corr_dist_threshold=1.0)
    with pytest.raises(gorio.GorioError):
        gorio.align_batch([a, b])  # different parameters in one lock-step batch
    with pytest.raises(gorio.GorioError):
        gorio.align_batch([a, a])  # the same handle twice
    c = make(gorio, sx, sl, tx, tl, corr_dist_threshold=2.0, cl_weight_points=1200)
    assert len(gorio.align_batch([a, c])) == 2  # cl_weight_points is per handle


@pytest.mark.gpu
def test_shared_target_refuses_other_covariance_parameters(gpu, gorio):
    """A shared target carries ONE set of covariances: a sharer whose k_correspondences / regularization differ from the ones they were
    estimated with is refused (the reference estimates covariances per object with its own settings, APD:149-154) -- at share time when the
    owner's covariances already exist, at align time when the parameters change afterwards; covariances supplied by the caller suit everyone."""
    sx, sl, tx, tl, _ = synth.scan_pair(3000, 3200, seed=91)
    kw = dict(corr_dist_threshold=2.0, search=1, transformation_epsilon=0.05)
    owner = gorio.ApdGicp(**kw)
    owner.setInputTarget(tx, tl)
    owner.setInputSource(sx, sl)
    ro = owner.align()  # estimates the target covariances with k = 20, PLANE
    other = gorio.ApdGicp(k_correspondences=10, **kw)
    with pytest.raises(gorio.GorioError):
        other.setInputTargetShared(owner)
    same = gorio.ApdGicp(**kw)
    same.setInputTargetShared(owner)
    same.setInputSource(sx, sl)
    assert np.array_equal(same.align()["T"], ro["T"])
    same.set_params(k_correspondences=12)
    with pytest.raises(gorio.GorioError):
        same.align()
    same.setInputTarget(tx, tl)  # a target of its own: estimated with k = 12
    assert np.isfinite(same.align()["T"]).all()
    owner.setTargetCovariances(owner.getTargetCovariances())  # supplied covariances carry no parameters
    other.setInputTargetShared(owner)
    other.setInputSource(sx, sl)
    assert np.isfinite(other.align()["T"]).all()


@pytest.mark.gpu
def test_schedule_optimisations_do_not_change_results(gpu, gorio):
    """A Gauss-Newton batch with unequal source sizes (one not a multiple of 256, one of 16 384 points) gives bit for bit the same poses,
    Hessians and counters with (i) the optimiser step fused into the linearisation launch (fence-free hand-over to the last workgroup,
    apd_kernels.hip) or run as its own launch, and (ii) the planned schedule of the pruned search (slow query waves cut into parts, heaviest
    first) or the natural one -- the hooks of gorio_apd_debug_set_schedule exist so that a regression of either can be localised."""
    sizes = [(16384, 16384), (5000, 5200), (4097, 6000), (2300, 2100)]
    pairs = [synth.scan_pair(n, m, seed=300 + q) for q, (n, m) in enumerate(sizes)]
    kw = dict(corr_dist_threshold=2.0, search=1, max_iterations=12, optimizer=0, rotation_epsilon=0.0, transformation_epsilon=0.0)
    out = {}
    for fuse in (True, False):
        for plan in (True, False):
            objs = []
            for sx, sl, tx, tl, _ in pairs:
                g = make(gorio, sx, sl, tx, tl, **kw)
                g.debugSetSchedule(fuse_step=fuse, plan_search=plan)
                objs.append(g)
            out[(fuse, plan)] = gorio.align_batch(objs)
    ref = out[(True, True)]
    assert all(r["n_linearize"] == 12 for r in ref)
    for key, res in out.items():
        for a, b in zip(ref, res):
            assert np.array_equal(a["T"], b["T"]) and np.array_equal(a["H"], b["H"]), key
            assert a["n_linearize"] == b["n_linearize"] and a["nr_iterations"] == b["nr_iterations"] and a["converged"] == b["converged"], key


@pytest.mark.gpu
def test_seeded_and_planned_searches_keep_ties_on_the_lowest_index(gpu, gorio):
    """Equal distances all the way through an align: a lattice target with exact duplicates, a source half a cell off (every query
    equidistant from four lattice points, some of them duplicated, in different tiles), eight fixed Gauss-Newton iterations -- so the
    unseeded search, the seeded one and the planned ones (query waves cut into parts that meet in an atomic min) all see ties.  The pruned
    search must give bit for bit what the exhaustive one gives (ties on the lowest original index), pose and correspondences included."""
    gx, gy = np.meshgrid(np.arange(70, dtype=np.float32), np.arange(70, dtype=np.float32))
    lattice = np.stack([gx.ravel(), gy.ravel(), 0.05 * np.sin(gx.ravel() * 0.3)], axis=1).astype(np.float32)
    tgt = np.concatenate([lattice, lattice[::5]])  # 4900 points + 980 exact duplicates with higher indices
    src = lattice[: 64 * 60] + np.array([0.5, 0.5, 0.0], np.float32)
    kw = dict(corr_dist_threshold=2.0, max_iterations=8, optimizer=0, rotation_epsilon=0.0, transformation_epsilon=0.0)
    res = {}
    for search in (0, 1):
        g = make(gorio, src, None, tgt, None, search=search, **kw)
        r = g.align()
        g.linearize(r["T"].astype(np.float64))
        res[search] = (r, g.getCorrespondences())
    (rb, (cb, sb)), (rp, (cp, sp)) = res[0], res[1]
    assert rb["n_linearize"] == 8 and rp["n_linearize"] == 8
    assert np.array_equal(rb["T"], rp["T"]) and np.array_equal(rb["H"], rp["H"])
    assert np.array_equal(cb, cp) and np.array_equal(sb, sp)
    assert (cp < 4900).all() and (cp >= 0).all()  # never a duplicate's (higher) index
