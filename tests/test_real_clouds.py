"""Known-transform registration on REAL sensor data, in the acceptance shape of the reference's own registration test
(/root/reference/fast_apdgicp/src/test/gicp_test.cpp:148-201: forward / backward / swap-and-set-source / swap-and-set-target, pose
within 0.05 m and 1 degree, hasConverged()).

The reference's test never instantiates FastAPDGICP and its data directory is absent, so no reference-held answer exists; the only
real clouds in the tree are the two consecutive LiDAR scans under ndt_omp/data.  tests/golden/real_lidar_pair.npz holds thinned
copies (data only; tests/golden/make_real_clouds.py made it).  Two disjoint samples of ONE scan, one of them moved by a known rigid
transform, give a registration problem whose answer is known; the pair of consecutive scans gives a forward / backward
consistency check.  The CPU test runs the oracle, the GPU test runs the HIP path through the C ABI and additionally compares it with
the oracle on these inputs (poses 1e-4, indices bit-exact).
"""
import importlib
import os

import numpy as np
import pytest

synth = importlib.import_module("go-rio_amd.synth")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "real_lidar_pair.npz")
T_TOL, R_TOL = 0.05, np.deg2rad(1.0)  # gicp_test.cpp:149-150


@pytest.fixture(scope="module")
def clouds():
    g = np.load(GOLD)  # allow_pickle stays False
    a0, a1, b0 = g["a_0"][:, :3].copy(), g["a_1"][:, :3].copy(), g["b_0"][:, :3].copy()
    T = synth.gt_transform([0.30, -0.20, 0.05], [0.5, -0.4, 2.0])
    moved = (a1.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    return dict(source=a0, target=moved, T=T, next_scan=b0)


def _zeros(x):
    return np.zeros(len(x), np.float32)  # no cluster labels on LiDAR data: every normal_x is 0 (APD:272 then always matches)


class OracleReg:
    """The oracle behind the same four calls the test needs (covariances are cached per cloud object, as the class does)."""

    def __init__(self, oa, **kw):
        self.oa, self.p = oa, oa.launch_params(**kw)
        self.src = self.tgt = None

    def _cov(self, x):
        return self.oa.calculate_covariances(x, self.p)

    def setInputSource(self, x):  # noqa: N802
        self.src = (x, self._cov(x))

    def setInputTarget(self, x):  # noqa: N802
        self.tgt = (x, self._cov(x))

    def swapSourceAndTarget(self):  # noqa: N802
        self.src, self.tgt = self.tgt, self.src

    def align(self):
        (s, cs), (t, ct) = self.src, self.tgt
        return self.oa.align(np.eye(4), s, _zeros(s), t, _zeros(t), cs, ct, self.p)


class GpuReg:
    def __init__(self, gorio, **kw):
        self.g = gorio.ApdGicp(corr_dist_threshold=2.0, transformation_epsilon=0.1, **kw)

    def setInputSource(self, x):  # noqa: N802
        self.g.setInputSource(x, _zeros(x))

    def setInputTarget(self, x):  # noqa: N802
        self.g.setInputTarget(x, _zeros(x))

    def swapSourceAndTarget(self):  # noqa: N802
        self.g.swapSourceAndTarget()

    def align(self):
        return self.g.align()


def _gicp_test_shape(make, c, pose_err):
    """gicp_test.cpp:159-201 verbatim in structure."""
    source, target, T = c["source"], c["target"], c["T"]
    out = {}
    reg = make()
    reg.setInputTarget(target)
    reg.setInputSource(source)
    r = reg.align()  # forward test
    te, re = pose_err(T, r["T"])
    assert te < T_TOL and re < R_TOL and r["converged"], ("FORWARD", te, re)
    out["forward"] = r
    reg.setInputTarget(source)
    reg.setInputSource(target)
    r = reg.align()  # backward test
    te, re = pose_err(T, np.linalg.inv(r["T"].astype(np.float64)))
    assert te < T_TOL and re < R_TOL and r["converged"], ("BACKWARD", te, re)
    out["backward"] = r
    reg = make()  # swap and set source
    reg.setInputSource(target)
    reg.swapSourceAndTarget()
    reg.setInputSource(source)
    r = reg.align()
    te, re = pose_err(T, r["T"])
    assert te < T_TOL and re < R_TOL and r["converged"], ("SWAP AND SET SOURCE", te, re)
    out["swap_source"] = r
    reg = make()  # swap and set target
    reg.setInputTarget(source)
    reg.swapSourceAndTarget()
    reg.setInputTarget(target)
    r = reg.align()
    te, re = pose_err(T, r["T"])
    assert te < T_TOL and re < R_TOL and r["converged"], ("SWAP AND SET TARGET", te, re)
    out["swap_target"] = r
    return out


def test_oracle_known_transform_on_real_scan(oracle_apd, clouds, pose_err):
    out = _gicp_test_shape(lambda: OracleReg(oracle_apd), clouds, pose_err)
    assert np.array_equal(out["forward"]["T"], out["swap_source"]["T"]) and np.array_equal(out["forward"]["T"], out["swap_target"]["T"])
    te, re = pose_err(clouds["T"], out["forward"]["T"])
    assert te < 0.01 and re < np.deg2rad(0.1)  # observed 2 mm / 0.01 deg: far inside the reference's own acceptance band


def test_oracle_consecutive_real_scans_forward_backward_consistent(oracle_apd, clouds, pose_err):
    a, b = clouds["source"], clouds["next_scan"]
    reg = OracleReg(oracle_apd)
    reg.setInputTarget(b)
    reg.setInputSource(a)
    rf = reg.align()
    reg.swapSourceAndTarget()
    rb = reg.align()
    assert rf["converged"] and rb["converged"]
    assert 0.3 < np.linalg.norm(rf["T"][:3, 3]) < 0.7  # the platform really moved between the two scans (0.48 m)
    te, re = pose_err(np.eye(4), rf["T"].astype(np.float64) @ rb["T"].astype(np.float64))
    assert te < T_TOL and re < R_TOL, (te, re)


@pytest.mark.gpu
@pytest.mark.parametrize("search", [0, 1])
def test_gpu_known_transform_on_real_scan(gpu, gorio, oracle_apd, clouds, pose_err, search):
    out = _gicp_test_shape(lambda: GpuReg(gorio, search=search), clouds, pose_err)
    ref = _gicp_test_shape(lambda: OracleReg(oracle_apd), clouds, pose_err)
    for k in out:  # every one of the four alignments equals the oracle's
        te, re = pose_err(ref[k]["T"], out[k]["T"])
        assert te < 1e-4 and re < 1e-4, (k, te, re)
        assert out[k]["n_linearize"] == ref[k]["n_linearize"] and out[k]["nr_iterations"] == ref[k]["nr_iterations"]


@pytest.mark.gpu
def test_gpu_real_scan_indices_bit_exact(gpu, gorio, oracle_apd, clouds):
    """Real (unevenly sampled, partly planar) data: k-NN lists and correspondences bit-exact, H / b / error 1e-9 vs the oracle."""
    a, b = clouds["source"], clouds["next_scan"]
    p = oracle_apd.launch_params()
    idx_o, _ = oracle_apd.knn_self(a, 20)
    cs, ct = oracle_apd.calculate_covariances(a, p), oracle_apd.calculate_covariances(b, p)
    err_o, H_o, b_o, corr_o, sqd_o, _ = oracle_apd.linearize(np.eye(4), a, _zeros(a), b, _zeros(b), cs, ct, p)
    for search in (0, 1):
        g = gorio.ApdGicp(corr_dist_threshold=2.0, search=search, keep_knn_indices=1)
        g.setInputTarget(b, _zeros(b))
        g.setInputSource(a, _zeros(a))
        err, H, bb = g.linearize(np.eye(4))
        corr, sqd = g.getCorrespondences()
        assert np.array_equal(g.getKnnIndices(0), idx_o)
        assert np.array_equal(corr, corr_o) and np.array_equal(sqd[corr >= 0], sqd_o[corr >= 0])
        assert np.abs(H - H_o).max() / np.abs(H_o).max() < 1e-9 and np.abs(bb - b_o).max() / np.abs(b_o).max() < 1e-9 and abs(err - err_o) / err_o < 1e-9


@pytest.mark.gpu
def test_gpu_consecutive_real_scans(gpu, gorio, oracle_apd, clouds, pose_err):
    a, b = clouds["source"], clouds["next_scan"]
    reg, ref = GpuReg(gorio, search=1), OracleReg(oracle_apd)
    for r_ in (reg, ref):
        r_.setInputTarget(b)
        r_.setInputSource(a)
    rf, of = reg.align(), ref.align()
    for r_ in (reg, ref):
        r_.swapSourceAndTarget()
    rb, ob = reg.align(), ref.align()
    for x, o in ((rf, of), (rb, ob)):
        te, re = pose_err(o["T"], x["T"])
        assert te < 1e-4 and re < 1e-4 and x["converged"] == o["converged"] and x["n_linearize"] == o["n_linearize"]
    te, re = pose_err(np.eye(4), rf["T"].astype(np.float64) @ rb["T"].astype(np.float64))
    assert te < T_TOL and re < R_TOL
