"""CPU tests that pin the UGPM oracle (no GPU).  The reference has no VelInt test, example or recorded vector:
parity is UNPINNED against the reference itself, so the restatement is checked against analytic cases."""
import importlib

import numpy as np
import pytest
from scipy.integrate import dblquad, quad
from scipy.spatial.transform import Rotation as Rot

synth = importlib.import_module("go-rio_amd.synth")


@pytest.fixture(scope="module")
def ugpm():
    import oracle
    from oracle import ugpm as u

    oracle.build()
    return u


def test_se_kernel_integral_vs_quadrature(ugpm):
    """seKernelIntegral(a, b, x) == integral_a^b seKernel(t, x) dt (MATH:114-126 vs MATH:102-110)."""
    l2, sf2, a = 0.0036, 0.7, 10.0
    xs = np.array([9.9, 10.0, 10.37, 11.2])
    bs = np.array([10.0, 10.25, 11.0])
    Ki = ugpm.se_kernel_integral(a, bs, xs, l2, sf2)
    for i, b in enumerate(bs):
        for j, x in enumerate(xs):
            ref, _ = quad(lambda t: sf2 * np.exp(-0.5 * (t - x) ** 2 / l2), a, b, epsabs=1e-13, epsrel=1e-13, points=[x] if a < x < b else None)
            assert Ki[i, j] == pytest.approx(ref, abs=1e-11)
    K = ugpm.se_kernel(xs, xs, l2, sf2)
    assert np.allclose(np.diag(K), sf2) and np.allclose(K, K.T)


def test_kss_int_vs_double_quadrature(ugpm):
    """kssInt(a, b) == double integral of the SE kernel over [a, b]^2 (MATH:378-382)."""
    l2, sf2, a, b = 0.0036, 1.3, 10.0, 10.4
    ref, _ = dblquad(lambda s, t: sf2 * np.exp(-0.5 * (s - t) ** 2 / l2), a, b, a, b, epsabs=1e-12, epsrel=1e-12)
    assert ugpm.kss_int(a, b, l2, sf2) == pytest.approx(ref, rel=1e-8)


def test_exp_log_maps(ugpm):
    rng = np.random.default_rng(0)
    for scale in (0.0, 1e-9, 1e-3, 1.0, 3.0):
        v = rng.normal(size=3)
        v = v / np.linalg.norm(v) * scale
        R = ugpm.exp_map(v)
        assert np.allclose(R, Rot.from_rotvec(v).as_matrix(), atol=1e-14)  # MATH:55-58
        assert np.allclose(ugpm.log_map(R), v, atol=1e-9)  # MATH:48-51, angle in [0, pi]
    assert np.allclose(ugpm.exp_map(np.zeros(3)), np.eye(3))


def test_jacobian_res_is_the_derivative_of_jr_times_dr(ugpm):
    """JacobianRes (COST:73-145) = d[J_r(r) dr]/d[r, dr]: checked against central differences and two entries of the
    reference's symbolic expression written out by hand (COST:102, 104)."""
    rng = np.random.default_rng(1)
    for scale in (1e-3, 0.3, 2.0):
        r = rng.normal(size=3) * scale
        dr = rng.normal(size=3)
        D = ugpm.jacobian_res(r, dr)
        f = lambda rr, dd: ugpm.jr(rr) @ dd
        num = np.zeros((3, 6))
        h = 1e-6
        for k in range(3):
            e = np.zeros(3)
            e[k] = h
            num[:, k] = (f(r + e, dr) - f(r - e, dr)) / (2 * h)
            num[:, 3 + k] = (f(r, dr + e) - f(r, dr - e)) / (2 * h)
        assert np.allclose(D, num, atol=2e-8 if scale > 0.01 else 1e-6)  # (n - sin n) / n^3 cancels catastrophically for tiny n, in the reference too
        n = np.linalg.norm(r)
        s, c = np.sin(n), np.cos(n)
        assert D[0, 3] == pytest.approx((r[1] ** 2 * (s - n)) / n**3 + (r[2] ** 2 * (s - n)) / n**3 + 1.0, rel=1e-12)  # COST:102
        assert D[0, 4] == pytest.approx(-(r[2] * (c - 1.0)) / n**2 - (r[0] * r[1] * (s - n)) / n**3, rel=1e-10, abs=1e-15)  # COST:104
    D0 = ugpm.jacobian_res(np.zeros(3), np.array([1.0, 2.0, 3.0]))  # small-angle branch COST:137-141
    assert np.allclose(D0[:, 3:], np.eye(3))


def test_constant_rate_rotation_is_exact(ugpm):
    """Noise-free constant angular rate: delta_R = Exp(w T) for both LPM and UGPM."""
    w0 = np.array([0.2, -0.1, 0.5])
    win = synth.imu_window(seed=0, noise=False, omega_fn=lambda t: np.tile(w0, (len(t), 1)), vel_fn=lambda t: np.zeros((len(t), 3)))
    for typ, tol in ((ugpm.LPM, 1e-10), (ugpm.UGPM, 1e-6)):
        res, _ = ugpm.preintegrate(win, type=typ)
        err = np.linalg.norm(Rot.from_matrix(Rot.from_rotvec(w0 * 1.0).as_matrix().T @ res[0]["delta_R"]).as_rotvec())
        assert err < tol, (typ, err)
        assert res[0]["dt"] == pytest.approx(1.0) and res[0]["dt_sq_half"] == pytest.approx(0.5)


def test_constant_velocity_without_rotation(ugpm):
    """Zero rotation + constant body velocity: delta_p = v T."""
    v0 = np.array([5.0, -0.3, 0.1])
    win = synth.imu_window(seed=0, noise=False, omega_fn=lambda t: np.zeros((len(t), 3)), vel_fn=lambda t: np.tile(v0, (len(t), 1)))
    for typ, tol in ((ugpm.LPM, 1e-9), (ugpm.UGPM, 1e-5)):
        res, _ = ugpm.preintegrate(win, type=typ)
        assert np.allclose(res[0]["delta_p"], v0, atol=tol), (typ, res[0]["delta_p"])
        assert np.allclose(res[0]["delta_R"], np.eye(3), atol=1e-9)


def _truth(win, n=100001):
    ts = np.linspace(win["start_t"], win["end_t"], n)
    dt = ts[1] - ts[0]
    om, vv = synth.omega_true(ts), synth.vel_true(ts)
    R, p = np.eye(3), np.zeros(3)
    for i in range(n - 1):
        p += R @ (0.5 * (vv[i] + vv[i + 1])) * dt
        R = R @ Rot.from_rotvec(0.5 * (om[i] + om[i + 1]) * dt).as_matrix()
    return R, p


@pytest.mark.parametrize("vel_hz", [200.0, 20.0])
def test_smooth_motion_noise_free(ugpm, vel_hz):
    """Noise-free C2 window (gyro 200 Hz; velocity 200 Hz -> S = 66, or 20 Hz -> state_freq 20): the GP reproduces the
    continuous-time integrals closely, and beats / matches LPM."""
    win = synth.imu_window(seed=0, noise=False, vel_hz=vel_hz)
    R, p = _truth(win)
    res, d = ugpm.preintegrate(win)
    assert d["nb_state"] == (66 if vel_hz == 200.0 else int(np.ceil(1.0 * d["state_freq"])) + 16)
    rot = np.linalg.norm(Rot.from_matrix(R.T @ res[0]["delta_R"]).as_rotvec())
    pos = np.linalg.norm(p - res[0]["delta_p"])
    assert rot < 2e-4 and pos < 2e-3, (rot, pos)
    c = res[0]["cov"]
    assert np.allclose(c, c.T, atol=1e-12) and np.all(np.linalg.eigvalsh(c) > -1e-12)


def test_noisy_window_error_is_consistent_with_covariance(ugpm):
    win = synth.imu_window(seed=3)
    R, p = _truth(win)
    res, d = ugpm.preintegrate(win)
    rot = Rot.from_matrix(R.T @ res[0]["delta_R"]).as_rotvec()
    sig = np.sqrt(np.diag(res[0]["cov"]))
    assert np.all(np.abs(rot) < 5 * sig[:3])
    assert 1 <= d["iters_rot"] <= 50 and 1 <= d["iters_vel"] <= 50


def test_bias_prior_and_cov_inflation(ugpm):
    win = synth.imu_window(seed=4)
    base, _ = ugpm.preintegrate(win)
    infl, _ = ugpm.preintegrate(win, vel_bias_std=0.3, gyr_bias_std=0.03)  # defaults of VelPreintegration::get (PRE:55)
    dc = infl[0]["cov"] - base[0]["cov"]
    assert np.allclose(dc[:3, :3], 0.03**2 * np.eye(3))  # J = [I 0; d_p_d_bw d_p_d_bv] (PRE:1747-1756)
    assert np.all(np.linalg.eigvalsh(dc) > -1e-12)
    # removing a known gyro bias through the prior == subtracting it from the data (PRE:198-200, 800-802)
    b = np.array([0.01, -0.02, 0.005])
    win_b = dict(win)
    win_b["gyr"] = win["gyr"] + b
    corr, _ = ugpm.preintegrate(win_b, gyr_bias=b)
    assert np.allclose(corr[0]["delta_R"], base[0]["delta_R"], atol=1e-9)


def test_error_conventions(ugpm):
    """Too little data raises (std::range_error in the reference, MATH:493 / PRE:680-686)."""
    win = synth.imu_window(seed=0)
    bad = dict(win)
    bad["gyr_t"], bad["gyr"] = win["gyr_t"][:1], win["gyr"][:1]
    with pytest.raises(RuntimeError):
        ugpm.preintegrate(bad, type=ugpm.LPM)
