"""CPU test: the two independent UGPM restatements pin each other.

oracle/ugpm_oracle.cpp (plain C++: restated Eigen / Ceres arithmetic) and oracle/ugpm_scipy.py (NumPy / SciPy: LAPACK, scipy Rotation,
scipy.optimize.least_squares, complex-step Jacobians) were written separately from the reference text.  Both are parity-UNPINNED
against the reference itself (no VelInt test or vector exists; Eigen / Ceres are absent here), but they must agree AT THE MINIMUM:
GP states, hyper-parameters, delta_R, delta_p, covariance and all five Jacobian outputs, on the C2 windows (200 Hz and 20 Hz
ego-velocity), the long S = 116 window, with bias priors, without correlation, for several query times, and for the LPM output
type.  Tolerances: the C++ side stops where Ceres would (function_tolerance 1e-10, PRE:948) while SciPy runs to 1e-15, so states
agree to ~1e-6 and the pre-integrated measurement to a few 1e-7 (1e-8 of the 10 m travelled in the 2 s window); the gate of SURVEY 8d is 1e-4.
"""
import importlib

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

synth = importlib.import_module("go-rio_amd.synth")


@pytest.fixture(scope="module")
def both():
    import oracle
    from oracle import ugpm as cpp
    from oracle import ugpm_scipy as sp

    oracle.build()
    return cpp, sp


def _agree(a, b, rot_tol=2e-7, pos_tol=5e-7, cov_rtol=1e-3, jac_tol=1e-6):
    rot = np.linalg.norm(Rot.from_matrix(b["delta_R"].T @ a["delta_R"]).as_rotvec())
    pos = np.linalg.norm(a["delta_p"] - b["delta_p"])
    assert rot < rot_tol and pos < pos_tol, (rot, pos)
    assert a["dt"] == pytest.approx(b["dt"], abs=1e-12) and a["dt_sq_half"] == pytest.approx(b["dt_sq_half"], abs=1e-12)
    assert np.abs(a["cov"] - b["cov"]).max() <= cov_rtol * np.abs(b["cov"]).max()
    for k in ("d_delta_R_d_bw", "d_delta_R_d_t", "d_delta_p_d_bw", "d_delta_p_d_bv", "d_delta_p_d_t"):
        assert np.abs(a[k] - b[k]).max() <= jac_tol * max(np.abs(b[k]).max(), 1.0), (k, np.abs(a[k] - b[k]).max())
    return rot, pos


@pytest.mark.parametrize("vel_hz", [200.0, 20.0])
def test_c2_window_states_and_measurement(both, vel_hz):
    cpp, sp = both
    win = synth.imu_window(seed=synth.BASE_SEED + 1, vel_hz=vel_hz)
    ro, do = cpp.preintegrate(win)
    rs, ds = sp.preintegrate(win)
    assert (ds["nb_state"], ds["nb_gyr"], ds["nb_vel"]) == (do["nb_state"], do["nb_gyr"], do["nb_vel"])
    assert ds["state_freq"] == pytest.approx(do["state_freq"], rel=1e-14)
    # the same minimum: 1/2 sum r^2 of both fits (Ceres' `final_cost`), and the states themselves
    # (the velocity fit is solved with the rotation states frozen at each side's own rotation solution, which differ by ~1e-7, and its
    # residuals carry a 1 / sigma_v = 1000 weight: its cost agrees less tightly than the rotation cost)
    assert ds["cost_rot"] == pytest.approx(do["cost_rot"], rel=1e-9) and ds["cost_vel"] == pytest.approx(do["cost_vel"], rel=1e-6)
    assert ds["cost_rot"] <= do["cost_rot"] * (1 + 1e-12)  # SciPy ran to the exact minimum, the C++ side stops where Ceres would
    st, hy = cpp.states(win)
    se3 = ds["se3"]
    assert np.allclose(hy[:, 0], se3.l2, rtol=1e-13) and np.allclose(hy[:, 1], se3.sf2, rtol=1e-9) and np.allclose(hy[:, 3], se3.mean, rtol=0, atol=1e-9)
    assert np.abs(st[:3] - se3.s_r).max() < 1e-6 and np.abs(st[3:] - se3.s_v).max() < 1e-6
    assert np.abs(se3.s_r - se3.init_r).max() > 1e-4  # the fit really moved the states: agreement is not inherited from the LPM seed
    _agree(rs[0], ro[0])


def test_long_window_s116(both):
    cpp, sp = both
    win = synth.imu_window(seed=91, duration=2.0)
    ro, do = cpp.preintegrate(win)
    rs, ds = sp.preintegrate(win)
    assert ds["nb_state"] == do["nb_state"] == 116
    st, _ = cpp.states(win)
    assert np.abs(st[:3] - ds["se3"].s_r).max() < 1e-6 and np.abs(st[3:] - ds["se3"].s_v).max() < 1e-6
    _agree(rs[0], ro[0])


def test_bias_prior_inflation_uncorrelated_and_queries(both):
    cpp, sp = both
    win = synth.imu_window(seed=7)
    kw = dict(gyr_bias=[0.01, -0.02, 0.005], vel_bias=[0.05, 0.0, -0.01], vel_bias_std=0.3, gyr_bias_std=0.03)
    _agree(sp.preintegrate(win, **kw)[0][0], cpp.preintegrate(win, **kw)[0][0])
    _agree(sp.preintegrate(win, correlate=False)[0][0], cpp.preintegrate(win, correlate=False)[0][0])
    q = [win["start_t"] + 0.25, win["start_t"] + 0.6, win["end_t"]]
    for a, b in zip(sp.preintegrate(win, infer_t=q)[0], cpp.preintegrate(win, infer_t=q)[0]):
        _agree(a, b)


def test_short_window_raises_state_frequency(both):
    """duration 0.08 s: state_freq = max(50, 5 / T) = 62.5 Hz (PRE:770), S = ceil(T f) + 16."""
    cpp, sp = both
    win = synth.imu_window(seed=3, duration=0.08)
    ro, do = cpp.preintegrate(win)
    rs, ds = sp.preintegrate(win)
    assert ds["state_freq"] == pytest.approx(62.5) and do["state_freq"] == pytest.approx(62.5) and ds["nb_state"] == do["nb_state"] == 21
    _agree(rs[0], ro[0])


@pytest.mark.parametrize("vel_hz", [200.0, 20.0])
def test_lpm_output_type(both, vel_hz):
    """opt.type = LPM (PRE:1567-1580): IterativeIntegrator::get(0, j) with covariance and numeric Jacobians (PRE:321-391, 524-667).
    No optimiser involved: the two restatements agree to rounding / finite-difference noise (the 1e-4 bias step amplifies 1e-16)."""
    cpp, sp = both
    win = synth.imu_window(seed=11, vel_hz=vel_hz)
    q = [win["start_t"] + 0.3, win["start_t"] + 0.7, win["end_t"]]
    kw = dict(gyr_bias=[0.002, -0.001, 0.003], vel_bias=[0.02, -0.01, 0.0])
    ro, _ = cpp.preintegrate(win, infer_t=q, type=0, **kw)
    rs, _ = sp.preintegrate(win, infer_t=q, type=0, **kw)
    for a, b in zip(rs, ro):
        rot, pos = _agree(a, b, rot_tol=1e-11, pos_tol=1e-11, cov_rtol=1e-9, jac_tol=1e-8)
    # with a coarse ego-velocity stream the 500 Hz filler stamps decide the integration steps: still identical
    ro2, _ = cpp.preintegrate(win, infer_t=q, type=0, min_freq=100.0)
    rs2, _ = sp.preintegrate(win, infer_t=q, type=0, min_freq=100.0)
    for a, b in zip(rs2, ro2):
        _agree(a, b, rot_tol=1e-11, pos_tol=1e-11, cov_rtol=1e-9, jac_tol=1e-8)


def test_scipy_side_analytic_cases(both):
    """The SciPy restatement on its own: constant rate => delta_R = Exp(w T); constant velocity => delta_p = v T."""
    _, sp = both
    w0, v0 = np.array([0.2, -0.1, 0.5]), np.array([5.0, -0.3, 0.1])
    win = synth.imu_window(seed=0, noise=False, omega_fn=lambda t: np.tile(w0, (len(t), 1)), vel_fn=lambda t: np.zeros((len(t), 3)))
    r = sp.preintegrate(win)[0][0]
    assert np.linalg.norm(Rot.from_matrix(Rot.from_rotvec(w0).as_matrix().T @ r["delta_R"]).as_rotvec()) < 1e-6
    win = synth.imu_window(seed=0, noise=False, omega_fn=lambda t: np.zeros((len(t), 3)), vel_fn=lambda t: np.tile(v0, (len(t), 1)))
    r = sp.preintegrate(win)[0][0]
    assert np.allclose(r["delta_p"], v0, atol=1e-5) and np.allclose(r["delta_R"], np.eye(3), atol=1e-9)
