"""Chunked pre-integration (preint.h:1584-1702, math_utils.h:540-726): the oracle's restatement of combinePreints against closed forms,
the library's host-side combine against the oracle, and the chunked oracle against the one-piece oracle.  CPU only."""
import importlib

import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

synth = importlib.import_module("go-rio_amd.synth")


@pytest.fixture(scope="module")
def u():
    import oracle
    from oracle import ugpm

    oracle.build()
    return ugpm


def _skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def _random_meas(rng, dt):
    a = rng.normal(size=(6, 6)) * 1e-3
    return dict(delta_R=Rot.from_rotvec(rng.normal(size=3) * 0.4).as_matrix(), delta_p=rng.normal(size=3) * 2.0, dt=dt, dt_sq_half=0.5 * dt * dt, cov=a @ a.T,
                d_delta_R_d_bw=rng.normal(size=(3, 3)), d_delta_R_d_t=rng.normal(size=3), d_delta_p_d_bw=rng.normal(size=(3, 3)), d_delta_p_d_bv=rng.normal(size=(3, 3)),
                d_delta_p_d_t=rng.normal(size=3))


def test_combine_preints_against_closed_forms(u):
    """With R = R1 R2 and p = p1 + R1 p2, first-order perturbation calculus gives the chained Jacobians in closed form:
    d_R = R2^T d_R1 + d_R2 (the log Jacobian at the identity is the vee map), d_p = d_p1 + R1 d_p2 - R1 [p2]x d_R1; the covariance is
    J blkdiag(C1, C2) J^T with J = [[R2^T, 0, I, 0], [-R1 [p2]x, I, 0, R1]] (forward differences with step 1e-5 in the reference)."""
    rng = np.random.default_rng(5)
    for _ in range(5):
        a, b = _random_meas(rng, 0.7), _random_meas(rng, 0.4)
        c = u.combine_preints(a, b)
        R1, R2, p2 = a["delta_R"], b["delta_R"], b["delta_p"]
        assert np.allclose(c["delta_R"], R1 @ R2, atol=1e-15) and np.allclose(c["delta_p"], a["delta_p"] + R1 @ p2, atol=1e-15)
        assert c["dt"] == pytest.approx(1.1) and c["dt_sq_half"] == pytest.approx(0.5 * 1.1 * 1.1)
        assert np.allclose(c["d_delta_R_d_bw"], R2.T @ a["d_delta_R_d_bw"] + b["d_delta_R_d_bw"], atol=1e-12)
        assert np.allclose(c["d_delta_R_d_t"], R2.T @ a["d_delta_R_d_t"] + b["d_delta_R_d_t"], atol=1e-12)
        assert np.allclose(c["d_delta_p_d_bv"], a["d_delta_p_d_bv"] + R1 @ b["d_delta_p_d_bv"], atol=1e-13)
        assert np.allclose(c["d_delta_p_d_bw"], a["d_delta_p_d_bw"] + R1 @ b["d_delta_p_d_bw"] - R1 @ _skew(p2) @ a["d_delta_R_d_bw"], atol=1e-12)
        assert np.allclose(c["d_delta_p_d_t"], a["d_delta_p_d_t"] + R1 @ b["d_delta_p_d_t"] - R1 @ _skew(p2) @ a["d_delta_R_d_t"], atol=1e-12)
        J = np.zeros((6, 12))
        J[0:3, 0:3], J[0:3, 6:9] = R2.T, np.eye(3)
        J[3:6, 0:3], J[3:6, 3:6], J[3:6, 9:12] = -R1 @ _skew(p2), np.eye(3), R1
        C = np.zeros((12, 12))
        C[:6, :6], C[6:, 6:] = a["cov"], b["cov"]
        want = J @ C @ J.T
        assert np.allclose(c["cov"], want, rtol=1e-3, atol=1e-4 * np.abs(want).max())
    # a zero-length second interval returns the first one untouched (math_utils.h:691-692)
    z = _random_meas(rng, 0.0)
    c = u.combine_preints(a, z)
    assert np.array_equal(c["delta_R"], a["delta_R"]) and np.array_equal(c["cov"], a["cov"]) and c["dt"] == a["dt"]


def test_log_jacobian_branches_of_combine(u):
    """propagateJacobianRR goes through jacobianLogMap(R2^T R2): numerically the identity, whose trace decides between the closed form
    and the constant branch (math_utils.h:233, 306-309).  A pair that is far from the identity exercises the long branch instead:
    chaining (a, b) then c must agree with first-order calculus as well."""
    rng = np.random.default_rng(6)
    a, b, c = _random_meas(rng, 0.3), _random_meas(rng, 0.3), _random_meas(rng, 0.3)
    ab = u.combine_preints(a, b)
    abc = u.combine_preints(ab, c)
    R3 = c["delta_R"]
    assert np.allclose(abc["d_delta_R_d_bw"], R3.T @ ab["d_delta_R_d_bw"] + c["d_delta_R_d_bw"], atol=1e-12)
    assert np.allclose(abc["delta_R"], a["delta_R"] @ b["delta_R"] @ R3, atol=1e-14)


def test_library_combine_matches_oracle(u):
    """gorio_ugpm_combine_preints is host arithmetic inside libgorio_amd.so (no device involved): same operations in the same order."""
    gorio = importlib.import_module("go-rio_amd")
    rng = np.random.default_rng(7)
    for _ in range(4):
        a, b = _random_meas(rng, 0.5), _random_meas(rng, 0.25)
        want, got = u.combine_preints(a, b), gorio.ugpm_combine_preints(a, b)
        for k in want:
            assert np.allclose(got[k], want[k], rtol=1e-13, atol=1e-15), k


def test_chunked_oracle_close_to_one_piece(u):
    """Chunks of 0.7 s of a 2 s request: same elapsed times, poses within the accuracy of a GP window whose data stop 8 sample periods
    beyond its ends (preint.h:1605-1607 against 8 STATE periods of padding, :777-783), stamps of the first chunk not chained."""
    w = synth.imu_window(seed=77, duration=2.0)
    q = [w["start_t"] + 0.6, w["start_t"] + 1.3, w["end_t"]]
    full, _ = u.preintegrate(w, infer_t=q)
    ch, d = u.preintegrate_chunked(w, 0.7, infer_t=q)
    assert len(ch) == 1 and len(ch[0]) == 3
    for a, b in zip(full, ch[0]):
        assert a["dt"] == pytest.approx(b["dt"], abs=1e-12)
        assert np.linalg.norm(Rot.from_matrix(b["delta_R"].T @ a["delta_R"]).as_rotvec()) < 2e-3
        assert np.linalg.norm(a["delta_p"] - b["delta_p"]) < 2e-3
    # one chunk longer than the request = a plain window over data cut at start - 8 sample periods
    one, _ = u.preintegrate_chunked(w, 10.0, infer_t=q)
    w2 = dict(w)
    keep = w["gyr_t"] > w["start_t"] - 8 * 0.005
    w2["gyr_t"], w2["gyr"] = w["gyr_t"][keep], w["gyr"][keep]
    keepv = w["vel_t"] > w["start_t"] - 8 * 0.005
    w2["vel_t"], w2["vel"] = w["vel_t"][keepv], w["vel"][keepv]
    plain, _ = u.preintegrate(w2, infer_t=q)
    for a, b in zip(plain, one[0]):
        assert np.array_equal(a["delta_R"], b["delta_R"]) and np.array_equal(a["cov"], b["cov"])
