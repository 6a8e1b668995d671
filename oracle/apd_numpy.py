"""Independent NumPy/SciPy restatement of APD-GICP's linearisation (TEST INFRASTRUCTURE ONLY).

Different code, same maths as oracle/apd_oracle.c: it exists so that the two restatements pin each other, because the
reference ships no test or fixture for FastAPDGICP and cannot be compiled here.  Where the C oracle hand-rolls
arithmetic, this file uses library routines on purpose:
  * 1-NN / k-NN: dense float32 distance matrices (numpy float32 ops are IEEE single without FMA, i.e. FLANN L2_Simple),
    argmin / stable argsort (ties -> lowest index);
  * covariance regularisation: numpy.linalg.svd -- a TRUE SVD with separate U and V, as Eigen::JacobiSVD at
    fast_apdgicp_impl.hpp:385, which checks the oracle's "U == V for symmetric PSD" shortcut;
  * Mahalanobis: numpy.linalg.inv of the full 4x4 (fast_apdgicp_impl.hpp:213-218);
  * H, b: explicit 4x6 Jacobians and matrix products (fast_apdgicp_impl.hpp:284-293).
Only for small clouds (dense n x m matrices).
"""
import numpy as np


def knn_self(xyz, k):
    x = np.asarray(xyz, np.float32)
    d = np.zeros((x.shape[0], x.shape[0]), np.float32)
    for a in range(3):  # ((dx*dx) + dy*dy) + dz*dz in float32
        diff = x[:, None, a] - x[None, :, a]
        d = d + diff * diff if a else diff * diff
    idx = np.argsort(d, axis=1, kind="stable")[:, :k]
    return idx.astype(np.int32), np.take_along_axis(d, idx, axis=1)


def covariances(xyz, knn_idx, regularization="PLANE"):
    x = np.asarray(xyz, np.float32).astype(np.float64)
    n, k = knn_idx.shape
    out = np.zeros((n, 4, 4))
    for i in range(n):
        nb = np.concatenate([x[knn_idx[i]], np.ones((k, 1))], axis=1).T  # 4 x k  (fast_apdgicp_impl.hpp:366-369)
        nb = nb - nb.mean(axis=1, keepdims=True)  # :371
        cov = nb @ nb.T / k  # :372
        if regularization == "NONE":
            out[i] = cov
            continue
        if regularization == "FROBENIUS":
            C = cov[:3, :3] + 1e-3 * np.eye(3)
            Ci = np.linalg.inv(C)
            out[i, :3, :3] = np.linalg.inv(Ci / np.linalg.norm(Ci))
            continue
        U, s, Vt = np.linalg.svd(cov[:3, :3])  # :385
        if regularization == "PLANE":
            vals = np.array([1.0, 1.0, 1e-3])
        elif regularization == "MIN_EIG":
            vals = np.maximum(s, 1e-3)
        elif regularization == "NORMALIZED_MIN_EIG":
            vals = np.maximum(s / s.max(), 1e-3)
        else:
            raise ValueError(regularization)
        out[i, :3, :3] = U @ np.diag(vals) @ Vt  # :405
    return out


def transform_f32(T, xyz):
    """Eigen Isometry3f * Vector4f in float32: ((m0 x + m1 y) + m2 z) + m3."""
    Tf = np.asarray(T, np.float64).astype(np.float32)
    x = np.asarray(xyz, np.float32)
    q = np.empty_like(x)
    for r in range(3):
        a = Tf[r, 0] * x[:, 0]
        a = a + Tf[r, 1] * x[:, 1]
        a = a + Tf[r, 2] * x[:, 2]
        q[:, r] = a + Tf[r, 3]
    return q


def nearest(q, tgt):
    t = np.asarray(tgt, np.float32)
    d = None
    for a in range(3):
        diff = q[:, None, a] - t[None, :, a]
        d = diff * diff if d is None else d + diff * diff
    j = np.argmin(d, axis=1)  # first minimum == lowest index
    return j.astype(np.int32), d[np.arange(q.shape[0]), j]


def rot_z(a):
    return np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])


def rot_y(a):
    return np.array([[np.cos(a), 0, np.sin(a)], [0, 1.0, 0], [-np.sin(a), 0, np.cos(a)]])


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0.0]])


def linearize(T, src_xyz, src_label, tgt_xyz, tgt_label, src_cov, tgt_cov, corr_dist_threshold=2.0, dist_var=0.86, azimuth_var=0.5,
              elevation_var=1.0):
    """fast_apdgicp_impl.hpp:160-307.  Returns (error, H, b, corr, sqd, maha)."""
    T = np.asarray(T, np.float64)
    src = np.asarray(src_xyz, np.float32)
    tgt = np.asarray(tgt_xyz, np.float32)
    n = src.shape[0]
    q = transform_f32(T, src)
    j, d = nearest(q, tgt)
    corr = np.where(d.astype(np.float64) < corr_dist_threshold * corr_dist_threshold, j, -1).astype(np.int32)
    maha = np.zeros((n, 4, 4))
    H = np.zeros((6, 6))
    b = np.zeros(6)
    err = 0.0
    for i in range(n):
        if corr[i] < 0:
            continue
        pt = q[i]
        dist = np.linalg.norm(pt.astype(np.float64))
        s = np.array([dist * dist_var / 400, dist * np.sin(azimuth_var / 180 * np.pi), dist * np.sin(elevation_var / 180 * np.pi)])
        rxy = np.float32(np.sqrt(np.float64(np.float32(pt[0] * pt[0] + pt[1] * pt[1]))))  # sqrt(float) of a float32 sum
        elevation = np.float64(np.float32(np.arctan2(np.float64(rxy), np.float64(pt[2]))))
        azimuth = np.float64(np.float32(np.arctan2(np.float64(pt[1]), np.float64(pt[0]))))
        A = rot_z(azimuth) @ rot_y(elevation) @ np.diag(s)
        cov_dist = np.zeros((4, 4))
        cov_dist[:3, :3] = A @ A.T
        RCR = (tgt_cov[corr[i]] + cov_dist) + T @ (src_cov[i] + cov_dist) @ T.T
        RCR[3, 3] = 1.0
        M = np.linalg.inv(RCR)
        M[3, 3] = 0.0
        maha[i] = M
        a4 = np.append(src[i].astype(np.float64), 1.0)
        b4 = np.append(tgt[corr[i]].astype(np.float64), 1.0)
        Ta = T @ a4
        e = b4 - Ta
        sv = np.linalg.svd(src_cov[i][:3, :3], compute_uv=False)
        geo = (sv / sv.max())[2]
        cl = 1.0 / n if tgt_label[corr[i]] == src_label[i] else 0.0
        err += (1.0 + geo + cl) * (e @ M @ e)
        J = np.zeros((4, 6))
        J[:3, :3] = skew(Ta[:3])
        J[:3, 3:] = -np.eye(3)
        H += J.T @ M @ J
        b += J.T @ M @ e
    return err, H, b, corr, d, maha
