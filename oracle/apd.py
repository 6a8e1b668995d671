"""ctypes binding of oracle/apd_oracle.c (CPU restatement of fast_apdgicp_impl.hpp / lsq_registration_impl.hpp).

Test infrastructure only -- see oracle/__init__.py.
"""
import ctypes as C
import os

import numpy as np

from . import BUILD_DIR, build

REG_NONE, REG_MIN_EIG, REG_NORMALIZED_MIN_EIG, REG_PLANE, REG_FROBENIUS = range(5)
OPT_GN, OPT_LM = 0, 1


class Params(C.Structure):
    _fields_ = [
        ("k_correspondences", C.c_int),
        ("regularization", C.c_int),
        ("dist_var", C.c_double),
        ("azimuth_var", C.c_double),
        ("elevation_var", C.c_double),
        ("corr_dist_threshold", C.c_double),
        ("max_iterations", C.c_int),
        ("rotation_epsilon", C.c_double),
        ("transformation_epsilon", C.c_double),
        ("optimizer", C.c_int),
        ("lm_max_iterations", C.c_int),
        ("lm_init_lambda_factor", C.c_double),
        ("num_threads", C.c_int),
        ("search", C.c_int),
    ]


class Counters(C.Structure):
    _fields_ = [("n_linearize", C.c_int), ("n_compute_error", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(BUILD_DIR, "libapd_oracle.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.apdo_linearize.restype = C.c_double
        _lib.apdo_compute_error.restype = C.c_double
    return _lib


def default_params(**kw) -> Params:
    p = Params()
    lib().apdo_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def launch_params(**kw) -> Params:
    """Shipped values of launch/ntu_loop3.launch:85-96 on top of the library defaults."""
    base = dict(corr_dist_threshold=2.0, transformation_epsilon=0.1, max_iterations=64, k_correspondences=20)
    base.update(kw)
    return default_params(**base)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def knn_self(xyz, k, num_threads=0, kdtree=False):
    xyz = _f32(xyz)
    n = xyz.shape[0]
    idx = np.empty((n, k), np.int32)
    sqd = np.empty((n, k), np.float32)
    fn = lib().apdo_knn_self_kdtree if kdtree else lib().apdo_knn_self
    rc = fn(_p(xyz, C.c_float), n, k, _p(idx, C.c_int), _p(sqd, C.c_float), num_threads)
    if rc != 0:
        raise ValueError(f"apdo_knn_self rc={rc} (needs n >= k)")
    return idx, sqd


def covariances_from_knn(xyz, knn_idx, regularization=REG_PLANE, num_threads=0):
    xyz = _f32(xyz)
    knn_idx = np.ascontiguousarray(knn_idx, np.int32)
    n, k = knn_idx.shape
    cov = np.empty((n, 4, 4), np.float64)
    rc = lib().apdo_covariances_from_knn(_p(xyz, C.c_float), n, _p(knn_idx, C.c_int), k, regularization, _p(cov, C.c_double), num_threads)
    if rc != 0:
        raise ValueError(f"apdo_covariances_from_knn rc={rc}")
    return cov


def calculate_covariances(xyz, params: Params):
    xyz = _f32(xyz)
    n = xyz.shape[0]
    cov = np.empty((n, 4, 4), np.float64)
    rc = lib().apdo_calculate_covariances(_p(xyz, C.c_float), n, C.byref(params), _p(cov, C.c_double))
    if rc != 0:
        raise ValueError(f"apdo_calculate_covariances rc={rc}")
    return cov


def geo_weights(cov):
    cov = _f64(cov)
    n = cov.shape[0]
    w = np.empty(n, np.float64)
    lib().apdo_geo_weights(_p(cov, C.c_double), n, _p(w, C.c_double))
    return w


def update_correspondences(T, src_xyz, tgt_xyz, src_cov, tgt_cov, params):
    T = _f64(T)
    src_xyz, tgt_xyz = _f32(src_xyz), _f32(tgt_xyz)
    src_cov, tgt_cov = _f64(src_cov), _f64(tgt_cov)
    n, m = src_xyz.shape[0], tgt_xyz.shape[0]
    corr = np.empty(n, np.int32)
    sqd = np.empty(n, np.float32)
    maha = np.zeros((n, 4, 4), np.float64)
    lib().apdo_update_correspondences(
        _p(T, C.c_double), _p(src_xyz, C.c_float), n, _p(tgt_xyz, C.c_float), m, _p(src_cov, C.c_double), _p(tgt_cov, C.c_double),
        C.byref(params), _p(corr, C.c_int), _p(sqd, C.c_float), _p(maha, C.c_double))
    return corr, sqd, maha


def linearize(T, src_xyz, src_label, tgt_xyz, tgt_label, src_cov, tgt_cov, params, geo_w=None):
    """Returns (error, H 6x6, b 6, corr, sqd, maha)."""
    T = _f64(T)
    src_xyz, tgt_xyz = _f32(src_xyz), _f32(tgt_xyz)
    src_label, tgt_label = _f32(src_label), _f32(tgt_label)
    src_cov, tgt_cov = _f64(src_cov), _f64(tgt_cov)
    n, m = src_xyz.shape[0], tgt_xyz.shape[0]
    if geo_w is None:
        geo_w = geo_weights(src_cov)
    geo_w = _f64(geo_w)
    corr = np.empty(n, np.int32)
    sqd = np.empty(n, np.float32)
    maha = np.zeros((n, 4, 4), np.float64)
    H = np.zeros((6, 6), np.float64)
    b = np.zeros(6, np.float64)
    err = lib().apdo_linearize(
        _p(T, C.c_double), _p(src_xyz, C.c_float), _p(src_label, C.c_float), n, _p(tgt_xyz, C.c_float), _p(tgt_label, C.c_float), m,
        _p(src_cov, C.c_double), _p(tgt_cov, C.c_double), _p(geo_w, C.c_double), C.byref(params),
        _p(corr, C.c_int), _p(sqd, C.c_float), _p(maha, C.c_double), _p(H, C.c_double), _p(b, C.c_double))
    return err, H, b, corr, sqd, maha


def compute_error(T, src_xyz, src_label, tgt_xyz, tgt_label, geo_w, params, corr, maha):
    T = _f64(T)
    src_xyz, tgt_xyz = _f32(src_xyz), _f32(tgt_xyz)
    src_label, tgt_label = _f32(src_label), _f32(tgt_label)
    geo_w, maha = _f64(geo_w), _f64(maha)
    corr = np.ascontiguousarray(corr, np.int32)
    n = src_xyz.shape[0]
    return lib().apdo_compute_error(
        _p(T, C.c_double), _p(src_xyz, C.c_float), _p(src_label, C.c_float), n, _p(tgt_xyz, C.c_float), _p(tgt_label, C.c_float),
        _p(geo_w, C.c_double), C.byref(params), _p(corr, C.c_int), _p(maha, C.c_double))


def align(guess, src_xyz, src_label, tgt_xyz, tgt_label, src_cov, tgt_cov, params, want_trace=False):
    """Full LsqRegistration::computeTransformation.

    Returns dict(T, H, converged, nr_iterations, n_linearize, n_compute_error[, trace, trace_corr]).
    """
    guess = _f32(guess)
    src_xyz, tgt_xyz = _f32(src_xyz), _f32(tgt_xyz)
    src_label, tgt_label = _f32(src_label), _f32(tgt_label)
    src_cov, tgt_cov = _f64(src_cov), _f64(tgt_cov)
    n, m = src_xyz.shape[0], tgt_xyz.shape[0]
    T = np.zeros((4, 4), np.float32)
    H = np.zeros((6, 6), np.float64)
    conv, nit = C.c_int(0), C.c_int(0)
    cnt = Counters()
    trace = trace_corr = None
    tp = cp = None
    if want_trace:
        trace = np.full((params.max_iterations, 4, 4), np.nan)
        trace_corr = np.full((params.max_iterations, n), -2, np.int32)
        tp, cp = _p(trace, C.c_double), _p(trace_corr, C.c_int)
    rc = lib().apdo_align(
        _p(guess, C.c_float), _p(src_xyz, C.c_float), _p(src_label, C.c_float), n, _p(tgt_xyz, C.c_float), _p(tgt_label, C.c_float), m,
        _p(src_cov, C.c_double), _p(tgt_cov, C.c_double), C.byref(params), _p(T, C.c_float), _p(H, C.c_double),
        C.byref(conv), C.byref(nit), C.byref(cnt), tp, cp)
    if rc != 0:
        raise RuntimeError(f"apdo_align rc={rc}")
    out = dict(T=T, H=H, converged=bool(conv.value), nr_iterations=nit.value, n_linearize=cnt.n_linearize, n_compute_error=cnt.n_compute_error)
    if want_trace:
        out["trace"] = trace[: cnt.n_linearize]
        out["trace_corr"] = trace_corr[: cnt.n_linearize]
    return out


def submap_assemble(frames, rel_poses, voxel_leaf=0.0):
    """SMO:602-618: transform the keyframe clouds by rel_poses (4x4 double), concatenate, downsample (voxel_leaf <= 0: the launch
    files' NONE = PassThrough).  frames: list of (xyz [n,3] float32, label [n] float32).  Returns (xyz, label)."""
    xyz = _f32(np.concatenate([f[0] for f in frames]))
    lab = _f32(np.concatenate([f[1] for f in frames]))
    cnt = np.ascontiguousarray([f[0].shape[0] for f in frames], np.int32)
    T = _f64(np.stack([np.asarray(t, np.float64) for t in rel_poses]))
    cap = xyz.shape[0]
    ox = np.empty((max(cap, 1), 3), np.float32)
    ol = np.empty(max(cap, 1), np.float32)
    n = lib().apdo_submap_assemble(_p(xyz, C.c_float), _p(lab, C.c_float), _p(cnt, C.c_int), _p(T, C.c_double), len(frames), C.c_double(voxel_leaf),
                                   _p(ox, C.c_float), _p(ol, C.c_float), cap)
    if n < 0:
        raise RuntimeError("apdo_submap_assemble failed")
    return ox[:n].copy(), ol[:n].copy()


def dbscan_labels(xyz, eps=0.9, min_pts=10, min_cluster=20, max_cluster=25000):
    """preprocessing_nodelet_ntu.cpp:518-568: DBSCAN cluster labels (normal_x) of one scan, ranked by centroid distance; 0 = unclustered."""
    xyz = _f32(xyz)
    n = xyz.shape[0]
    lab = np.zeros(max(n, 1), np.float32)
    nc = lib().apdo_dbscan_labels(_p(xyz, C.c_float), n, C.c_double(eps), int(min_pts), int(min_cluster), int(max_cluster), _p(lab, C.c_float))
    return lab[:n].copy(), int(nc)


def radius_outlier_mask(xyz, radius=2.0, min_pts=2):
    """pcl::RadiusOutlierRemoval (preprocessing_nodelet_ntu.cpp:163-171): boolean keep mask."""
    xyz = _f32(xyz)
    n = xyz.shape[0]
    keep = np.zeros(max(n, 1), np.uint8)
    lib().apdo_radius_outlier_mask(_p(xyz, C.c_float), n, C.c_double(radius), int(min_pts), _p(keep, C.c_ubyte))
    return keep[:n].astype(bool)


def statistical_outlier_mask(xyz, mean_k=20, stddev_mul=1.0):
    """pcl::StatisticalOutlierRemoval (preprocessing_nodelet_ntu.cpp:153-162): (boolean keep mask, per-point mean neighbour distances)."""
    xyz = _f32(xyz)
    n = xyz.shape[0]
    keep = np.zeros(max(n, 1), np.uint8)
    dist = np.zeros(max(n, 1), np.float32)
    rc = lib().apdo_statistical_outlier_mask(_p(xyz, C.c_float), n, int(mean_k), C.c_double(stddev_mul), _p(keep, C.c_ubyte), _p(dist, C.c_float))
    if rc < 0:
        raise ValueError("statistical_outlier_mask: n must exceed mean_k (and mean_k <= 63)")
    return keep[:n].astype(bool), dist[:n].copy()


class ReveConfig(C.Structure):
    """apdo_reve_config == RadarEgoVelocityEstimatorConfig (radar_ego_velocity_estimator.h:30-60), the fields the estimator reads."""
    _fields_ = [(k, C.c_float) for k in (
        "min_dist", "max_dist", "min_db", "elevation_thresh_deg", "azimuth_thresh_deg", "doppler_velocity_correction_factor", "thresh_zero_velocity",
        "allowed_outlier_percentage", "sigma_zero_velocity_x", "sigma_zero_velocity_y", "sigma_zero_velocity_z", "sigma_offset_radar_x", "sigma_offset_radar_y",
        "sigma_offset_radar_z", "max_sigma_x", "max_sigma_y", "max_sigma_z", "inlier_thresh")] + [("use_ransac", C.c_int), ("n_ransac_points", C.c_int)]


def reve_default_config(**kw):
    c = ReveConfig()
    lib().apdo_reve_default_config(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def reve_estimate(targets, samples, cfg=None):
    """RadarEgoVelocityEstimator::estimate (radar_ego_velocity_estimator.cpp:60-170).  targets [n,5] = x y z intensity doppler;
    samples [n_iter, N_ransac_points] uint32 indices into the valid targets.  Returns dict(success, v_r, sigma_v_r, inlier, outlier, n_valid, zero_velocity)."""
    cfg = cfg or reve_default_config()
    t = _f32(targets)
    s = np.ascontiguousarray(samples, np.uint32)
    n = t.shape[0]
    v, sg = np.zeros(3), np.zeros(3)
    inl, outl = np.zeros(max(n, 1), np.uint8), np.zeros(max(n, 1), np.uint8)
    nv, zv = C.c_int(0), C.c_int(0)
    ok = lib().apdo_reve_estimate(_p(t, C.c_float), n, C.byref(cfg), _p(s, C.c_uint), s.shape[0], _p(v, C.c_double), _p(sg, C.c_double), _p(inl, C.c_ubyte), _p(outl, C.c_ubyte),
                                  C.byref(nv), C.byref(zv))
    return dict(success=bool(ok), v_r=v, sigma_v_r=sg, inlier=inl[:n].astype(bool), outlier=outl[:n].astype(bool), n_valid=nv.value, zero_velocity=bool(zv.value))
