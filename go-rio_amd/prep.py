"""ctypes binding of include/gorio_prep.h: preprocessing steps that feed the hot path, on the GPU (no numerics here, no CPU fallback)."""
import ctypes as C

import numpy as np

from .apd import GorioError, load_library

PREP_SYMBOLS = ["gorio_prep_dbscan_labels", "gorio_prep_radius_outlier_mask", "gorio_prep_statistical_outlier_mask", "gorio_prep_voxel_downsample", "gorio_prep_last_error", "gorio_prep_reve_default_config", "gorio_prep_reve_ransac_iterations", "gorio_prep_ego_velocity"]


def dbscan_labels(xyz, eps=0.9, core_min_pts=10, min_cluster_size=20, max_cluster_size=25000, device=0):
    """preprocessing_nodelet_ntu.cpp:518-568 with the nodelet's parameters as defaults: returns (labels float32 [n], n_clusters)."""
    lib = load_library()
    lib.gorio_prep_last_error.restype = C.c_char_p
    xyz = np.ascontiguousarray(xyz, np.float32)
    if xyz.ndim != 2 or xyz.shape[1] != 3:
        raise ValueError("xyz must be [n, 3]")
    n = xyz.shape[0]
    lab = np.zeros(n, np.float32)
    nc = C.c_int(0)
    rc = lib.gorio_prep_dbscan_labels(int(device), C.c_void_p(xyz.__array_interface__["data"][0]), n, 12, C.c_double(eps), int(core_min_pts), int(min_cluster_size),
                                      int(max_cluster_size), C.c_void_p(lab.__array_interface__["data"][0]), 4, C.byref(nc))
    if rc < 0:
        msg = lib.gorio_prep_last_error()
        raise GorioError(rc, msg.decode() if msg else "")
    return lab, nc.value


def voxel_downsample(xyz, leaf=0.1, device=0):
    """pcl::VoxelGrid (preprocessing_nodelet_ntu.cpp:137-139, 608-622): centroids [m,3], ordered by voxel index."""
    lib = load_library()
    lib.gorio_prep_last_error.restype = C.c_char_p
    xyz = np.ascontiguousarray(xyz, np.float32)
    n = xyz.shape[0]
    out = np.zeros((n, 3), np.float32)
    m = C.c_int(0)
    rc = lib.gorio_prep_voxel_downsample(int(device), C.c_void_p(xyz.__array_interface__["data"][0]), n, 12, C.c_double(leaf), C.c_void_p(out.__array_interface__["data"][0]), 12, n, C.byref(m))
    if rc < 0:
        msg = lib.gorio_prep_last_error()
        raise GorioError(rc, msg.decode() if msg else "")
    return out[:m.value].copy()


def radius_outlier_mask(xyz, radius=2.0, min_neighbors=2, device=0):
    """pcl::RadiusOutlierRemoval (preprocessing_nodelet_ntu.cpp:163-171): boolean keep mask [n]."""
    lib = load_library()
    lib.gorio_prep_last_error.restype = C.c_char_p
    xyz = np.ascontiguousarray(xyz, np.float32)
    if xyz.ndim != 2 or xyz.shape[1] != 3:
        raise ValueError("xyz must be [n, 3]")
    n = xyz.shape[0]
    keep = np.zeros(n, np.uint8)
    nk = C.c_int(0)
    rc = lib.gorio_prep_radius_outlier_mask(int(device), C.c_void_p(xyz.__array_interface__["data"][0]), n, 12, C.c_double(radius), int(min_neighbors),
                                            C.c_void_p(keep.__array_interface__["data"][0]), C.byref(nk))
    if rc < 0:
        msg = lib.gorio_prep_last_error()
        raise GorioError(rc, msg.decode() if msg else "")
    return keep.astype(bool)


def statistical_outlier_mask(xyz, mean_k=20, stddev_mul=1.0, device=0, return_distances=False):
    """pcl::StatisticalOutlierRemoval (preprocessing_nodelet_ntu.cpp:153-162, the nodelet's default filter): boolean keep mask [n]
    (and, on request, the per-point mean neighbour distances)."""
    lib = load_library()
    lib.gorio_prep_last_error.restype = C.c_char_p
    xyz = np.ascontiguousarray(xyz, np.float32)
    if xyz.ndim != 2 or xyz.shape[1] != 3:
        raise ValueError("xyz must be [n, 3]")
    n = xyz.shape[0]
    keep = np.zeros(n, np.uint8)
    dist = np.zeros(n, np.float32)
    nk = C.c_int(0)
    rc = lib.gorio_prep_statistical_outlier_mask(int(device), C.c_void_p(xyz.__array_interface__["data"][0]), n, 12, int(mean_k), C.c_double(stddev_mul),
                                                 C.c_void_p(keep.__array_interface__["data"][0]), C.byref(nk), C.c_void_p(dist.__array_interface__["data"][0]))
    if rc < 0:
        msg = lib.gorio_prep_last_error()
        raise GorioError(rc, msg.decode() if msg else "")
    return (keep.astype(bool), dist) if return_distances else keep.astype(bool)


class ReveConfig(C.Structure):
    """gorio_reve_config (include/gorio_prep.h) == RadarEgoVelocityEstimatorConfig (radar_ego_velocity_estimator.h:30-60)."""
    _fields_ = [(k, C.c_float) for k in (
        "min_dist", "max_dist", "min_db", "elevation_thresh_deg", "azimuth_thresh_deg", "doppler_velocity_correction_factor", "thresh_zero_velocity",
        "allowed_outlier_percentage", "sigma_zero_velocity_x", "sigma_zero_velocity_y", "sigma_zero_velocity_z", "sigma_offset_radar_x", "sigma_offset_radar_y",
        "sigma_offset_radar_z", "max_sigma_x", "max_sigma_y", "max_sigma_z", "inlier_thresh")] + [("use_ransac", C.c_int), ("n_ransac_points", C.c_int),
                                                                                                   ("outlier_prob", C.c_float), ("success_prob", C.c_float)]


def reve_default_config(**kw):
    c = ReveConfig()
    load_library().gorio_prep_reve_default_config(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def reve_ransac_iterations(cfg=None):
    return load_library().gorio_prep_reve_ransac_iterations(C.byref(cfg or reve_default_config()))


def ego_velocity(targets, samples, cfg=None, device=0):
    """RadarEgoVelocityEstimator::estimate on the GPU.  targets [n,5] float32 = x y z intensity doppler; samples [n_iter, N] uint32."""
    lib = load_library()
    lib.gorio_prep_last_error.restype = C.c_char_p
    cfg = cfg or reve_default_config()
    t = np.ascontiguousarray(targets, np.float32)
    s = np.ascontiguousarray(samples, np.uint32).reshape(-1, cfg.n_ransac_points) if len(samples) else np.zeros((0, cfg.n_ransac_points), np.uint32)
    n = t.shape[0]
    base = t.__array_interface__["data"][0]
    v, sg = np.zeros(3), np.zeros(3)
    inl, outl = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    nv, zv, ok = C.c_int(0), C.c_int(0), C.c_int(0)
    rc = lib.gorio_prep_ego_velocity(int(device), C.c_void_p(base), C.c_void_p(base + 12), C.c_void_p(base + 16), n, 20, C.byref(cfg),
                                     C.c_void_p(s.__array_interface__["data"][0]) if s.shape[0] else None, s.shape[0], C.c_void_p(v.__array_interface__["data"][0]),
                                     C.c_void_p(sg.__array_interface__["data"][0]), C.c_void_p(inl.__array_interface__["data"][0]), C.c_void_p(outl.__array_interface__["data"][0]),
                                     C.byref(nv), C.byref(zv), C.byref(ok))
    if rc < 0:
        msg = lib.gorio_prep_last_error()
        raise GorioError(rc, msg.decode() if msg else "")
    return dict(success=bool(ok.value), v_r=v, sigma_v_r=sg, inlier=inl.astype(bool), outlier=outl.astype(bool), n_valid=nv.value, zero_velocity=bool(zv.value))
