"""ctypes binding of include/gorio_prep.h: preprocessing steps that feed the hot path, on the GPU (no numerics here, no CPU fallback)."""
import ctypes as C

import numpy as np

from .apd import GorioError, load_library

PREP_SYMBOLS = ["gorio_prep_dbscan_labels", "gorio_prep_last_error"]


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
