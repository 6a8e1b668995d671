"""ctypes binding of include/gorio_apd.h (APD-GICP scan matching on MI355X).

`ApdGicp` mirrors the public surface of fast_gicp::FastAPDGICP (fast_apdgicp.hpp:48-79 + the pcl::Registration calls the
nodelets make), one method per reference member, so parity tests read like the reference's own gicp_test.cpp.  All numerics
happen in libgorio_amd.so on the GPU; nothing here computes.
"""
import ctypes as C
import os

import numpy as np

REG_NONE, REG_MIN_EIG, REG_NORMALIZED_MIN_EIG, REG_PLANE, REG_FROBENIUS = range(5)
OPT_GAUSS_NEWTON, OPT_LEVENBERG_MARQUARDT = 0, 1
SEARCH_BRUTE_FORCE, SEARCH_PRUNED = 0, 1


class GorioError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gorio error {code}: {msg}")
        self.code = code


class ApdParams(C.Structure):
    """gorio_apd_params (include/gorio_apd.h)."""

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
        ("search", C.c_int),
        ("cl_weight_points", C.c_int),
        ("keep_knn_indices", C.c_int),
    ]


# every symbol include/gorio_apd.h declares (tests check that the library exports all of them)
APD_SYMBOLS = [
    "gorio_apd_create", "gorio_apd_destroy", "gorio_apd_last_error", "gorio_apd_default_params", "gorio_apd_set_params",
    "gorio_apd_get_params", "gorio_apd_set_source", "gorio_apd_set_target", "gorio_apd_set_source_device",
    "gorio_apd_set_target_device", "gorio_apd_set_clouds_device_batch", "gorio_apd_clear_source", "gorio_apd_clear_target", "gorio_apd_swap_source_and_target",
    "gorio_apd_set_source_covariances", "gorio_apd_set_target_covariances", "gorio_apd_get_source_covariances",
    "gorio_apd_get_target_covariances", "gorio_apd_calculate_covariances", "gorio_apd_get_knn_indices", "gorio_apd_align",
    "gorio_apd_align_batch", "gorio_apd_linearize", "gorio_apd_compute_error", "gorio_apd_get_correspondences",
    "gorio_apd_get_mahalanobis", "gorio_apd_transform_source", "gorio_apd_fitness_score", "gorio_apd_set_profiling",
    "gorio_apd_get_stage_times", "gorio_apd_set_target_shared", "gorio_comm_get_unique_id", "gorio_apd_comm_init", "gorio_apd_comm_destroy", "gorio_apd_comm_info", "gorio_apd_debug_set_shard", "gorio_apd_debug_set_schedule", "gorio_apd_set_target_submap", "gorio_apd_get_target_points",
]

_lib = None


def load_library():
    """dlopen go-rio_amd/lib/libgorio_amd.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        from . import LIB_PATH

        if not os.path.exists(LIB_PATH):
            raise GorioError(-2, f"{LIB_PATH} is missing: run __graft_entry__.build() (no CPU fallback exists)")
        _lib = C.CDLL(LIB_PATH)
        _lib.gorio_apd_last_error.restype = C.c_char_p
    return _lib


def _p(a, t):
    """Typed pointer to a contiguous array without going through ndarray.ctypes (slow once torch is imported)."""
    return C.cast(a.__array_interface__["data"][0], C.POINTER(t))


def _check(h, rc):
    if rc < 0:
        msg = load_library().gorio_apd_last_error(h)
        raise GorioError(rc, msg.decode() if msg else "")
    return rc


class ApdGicp:
    """One fast_gicp::FastAPDGICP object living on one GPU."""

    def __init__(self, device=0, **params):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.gorio_apd_create(C.byref(self._h), int(device))
        if rc != 0:
            raise GorioError(rc, "gorio_apd_create failed (no usable HIP device?)")
        self.params = ApdParams()
        self._lib.gorio_apd_default_params(C.byref(self.params))
        self._n_src = self._n_tgt = 0
        if params:
            self.set_params(**params)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.gorio_apd_destroy(h)
            self._h = None

    # ---- setters of fast_apdgicp.hpp:48-58, lsq_registration.hpp:51-53 and the pcl::Registration setters (registrations.cpp:41-48)
    def set_params(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.params, k):
                raise AttributeError(k)
            setattr(self.params, k, v)
        _check(self._h, self._lib.gorio_apd_set_params(self._h, C.byref(self.params)))

    def setNumThreads(self, n):  # noqa: N802 -- no-op on the GPU (APD:34-42)
        pass

    def setCorrespondenceRandomness(self, k):  # noqa: N802
        self.set_params(k_correspondences=int(k))

    def setRegularizationMethod(self, method):  # noqa: N802
        self.set_params(regularization=int(method))

    def setAzimuthVar(self, v):  # noqa: N802
        self.set_params(azimuth_var=float(v))

    def setElevationVar(self, v):  # noqa: N802
        self.set_params(elevation_var=float(v))

    def setDistVar(self, v):  # noqa: N802
        self.set_params(dist_var=float(v))

    def setTransformationEpsilon(self, v):  # noqa: N802
        self.set_params(transformation_epsilon=float(v))

    def setRotationEpsilon(self, v):  # noqa: N802
        self.set_params(rotation_epsilon=float(v))

    def setMaximumIterations(self, v):  # noqa: N802
        self.set_params(max_iterations=int(v))

    def setMaxCorrespondenceDistance(self, v):  # noqa: N802
        self.set_params(corr_dist_threshold=float(v))

    def setInitialLambdaFactor(self, v):  # noqa: N802
        self.set_params(lm_init_lambda_factor=float(v))

    # ---- clouds
    @staticmethod
    def _cloud_args(xyz, label):
        xyz = np.ascontiguousarray(xyz, np.float32)
        if xyz.ndim != 2 or xyz.shape[1] != 3:
            raise ValueError("xyz must be [n, 3]")
        lab = None if label is None else np.ascontiguousarray(label, np.float32)
        return xyz, lab

    def setInputSource(self, xyz, label=None):  # noqa: N802
        xyz, lab = self._cloud_args(xyz, label)
        # labels are packed with their own 4-byte stride: pass a [n,4] interleaved buffer so one stride serves both
        buf = np.empty((xyz.shape[0], 4), np.float32)
        buf[:, :3] = xyz
        buf[:, 3] = 0.0 if lab is None else lab
        _check(self._h, self._lib.gorio_apd_set_source(self._h, _p(buf, C.c_float), _p(buf[:, 3:], C.c_float) if lab is not None else None, buf.shape[0], 16))
        self._n_src = buf.shape[0]

    def setInputTarget(self, xyz, label=None):  # noqa: N802
        xyz, lab = self._cloud_args(xyz, label)
        buf = np.empty((xyz.shape[0], 4), np.float32)
        buf[:, :3] = xyz
        buf[:, 3] = 0.0 if lab is None else lab
        _check(self._h, self._lib.gorio_apd_set_target(self._h, _p(buf, C.c_float), _p(buf[:, 3:], C.c_float) if lab is not None else None, buf.shape[0], 16))
        self._n_tgt = buf.shape[0]

    def setInputSourcePcl(self, points):  # noqa: N802
        """points: [n, 12] float32 laid out as pcl::PointXYZINormal (48 bytes: x y z 1 | normal_x normal_y normal_z 0 | intensity curvature - -):
        exactly the pointers and stride the C++ drop-in passes (&pts[0].x, &pts[0].normal_x, sizeof(PointT)); no repacking on the way."""
        pts = np.ascontiguousarray(points, np.float32)
        if pts.ndim != 2 or pts.shape[1] != 12:
            raise ValueError("points must be [n, 12] float32 (PointXYZINormal)")
        _check(self._h, self._lib.gorio_apd_set_source(self._h, _p(pts, C.c_float), _p(pts[:, 4:], C.c_float), pts.shape[0], 48))
        self._n_src = pts.shape[0]

    def setInputTargetPcl(self, points):  # noqa: N802
        pts = np.ascontiguousarray(points, np.float32)
        if pts.ndim != 2 or pts.shape[1] != 12:
            raise ValueError("points must be [n, 12] float32 (PointXYZINormal)")
        _check(self._h, self._lib.gorio_apd_set_target(self._h, _p(pts, C.c_float), _p(pts[:, 4:], C.c_float), pts.shape[0], 48))
        self._n_tgt = pts.shape[0]

    def setInputSourceDevice(self, d_x, d_y, d_z, d_label, n):  # noqa: N802 -- raw device pointers (ints)
        _check(self._h, self._lib.gorio_apd_set_source_device(self._h, C.c_void_p(d_x), C.c_void_p(d_y), C.c_void_p(d_z), C.c_void_p(d_label or 0), int(n)))
        self._n_src = int(n)

    def setInputTargetDevice(self, d_x, d_y, d_z, d_label, n):  # noqa: N802
        _check(self._h, self._lib.gorio_apd_set_target_device(self._h, C.c_void_p(d_x), C.c_void_p(d_y), C.c_void_p(d_z), C.c_void_p(d_label or 0), int(n)))
        self._n_tgt = int(n)

    def setInputTargetShared(self, owner):  # noqa: N802 -- gorio_apd_set_target_shared: one device copy of the map for many objects
        _check(self._h, self._lib.gorio_apd_set_target_shared(self._h, owner._h))
        self._n_tgt = owner._n_tgt

    def setInputTargetSubmap(self, frames, rel_poses, voxel_leaf=0.0):  # noqa: N802
        """gorio_apd_set_target_submap: frames = list of (xyz [n,3], label [n] or None), rel_poses = list of 4x4 (odom_i^-1 odom_newest)."""
        n = len(frames)
        arr = (Keyframe * n)()
        keep = []
        for i, ((xyz, lab), T) in enumerate(zip(frames, rel_poses)):
            xyz, lab = self._cloud_args(xyz, lab)
            buf = np.empty((xyz.shape[0], 4), np.float32)
            buf[:, :3] = xyz
            buf[:, 3] = 0.0 if lab is None else lab
            Td = np.ascontiguousarray(T, np.float64)
            keep += [buf, Td]
            arr[i].xyz = buf.__array_interface__["data"][0]
            arr[i].label = buf.__array_interface__["data"][0] + 12
            arr[i].n, arr[i].point_stride_bytes = buf.shape[0], 16
            arr[i].rel_pose = Td.__array_interface__["data"][0]
        cnt = C.c_int(0)
        _check(self._h, self._lib.gorio_apd_set_target_submap(self._h, arr, n, C.c_double(voxel_leaf), C.byref(cnt)))
        self._n_tgt = cnt.value
        return cnt.value

    def getTargetPoints(self):  # noqa: N802
        buf = np.empty((self._n_tgt, 4), np.float32)
        _check(self._h, self._lib.gorio_apd_get_target_points(self._h, _p(buf, C.c_float), _p(buf[:, 3:], C.c_float), self._n_tgt, 16))
        return buf[:, :3].copy(), buf[:, 3].copy()

    # ---- sharded-source mode (RCCL): see include/gorio_apd.h
    @staticmethod
    def commUniqueId():  # noqa: N802
        buf = C.create_string_buffer(128)
        rc = load_library().gorio_comm_get_unique_id(buf)
        if rc != 0:
            raise GorioError(rc, "gorio_comm_get_unique_id failed (librccl missing?)")
        return buf.raw

    def commInit(self, world_size, rank, unique_id):  # noqa: N802
        _check(self._h, self._lib.gorio_apd_comm_init(self._h, int(world_size), int(rank), C.c_char_p(bytes(unique_id))))

    def commInfo(self):  # noqa: N802 -- (world size, rank) as RCCL reports them, ncclAllReduce calls enqueued so far
        w, r, c = C.c_int(0), C.c_int(0), C.c_longlong(0)
        _check(self._h, self._lib.gorio_apd_comm_info(self._h, C.byref(w), C.byref(r), C.byref(c)))
        return w.value, r.value, c.value

    def debugSetShard(self, world_size, rank):  # noqa: N802 -- test hook: the partition of a rank without the collectives
        _check(self._h, self._lib.gorio_apd_debug_set_shard(self._h, int(world_size), int(rank)))

    def debugSetSchedule(self, fuse_step=True, plan_search=True):  # noqa: N802 -- test hook: the two schedule optimisations of a GN align
        _check(self._h, self._lib.gorio_apd_debug_set_schedule(self._h, int(bool(fuse_step)), int(bool(plan_search))))

    def commDestroy(self):  # noqa: N802
        _check(self._h, self._lib.gorio_apd_comm_destroy(self._h))

    def clearSource(self):  # noqa: N802
        _check(self._h, self._lib.gorio_apd_clear_source(self._h))
        self._n_src = 0

    def clearTarget(self):  # noqa: N802
        _check(self._h, self._lib.gorio_apd_clear_target(self._h))
        self._n_tgt = 0

    def swapSourceAndTarget(self):  # noqa: N802
        _check(self._h, self._lib.gorio_apd_swap_source_and_target(self._h))
        self._n_src, self._n_tgt = self._n_tgt, self._n_src

    # ---- covariances
    def setSourceCovariances(self, cov):  # noqa: N802
        cov = np.ascontiguousarray(cov, np.float64).reshape(-1, 16)
        _check(self._h, self._lib.gorio_apd_set_source_covariances(self._h, _p(cov, C.c_double), cov.shape[0]))

    def setTargetCovariances(self, cov):  # noqa: N802
        cov = np.ascontiguousarray(cov, np.float64).reshape(-1, 16)
        _check(self._h, self._lib.gorio_apd_set_target_covariances(self._h, _p(cov, C.c_double), cov.shape[0]))

    def _get_covs(self, fn, n):
        cnt = _check(self._h, fn(self._h, None, 0))
        out = np.zeros((cnt, 4, 4), np.float64)
        if cnt:
            _check(self._h, fn(self._h, _p(out, C.c_double), cnt))
        return out

    def getSourceCovariances(self):  # noqa: N802
        return self._get_covs(self._lib.gorio_apd_get_source_covariances, self._n_src)

    def getTargetCovariances(self):  # noqa: N802
        return self._get_covs(self._lib.gorio_apd_get_target_covariances, self._n_tgt)

    def calculateCovariances(self):  # noqa: N802
        _check(self._h, self._lib.gorio_apd_calculate_covariances(self._h))

    def getKnnIndices(self, which):  # noqa: N802
        n = self._n_src if which == 0 else self._n_tgt
        k = self.params.k_correspondences
        idx = np.empty((n, k), np.int32)
        _check(self._h, self._lib.gorio_apd_get_knn_indices(self._h, int(which), _p(idx, C.c_int), n * k))
        return idx

    # ---- registration
    def align(self, guess=None):
        """pcl::Registration::align(output, guess) minus the output cloud; returns a dict like oracle.apd.align."""
        g = np.ascontiguousarray(np.eye(4) if guess is None else guess, np.float32)
        T = np.zeros((4, 4), np.float32)
        H = np.zeros((6, 6), np.float64)
        conv, nit, nlin = C.c_int(0), C.c_int(0), C.c_int(0)
        _check(self._h, self._lib.gorio_apd_align(self._h, _p(g, C.c_float), _p(T, C.c_float), _p(H, C.c_double), C.byref(conv), C.byref(nit), C.byref(nlin)))
        self._final = T
        self._converged = bool(conv.value)
        return dict(T=T, H=H, converged=bool(conv.value), nr_iterations=nit.value, n_linearize=nlin.value)

    def hasConverged(self):  # noqa: N802
        return self._converged

    def getFinalTransformation(self):  # noqa: N802
        return self._final

    def linearize(self, T):
        T = np.ascontiguousarray(T, np.float64)
        H = np.zeros((6, 6), np.float64)
        b = np.zeros(6, np.float64)
        err = C.c_double(0.0)
        _check(self._h, self._lib.gorio_apd_linearize(self._h, _p(T, C.c_double), _p(H, C.c_double), _p(b, C.c_double), C.byref(err)))
        return err.value, H, b

    def evaluateCost(self, relative_pose):  # noqa: N802 -- lsq_registration_impl.hpp:50-52
        return self.linearize(np.asarray(relative_pose, np.float32).astype(np.float64))

    def compute_error(self, T):
        T = np.ascontiguousarray(T, np.float64)
        err = C.c_double(0.0)
        _check(self._h, self._lib.gorio_apd_compute_error(self._h, _p(T, C.c_double), C.byref(err)))
        return err.value

    def getCorrespondences(self):  # noqa: N802
        corr = np.empty(self._n_src, np.int32)
        sqd = np.empty(self._n_src, np.float32)
        _check(self._h, self._lib.gorio_apd_get_correspondences(self._h, _p(corr, C.c_int), _p(sqd, C.c_float), self._n_src))
        return corr, sqd

    def getMahalanobis(self):  # noqa: N802
        m = np.empty((self._n_src, 4, 4), np.float64)
        _check(self._h, self._lib.gorio_apd_get_mahalanobis(self._h, _p(m, C.c_double), self._n_src))
        return m

    def transformSource(self, T):  # noqa: N802
        T = np.ascontiguousarray(T, np.float32)
        out = np.empty((self._n_src, 3), np.float32)
        _check(self._h, self._lib.gorio_apd_transform_source(self._h, _p(T, C.c_float), _p(out, C.c_float), self._n_src, 12))
        return out

    def getFitnessScore(self, T=None, max_range=np.finfo(np.float64).max, inlier_dist=0.5):  # noqa: N802
        """(pcl getFitnessScore(max_range), inlier fraction of scan_matching_odometry_nodelet.cpp:677-689 with its 0.5 m bound)."""
        T = np.ascontiguousarray(self._final if T is None else T, np.float32)
        score, inl = C.c_double(0.0), C.c_double(0.0)
        _check(self._h, self._lib.gorio_apd_fitness_score(self._h, _p(T, C.c_float), C.c_double(max_range), C.c_double(inlier_dist), C.byref(score), C.byref(inl)))
        return score.value, inl.value

    def setProfiling(self, on):  # noqa: N802
        _check(self._h, self._lib.gorio_apd_set_profiling(self._h, int(bool(on))))

    def getStageTimes(self):  # noqa: N802
        s = (C.c_double * 8)()
        c = (C.c_int * 8)()
        _check(self._h, self._lib.gorio_apd_get_stage_times(self._h, s, c))
        return list(s), list(c)


class Keyframe(C.Structure):
    """gorio_apd_keyframe (include/gorio_apd.h)."""
    _fields_ = [("xyz", C.c_void_p), ("label", C.c_void_p), ("n", C.c_int), ("point_stride_bytes", C.c_int), ("rel_pose", C.c_void_p)]


class DeviceCloud(C.Structure):
    """gorio_apd_device_cloud (include/gorio_apd.h)."""
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("label", C.c_void_p), ("n", C.c_int)]


class DeviceInputs:
    """Descriptor arrays for gorio_apd_set_clouds_device_batch, built once for a list of objects whose inputs live in
    device-resident SoA buffers: sources / targets = lists of ((x, y, z, label) device addresses, n) per object, or None."""

    def __init__(self, objs, sources=None, targets=None):
        self.objs = list(objs)
        n = len(self.objs)
        self.handles = (C.c_void_p * n)(*[o._h for o in self.objs])

        def pack(lst):
            if lst is None:
                return None
            arr = (DeviceCloud * n)()
            for i, (ptrs, cnt) in enumerate(lst):
                arr[i].x, arr[i].y, arr[i].z = ptrs[0], ptrs[1], ptrs[2]
                arr[i].label = ptrs[3] if len(ptrs) > 3 and ptrs[3] else None
                arr[i].n = int(cnt)
            return arr
        self.src, self.tgt = pack(sources), pack(targets)
        self._ns = None if sources is None else [int(c) for _, c in sources]
        self._nt = None if targets is None else [int(c) for _, c in targets]

    def apply(self):
        """setInputSource / setInputTarget of every object from its device buffers: one C call, one copy launch."""
        lib = load_library()
        rc = lib.gorio_apd_set_clouds_device_batch(self.handles, len(self.objs), self.src, self.tgt)
        _check(self.objs[0]._h, rc)
        for i, o in enumerate(self.objs):
            if self._ns is not None:
                o._n_src = self._ns[i]
            if self._nt is not None:
                o._n_tgt = self._nt[i]


def align_batch(objs, guesses=None):
    """gorio_apd_align_batch over a list of ApdGicp objects (all on one device).  Returns a list of result dicts."""
    lib = load_library()
    n = len(objs)
    g = np.ascontiguousarray(np.tile(np.eye(4, dtype=np.float32), (n, 1, 1)) if guesses is None else guesses, np.float32).reshape(n, 16)
    T = np.zeros((n, 4, 4), np.float32)
    H = np.zeros((n, 6, 6), np.float64)
    conv = np.zeros(n, np.int32)
    nit = np.zeros(n, np.int32)
    nlin = np.zeros(n, np.int32)
    arr = (C.c_void_p * n)(*[o._h for o in objs])
    rc = lib.gorio_apd_align_batch(arr, n, _p(g, C.c_float), _p(T, C.c_float), _p(H, C.c_double), _p(conv, C.c_int), _p(nit, C.c_int), _p(nlin, C.c_int))
    _check(objs[0]._h, rc)
    out = []
    for q, o in enumerate(objs):
        o._final = T[q]
        o._converged = bool(conv[q])
        out.append(dict(T=T[q], H=H[q], converged=bool(conv[q]), nr_iterations=int(nit[q]), n_linearize=int(nlin[q])))
    return out
