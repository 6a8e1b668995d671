"""ctypes binding of include/gorio_ugpm.h (UGPM GP pre-integration on MI355X).

`VelPreintegration` mirrors ugpm::VelPreintegration (VelInt/preint.h:22-82): construct with the IMU data, start time, inference
times, options and bias prior; read results with get(...).  `ugpm_preint_batch` is the batched form bench.py uses.  All numerics
run in libgorio_amd.so on the GPU; nothing here computes and there is no CPU fallback.
"""
import ctypes as C

import numpy as np

from .apd import GorioError, load_library

LPM, UGPM = 0, 1
REC = 83

UGPM_SYMBOLS = ["gorio_ugpm_default_window", "gorio_ugpm_preint_batch", "gorio_ugpm_last_error", "gorio_ugpm_get_stage_times", "gorio_ugpm_debug_set_schedule",
                "gorio_ugpm_combine_preints"]


class UgpmWindow(C.Structure):
    """gorio_ugpm_window (include/gorio_ugpm.h)."""

    _fields_ = [
        ("gyr_t", C.c_void_p), ("gyr", C.c_void_p), ("n_gyr", C.c_int),
        ("vel_t", C.c_void_p), ("vel", C.c_void_p), ("n_vel", C.c_int),
        ("gyr_var", C.c_double), ("vel_var", C.c_double), ("start_t", C.c_double),
        ("infer_t", C.c_void_p), ("n_infer", C.c_int), ("type", C.c_int),
        ("min_freq", C.c_double), ("quantum", C.c_double), ("state_freq", C.c_double),
        ("correlate", C.c_int), ("overlap", C.c_int),
        ("gyr_bias", C.c_double * 3), ("vel_bias", C.c_double * 3),
        ("vel_bias_std", C.c_double), ("gyr_bias_std", C.c_double),
        ("group_sizes", C.c_void_p), ("n_groups", C.c_int),
    ]


class UgpmDiag(C.Structure):
    _fields_ = [("nb_state", C.c_int), ("nb_gyr", C.c_int), ("nb_vel", C.c_int), ("iters_rot", C.c_int), ("iters_vel", C.c_int), ("status", C.c_int),
                ("cost_rot", C.c_double), ("cost_vel", C.c_double), ("state_freq", C.c_double)]


def _dp(a):
    """Address of a contiguous float64 array (ndarray.ctypes is slow once torch is imported: ~40 us per access)."""
    return a.__array_interface__["data"][0]


def unpack(rec):
    rec = np.asarray(rec)
    return dict(
        delta_R=rec[0:9].reshape(3, 3).copy(), delta_p=rec[9:12].copy(), dt=float(rec[12]), dt_sq_half=float(rec[13]),
        cov=rec[14:50].reshape(6, 6).copy(), d_delta_R_d_bw=rec[50:59].reshape(3, 3).copy(), d_delta_R_d_t=rec[59:62].copy(),
        d_delta_p_d_bw=rec[62:71].reshape(3, 3).copy(), d_delta_p_d_bv=rec[71:80].reshape(3, 3).copy(), d_delta_p_d_t=rec[80:83].copy())


class UgpmBatch:
    """A batch of windows marshalled once for gorio_ugpm_preint_batch (the struct array and the contiguous input arrays are kept
    alive here).  run() is one call through the C ABI: everything on the device side, the host preparation of the library and
    both transfers, happens inside it, every time."""

    def __init__(self, windows, device=0, infer_t=None, type=UGPM, state_freq=50.0, correlate=True, overlap=8, quantum=-1.0, min_freq=500.0,
                 gyr_bias=None, vel_bias=None, vel_bias_std=0.0, gyr_bias_std=0.0, groups=None):
        lib = load_library()
        lib.gorio_ugpm_last_error.restype = C.c_char_p
        n = len(windows)
        self.n, self.device = n, int(device)
        self.arr = (UgpmWindow * n)()
        self.keep = []
        self.counts = []
        for i, w in enumerate(windows):
            lib.gorio_ugpm_default_window(C.byref(self.arr[i]))
            gt = np.ascontiguousarray(w["gyr_t"], np.float64)
            g = np.ascontiguousarray(w["gyr"], np.float64)
            vt = np.ascontiguousarray(w["vel_t"], np.float64)
            v = np.ascontiguousarray(w["vel"], np.float64)
            q = np.ascontiguousarray([w["end_t"]] if infer_t is None else infer_t[i], np.float64)
            self.keep += [gt, g, vt, v, q]
            a = self.arr[i]
            a.gyr_t, a.gyr, a.n_gyr = _dp(gt), _dp(g), len(gt)
            a.vel_t, a.vel, a.n_vel = _dp(vt), _dp(v), len(vt)
            a.gyr_var, a.vel_var, a.start_t = w["gyr_var"], w["vel_var"], w["start_t"]
            a.infer_t, a.n_infer = _dp(q), len(q)
            a.type, a.min_freq, a.state_freq = int(type), min_freq, state_freq
            a.quantum = float(quantum[i]) if np.ndim(quantum) > 0 else float(quantum)  # per-window list or one value; > 0 = chunked mode
            a.correlate, a.overlap = int(bool(correlate)), int(overlap)
            for k in range(3):
                a.gyr_bias[k] = 0.0 if gyr_bias is None else float(gyr_bias[k])
                a.vel_bias[k] = 0.0 if vel_bias is None else float(vel_bias[k])
            a.vel_bias_std, a.gyr_bias_std = vel_bias_std, gyr_bias_std
            if groups is not None and groups[i] is not None:  # lengths of the inner vectors of a vector<vector<double>> infer_t
                gs = np.ascontiguousarray(groups[i], np.int32)
                assert int(gs.sum()) == len(q)
                self.keep.append(gs)
                a.group_sizes, a.n_groups = gs.__array_interface__["data"][0], len(gs)
            self.counts.append(len(q))
        self.out = np.zeros((sum(self.counts), REC))
        self.diag = (UgpmDiag * n)()

    def run(self):
        """Returns the raw record array [sum(n_infer), 83] (see unpack)."""
        lib = load_library()
        rc = lib.gorio_ugpm_preint_batch(self.arr, self.n, C.c_void_p(_dp(self.out)), self.diag, self.device)
        if rc < 0:
            msg = lib.gorio_ugpm_last_error()
            raise GorioError(rc, msg.decode() if msg else "")
        return self.out

    def results(self):
        res, k = [], 0
        for cnt in self.counts:
            res.append([unpack(self.out[k + j]) for j in range(cnt)])
            k += cnt
        return res

    def diagnostics(self):
        return [dict(nb_state=d.nb_state, nb_gyr=d.nb_gyr, nb_vel=d.nb_vel, iters_rot=d.iters_rot, iters_vel=d.iters_vel, status=d.status,
                     cost_rot=d.cost_rot, cost_vel=d.cost_vel, state_freq=d.state_freq) for d in self.diag]


def ugpm_preint_batch(windows, device=0, infer_t=None, type=UGPM, state_freq=50.0, correlate=True, overlap=8, quantum=-1.0, min_freq=500.0,
                      gyr_bias=None, vel_bias=None, vel_bias_std=0.0, gyr_bias_std=0.0, return_diag=False, groups=None):
    """gorio_ugpm_preint_batch over a list of window dicts (as made by synth.imu_window).  `infer_t`: None (each window's end_t) or a
    list of per-window arrays.  Returns a list (per window) of lists (per inference time) of PreintMeas dicts."""
    b = UgpmBatch(windows, device=device, infer_t=infer_t, type=type, state_freq=state_freq, correlate=correlate, overlap=overlap, quantum=quantum,
                  min_freq=min_freq, gyr_bias=gyr_bias, vel_bias=vel_bias, vel_bias_std=vel_bias_std, gyr_bias_std=gyr_bias_std, groups=groups)
    b.run()
    if return_diag:
        return b.results(), b.diagnostics()
    return b.results()


def pack(m):
    """The gorio_ugpm_meas record (83 doubles) of an unpacked PreintMeas dict."""
    return np.concatenate([np.asarray(m["delta_R"], np.float64).ravel(), np.asarray(m["delta_p"], np.float64), [m["dt"], m["dt_sq_half"]],
                           np.asarray(m["cov"], np.float64).ravel(), np.asarray(m["d_delta_R_d_bw"], np.float64).ravel(), np.asarray(m["d_delta_R_d_t"], np.float64),
                           np.asarray(m["d_delta_p_d_bw"], np.float64).ravel(), np.asarray(m["d_delta_p_d_bv"], np.float64).ravel(), np.asarray(m["d_delta_p_d_t"], np.float64)])


def ugpm_combine_preints(prev, cur):
    """gorio_ugpm_combine_preints = ugpm::combinePreints (VelInt/math_utils.h:689-726) on two PreintMeas dicts (host arithmetic)."""
    lib = load_library()
    a, b, out = np.ascontiguousarray(pack(prev)), np.ascontiguousarray(pack(cur)), np.zeros(REC)
    lib.gorio_ugpm_combine_preints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    rc = lib.gorio_ugpm_combine_preints(C.c_void_p(_dp(a)), C.c_void_p(_dp(b)), C.c_void_p(_dp(out)))
    if rc < 0:
        raise GorioError(rc, lib.gorio_ugpm_last_error().decode())
    return unpack(out)


def ugpm_debug_set_schedule(speculative_rot=True):
    """gorio_ugpm_debug_set_schedule (test hook, process-wide): the rotation fit as three launches per iteration (default) or four."""
    lib = load_library()
    lib.gorio_ugpm_debug_set_schedule.argtypes = [C.c_int]
    lib.gorio_ugpm_debug_set_schedule.restype = None
    lib.gorio_ugpm_debug_set_schedule(1 if speculative_rot else 0)


def ugpm_stage_times():
    lib = load_library()
    s = (C.c_double * 8)()
    c = (C.c_int * 8)()
    lib.gorio_ugpm_get_stage_times(s, c)
    return list(s), list(c)


class PreintOption:
    """ugpm::PreintOption (VelInt/types.h:285-292)."""

    def __init__(self, min_freq=500.0, type=UGPM, quantum=-1.0, state_freq=50.0, correlate=True):
        self.min_freq, self.type, self.quantum, self.state_freq, self.correlate = min_freq, type, quantum, state_freq, correlate


class PreintPrior:
    """ugpm::PreintPrior (VelInt/types.h:294-298)."""

    def __init__(self, vel_bias=(0.0, 0.0, 0.0), gyr_bias=(0.0, 0.0, 0.0)):
        self.vel_bias, self.gyr_bias = list(vel_bias), list(gyr_bias)


class VelPreintegration:
    """ugpm::VelPreintegration (VelInt/preint.h:28-64).  infer_t: float, 1-D list or list of lists, as the three constructors."""

    def __init__(self, imu_data, start_t, infer_t, opt=None, prior=None, rot_only=False, overlap=8, device=0):
        opt = opt or PreintOption()
        prior = prior or PreintPrior()
        if np.isscalar(infer_t):
            self._kind, groups = "single", [[float(infer_t)]]
        elif len(infer_t) and np.isscalar(infer_t[0]):
            self._kind, groups = "vec", [list(map(float, infer_t))]
        else:
            self._kind, groups = "vecvec", [list(map(float, g)) for g in infer_t]
        flat = [t for g in groups for t in g]
        win = dict(imu_data)
        win["start_t"] = start_t
        self._args = dict(device=device, infer_t=[flat], type=opt.type, state_freq=opt.state_freq, correlate=opt.correlate, overlap=overlap,
                          quantum=opt.quantum, min_freq=opt.min_freq, gyr_bias=prior.gyr_bias, vel_bias=prior.vel_bias,
                          groups=[[len(g) for g in groups]])
        self._win = win
        self._groups = groups
        self._prior = prior
        self._cache = {}

    def _compute(self, vel_bias_std, gyr_bias_std):
        key = (vel_bias_std, gyr_bias_std)
        if key not in self._cache:
            flat = ugpm_preint_batch([self._win], vel_bias_std=vel_bias_std, gyr_bias_std=gyr_bias_std, **self._args)[0]
            out, k = [], 0
            for g in self._groups:
                out.append(flat[k:k + len(g)])
                k += len(g)
            self._cache[key] = out
        return self._cache[key]

    def get(self, *idx, vel_bias_std=0.3, gyr_bias_std=0.03):
        pre = self._compute(vel_bias_std, gyr_bias_std)
        if len(idx) == 2:
            i, j = idx
        elif len(idx) == 1:
            if self._kind != "vec":
                raise IndexError("VelPreintegration::get: The type of query does not math the type of constructor")  # preint.h:1772
            i, j = 0, idx[0]
        else:
            if self._kind != "single":
                raise IndexError("VelPreintegration::get: The type of query does not math the type of constructor")  # preint.h:1780
            i, j = 0, 0
        if not (0 <= i < len(pre) and 0 <= j < len(pre[i])):
            raise IndexError("VelPreintegration::get: Trying to get precomputed preintegrated measurements (wrong index query?)")  # preint.h:1762
        return pre[i][j]

    def getPrior(self):  # noqa: N802
        return self._prior
