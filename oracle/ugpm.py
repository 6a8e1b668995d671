"""ctypes binding of oracle/ugpm_oracle.cpp (CPU restatement of VelInt/preint.h, math_utils.h, cost_functions.h).

Test infrastructure only -- see oracle/__init__.py.
"""
import ctypes as C
import os

import numpy as np

from . import BUILD_DIR, build

LPM, UGPM = 0, 1
REC = 83  # doubles per PreintMeas record, see pack() in ugpm_oracle.cpp

_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(BUILD_DIR, "libugpm_oracle.so")
        if not os.path.exists(path):
            build()
        _lib = C.CDLL(path)
        _lib.ugpmo_kss_int.restype = C.c_double
        _lib.ugpmo_kss_int.argtypes = [C.c_double] * 4
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def unpack(rec):
    rec = np.asarray(rec)
    return dict(
        delta_R=rec[0:9].reshape(3, 3).copy(), delta_p=rec[9:12].copy(), dt=float(rec[12]), dt_sq_half=float(rec[13]),
        cov=rec[14:50].reshape(6, 6).copy(), d_delta_R_d_bw=rec[50:59].reshape(3, 3).copy(), d_delta_R_d_t=rec[59:62].copy(),
        d_delta_p_d_bw=rec[62:71].reshape(3, 3).copy(), d_delta_p_d_bv=rec[71:80].reshape(3, 3).copy(), d_delta_p_d_t=rec[80:83].copy())


def preintegrate(win, infer_t=None, type=UGPM, min_freq=500.0, state_freq=50.0, correlate=True, overlap=8, gyr_bias=None, vel_bias=None,
                 vel_bias_std=0.0, gyr_bias_std=0.0):
    """ugpm::VelPreintegration(imu, start_t, infer_t, opt, prior).get(0, j, vel_bias_std, gyr_bias_std) for every j.

    `win` is a dict from go-rio_amd.synth.imu_window (gyr_t, gyr, vel_t, vel, gyr_var, vel_var, start_t, end_t).
    Returns (list of unpacked PreintMeas dicts, diag dict).
    """
    gt, g = np.ascontiguousarray(win["gyr_t"], np.float64), np.ascontiguousarray(win["gyr"], np.float64)
    vt, v = np.ascontiguousarray(win["vel_t"], np.float64), np.ascontiguousarray(win["vel"], np.float64)
    q = np.ascontiguousarray([win["end_t"]] if infer_t is None else infer_t, np.float64)
    out = np.zeros((len(q), REC))
    diag = np.zeros(8)
    err = C.create_string_buffer(512)
    gb = np.zeros(3) if gyr_bias is None else np.ascontiguousarray(gyr_bias, np.float64)
    vb = np.zeros(3) if vel_bias is None else np.ascontiguousarray(vel_bias, np.float64)
    rc = lib().ugpmo_preintegrate(
        _p(gt), _p(g), len(gt), _p(vt), _p(v), len(vt), C.c_double(win["gyr_var"]), C.c_double(win["vel_var"]), C.c_double(win["start_t"]),
        _p(q), len(q), int(type), C.c_double(min_freq), C.c_double(state_freq), int(bool(correlate)), int(overlap), _p(gb), _p(vb),
        C.c_double(vel_bias_std), C.c_double(gyr_bias_std), _p(out), _p(diag), err, 512)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    d = dict(nb_state=int(diag[0]), nb_gyr=int(diag[1]), nb_vel=int(diag[2]), iters_rot=int(diag[3]), iters_vel=int(diag[4]), cost_rot=diag[5],
             cost_vel=diag[6], state_freq=diag[7])
    return [unpack(r) for r in out], d


def pack(m):
    """The 83-double record of an unpacked PreintMeas dict (inverse of unpack)."""
    return np.concatenate([np.asarray(m["delta_R"], np.float64).ravel(), np.asarray(m["delta_p"], np.float64), [m["dt"], m["dt_sq_half"]],
                           np.asarray(m["cov"], np.float64).ravel(), np.asarray(m["d_delta_R_d_bw"], np.float64).ravel(), np.asarray(m["d_delta_R_d_t"], np.float64),
                           np.asarray(m["d_delta_p_d_bw"], np.float64).ravel(), np.asarray(m["d_delta_p_d_bv"], np.float64).ravel(), np.asarray(m["d_delta_p_d_t"], np.float64)])


def preintegrate_chunked(win, quantum, infer_t=None, type=UGPM, min_freq=500.0, state_freq=50.0, correlate=True, overlap=8, gyr_bias=None, vel_bias=None,
                         vel_bias_std=0.0, gyr_bias_std=0.0):
    """ugpm::VelPreintegration with opt.quantum = `quantum` >= 0 (chunked mode, preint.h:1584-1702).

    `infer_t`: None (one vector holding win["end_t"]), a flat sequence (one vector) or a list of sequences (the vector-of-vector constructor).
    Returns (list per inner vector of unpacked PreintMeas dicts, diag dict).
    """
    gt, g = np.ascontiguousarray(win["gyr_t"], np.float64), np.ascontiguousarray(win["gyr"], np.float64)
    vt, v = np.ascontiguousarray(win["vel_t"], np.float64), np.ascontiguousarray(win["vel"], np.float64)
    if infer_t is None:
        groups = [[win["end_t"]]]
    elif len(infer_t) > 0 and np.ndim(infer_t[0]) > 0:
        groups = [list(x) for x in infer_t]
    else:
        groups = [list(infer_t)]
    q = np.ascontiguousarray([t for grp in groups for t in grp], np.float64)
    sizes = (C.c_int * len(groups))(*[len(grp) for grp in groups])
    out = np.zeros((max(1, len(q)), REC))
    diag = np.zeros(8)
    err = C.create_string_buffer(512)
    gb = np.zeros(3) if gyr_bias is None else np.ascontiguousarray(gyr_bias, np.float64)
    vb = np.zeros(3) if vel_bias is None else np.ascontiguousarray(vel_bias, np.float64)
    rc = lib().ugpmo_preintegrate_chunked(
        _p(gt), _p(g), len(gt), _p(vt), _p(v), len(vt), C.c_double(win["gyr_var"]), C.c_double(win["vel_var"]), C.c_double(win["start_t"]),
        _p(q), sizes, len(groups), int(type), C.c_double(min_freq), C.c_double(state_freq), int(bool(correlate)), int(overlap), C.c_double(quantum),
        _p(gb), _p(vb), C.c_double(vel_bias_std), C.c_double(gyr_bias_std), _p(out), _p(diag), err, 512)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    d = dict(nb_state=int(diag[0]), nb_gyr=int(diag[1]), nb_vel=int(diag[2]), iters_rot=int(diag[3]), iters_vel=int(diag[4]), cost_rot=diag[5],
             cost_vel=diag[6], state_freq=diag[7])
    res, o = [], 0
    for grp in groups:
        res.append([unpack(out[o + k]) for k in range(len(grp))])
        o += len(grp)
    return res, d


def combine_preints(prev, cur):
    """combinePreints(prev, cur) of math_utils.h:689-726 on two unpacked PreintMeas dicts."""
    a, b = np.ascontiguousarray(pack(prev)), np.ascontiguousarray(pack(cur))
    out = np.zeros(REC)
    lib().ugpmo_combine_preints(_p(a), _p(b), _p(out))
    return unpack(out)


def states(win, state_freq=50.0, correlate=True, overlap=8, gyr_bias=None, vel_bias=None):
    """Optimised GP states [6, S] (mean-subtracted) and hyper-parameters [6, 4] = (l2, sf2, sz2, mean) of one UGPM window."""
    gt, g = np.ascontiguousarray(win["gyr_t"], np.float64), np.ascontiguousarray(win["gyr"], np.float64)
    vt, v = np.ascontiguousarray(win["vel_t"], np.float64), np.ascontiguousarray(win["vel"], np.float64)
    gb = np.zeros(3) if gyr_bias is None else np.ascontiguousarray(gyr_bias, np.float64)
    vb = np.zeros(3) if vel_bias is None else np.ascontiguousarray(vel_bias, np.float64)
    cap = 6 * 4096
    st = np.zeros(cap)
    hy = np.zeros((6, 4))
    err = C.create_string_buffer(512)
    S = lib().ugpmo_states(_p(gt), _p(g), len(gt), _p(vt), _p(v), len(vt), C.c_double(win["gyr_var"]), C.c_double(win["vel_var"]), C.c_double(win["start_t"]),
                           C.c_double(win["end_t"]), C.c_double(state_freq), int(bool(correlate)), int(overlap), _p(gb), _p(vb), _p(st), cap, _p(hy), err, 512)
    if S < 0:
        raise RuntimeError(err.value.decode() or f"ugpmo_states rc={S}")
    return st[: 6 * S].reshape(6, S).copy(), hy


def se_kernel(x1, x2, l2, sf2):
    x1, x2 = np.ascontiguousarray(x1, np.float64), np.ascontiguousarray(x2, np.float64)
    out = np.zeros((len(x1), len(x2)))
    lib().ugpmo_se_kernel(_p(x1), len(x1), _p(x2), len(x2), C.c_double(l2), C.c_double(sf2), _p(out))
    return out


def se_kernel_integral(a, b, x2, l2, sf2):
    b, x2 = np.ascontiguousarray(b, np.float64), np.ascontiguousarray(x2, np.float64)
    out = np.zeros((len(b), len(x2)))
    lib().ugpmo_se_kernel_integral(C.c_double(a), _p(b), len(b), _p(x2), len(x2), C.c_double(l2), C.c_double(sf2), _p(out))
    return out


def kss_int(a, b, l2, sf2):
    return lib().ugpmo_kss_int(a, b, l2, sf2)


def exp_map(v):
    v = np.ascontiguousarray(v, np.float64)
    R = np.zeros((3, 3))
    lib().ugpmo_exp_map(_p(v), _p(R))
    return R


def log_map(R):
    R = np.ascontiguousarray(R, np.float64)
    v = np.zeros(3)
    lib().ugpmo_log_map(_p(R), _p(v))
    return v


def jacobian_res(r, dr):
    r, dr = np.ascontiguousarray(r, np.float64), np.ascontiguousarray(dr, np.float64)
    out = np.zeros((3, 6))
    lib().ugpmo_jacobian_res(_p(r), _p(dr), _p(out))
    return out


def jr(r):
    r = np.ascontiguousarray(r, np.float64)
    out = np.zeros((3, 3))
    lib().ugpmo_jr(_p(r), _p(out))
    return out
