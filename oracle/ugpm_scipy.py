"""Second, INDEPENDENT restatement of the UGPM / LPM pre-integration in NumPy / SciPy.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) -- and "parity unpinned" like the C++ oracle: the reference holds no test or
vector for VelInt and cannot be compiled here.  What this file buys is that two restatements written separately, with different
machinery, pin each other:

                      oracle/ugpm_oracle.cpp                          this file
  dense algebra       hand-written LU / Cholesky                      numpy.linalg (LAPACK)
  SO(3) exp / log     restated Eigen AngleAxis / quaternion code      scipy.spatial.transform.Rotation
  the two GP fits     restated Ceres 2.1 trust-region LM              scipy.optimize.least_squares (tight tolerances)
  d(J_r(r) dr)/dr     closed form of the reference's symbolic dump    complex-step differentiation of J_r(r) dr
  time-line merge     emulated SortIndexTracker2                      numpy stable argsort + searchsorted

tests/test_oracle_ugpm.py asserts that both agree AT THE MINIMUM (states, delta_R, delta_p, covariance, Jacobians).
Every function cites the reference lines it follows; paths are relative to /root/reference/4DRadarSLAM/include/VelInt:
PRE = preint.h, MATH = math_utils.h, COST = cost_functions.h, TYPES = types.h.
"""
import numpy as np
from scipy.optimize import least_squares
from scipy.spatial.transform import Rotation as Rot
from scipy.special import erf

K_DT = 0.01        # kNumDtJacobianDelta, MATH:15
K_BW = 1e-4        # kNumGyrBiasJacobianDelta, MATH:17
EXP_TOL = 1e-14    # kExpNormTolerance, MATH:11


# ----------------------------------------------------------------------------------------------------- SO(3)

def skew(v):
    x, y, z = v
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]], dtype=np.result_type(x, y, z))


def exp_so3(v):  # MATH:55-58
    return Rot.from_rotvec(np.asarray(v, float)).as_matrix()


def log_so3(R):  # MATH:48-51 (angle in [0, pi])
    return Rot.from_matrix(R).as_rotvec()


def jr_times(r, v):
    """J_r(r) v for batches r, v of shape (..., 3); written with analytic elementary functions only (no abs, no conj) so that a
    complex perturbation of r differentiates it exactly.  J_r = I - (1 - cos n)/n^2 [r]x + (n - sin n)/n^3 [r]x^2, MATH:63-80."""
    r = np.asarray(r)
    v = np.asarray(v)
    n2 = np.sum(r * r, axis=-1, keepdims=True)
    n = np.sqrt(n2)
    small = np.abs(n) <= EXP_TOL
    n_safe = np.where(small, 1.0, n)
    a = np.where(small, 0.0, (1.0 - np.cos(n_safe)) / (n_safe * n_safe))
    b = np.where(small, 0.0, (n_safe - np.sin(n_safe)) / (n_safe ** 3))
    rv = np.cross(r, v)
    rrv = np.cross(r, rv)
    return v - a * rv + b * rrv


def jr(r):
    return np.stack([jr_times(r, e) for e in np.eye(3)], axis=-1)


def jr_inv(r):  # MATH:83-99 (as written there: + 0.5 [r]x)
    r = np.asarray(r, float)
    n = np.linalg.norm(r)
    out = np.eye(3)
    if n > EXP_TOL:
        S = skew(r)
        out = out + 0.5 * S + ((1.0 / (n * n)) - ((1 + np.cos(n)) / (2.0 * n * np.sin(n)))) * (S @ S)
    return out


def add_n_2pi(r, n):  # MATH:385-397
    nr = np.linalg.norm(r)
    return r if nr == 0 else r / nr * (2.0 * np.pi * n + nr)


# ----------------------------------------------------------------------------------------------------- SE kernel

def se_kernel(x1, x2, l2, sf2):  # MATH:102-110
    d = np.subtract.outer(np.asarray(x1, float), np.asarray(x2, float))
    return sf2 * np.exp(-0.5 * d * d / l2)


def se_kernel_integral(a, b, x2, l2, sf2):  # MATH:114-126: int_a^b k(s, x2) ds
    b = np.atleast_1d(np.asarray(b, float))
    x2 = np.asarray(x2, float)
    s = 1.0 / np.sqrt(2.0 * l2)
    alpha = sf2 * np.sqrt(np.pi * l2 / 2.0)
    return alpha * (erf(np.subtract.outer(b, x2) * s) - erf((a - x2) * s)[None, :])


def se_kernel_integral_dt(a, b, x2, l2, sf2):  # MATH:130-141
    b = np.atleast_1d(np.asarray(b, float))
    x2 = np.asarray(x2, float)
    return sf2 * np.exp(-np.subtract.outer(b, x2) ** 2 / (2.0 * l2)) - sf2 * np.exp(-(x2 - a) ** 2 / (2.0 * l2))[None, :]


def kss_int(a, b, l2, sf2):  # MATH:378-382: int_a^b int_a^b k(s, u) ds du
    d = a - b
    return 2.0 * l2 * sf2 * (np.exp(-d * d / (2.0 * l2)) - 1.0) + np.sqrt(2.0 * np.pi * l2) * sf2 * erf(d / np.sqrt(2.0 * l2)) * d


# ----------------------------------------------------------------------------------------------------- LPM (IterativeIntegrator)

def _interp(data, time, tq):
    """linearInterpolation, MATH:487-532: piecewise linear with a forward-moving segment pointer; before the first sample the
    first segment is extrapolated, after the last one the last segment.  tq ascending."""
    time = np.asarray(time, float)
    if len(time) < 2:
        raise ValueError("InterpolateLinear: this function need at least 2 data points to interpolate")  # MATH:493
    p = np.clip(np.searchsorted(time, tq, side="left") - 1, 0, len(time) - 2)  # time[p] < t <= time[p + 1]
    alpha = (data[p + 1] - data[p]) / (time[p + 1] - time[p])
    beta = data[p] - alpha * time[p]
    return alpha * tq + beta


def _merge(lists):
    """SortIndexTracker2 (TYPES:332-458): merged ascending time line with (list, position) of every stamp."""
    vals = np.concatenate([np.asarray(l, float) for l in lists])
    lid = np.concatenate([np.full(len(l), i) for i, l in enumerate(lists)])
    pos = np.concatenate([np.arange(len(l)) for l in lists])
    o = np.argsort(vals, kind="stable")
    return vals[o], lid[o], pos[o]


def _cumulative(steps, start_index):
    """rotIterativeIntegration (PRE:489-519): R_{i+1} = R_i E_i, then everything re-referenced to the stamp `start_index`.
    For every j the net effect of PRE:508-517 is R_j = P_start^T P_j with P_j = E_0 ... E_{j-1}."""
    P = np.empty((len(steps) + 1, 3, 3))
    P[0] = np.eye(3)
    for i, E in enumerate(steps):
        P[i + 1] = P[i] @ E
    return np.einsum("ji,njk->nik", P[start_index], P)


class Lpm:
    """IterativeIntegrator (PRE:170-742).  gyr / vel: (n, 3); time_lists: the query lists.  Results per list in self.meas[i][j]
    (dicts with the PreintMeas members, TYPES:236-281)."""

    def __init__(self, gyr_t, gyr, vel_t, vel, gyr_var, vel_var, start, time_lists, gyr_bias=(0, 0, 0), vel_bias=(0, 0, 0), min_freq=500.0,
                 bare=False, rot_only=False):
        gyr_t, vel_t = np.asarray(gyr_t, float), np.asarray(vel_t, float)
        gyr = np.asarray(gyr, float) - np.asarray(gyr_bias, float)  # PRE:196-202
        vel = np.asarray(vel, float) - np.asarray(vel_bias, float)  # PRE:203-209
        nq = len(time_lists)
        lists = [np.asarray(l, float) for l in time_lists] + [np.array([start, start + K_DT]), vel_t]  # PRE:214-222
        t, lid, pos = _merge(lists)
        if t[-1] - t[-2] > 1.0 / min_freq:  # PRE:228: getSmallestGap() returns the LAST gap (TYPES:442-450)
            nb = int(np.floor((t[-1] - t[0]) * min_freq))
            lists.append(t[0] + np.arange(nb) * ((t[-1] - t[0]) / nb))  # PRE:230-235
            t, lid, pos = _merge(lists)
        self.start = start
        s_idx = int(np.flatnonzero((lid == nq) & (pos == 0))[0])  # PRE:239
        w = np.stack([_interp(gyr[:, a], gyr_t, t) for a in range(3)], axis=1)  # PRE:336-340
        dt = np.diff(t)
        R = _cumulative(exp_so3(w[:-1] * dt[:, None]), s_idx)
        T = len(t)
        cov = np.zeros((T, 6, 6))
        dRdt = np.zeros((T, 3))
        dRdbw = np.zeros((T, 3, 3))
        if not bare:
            # covariance propagation, PRE:456-466 (only beyond the start stamp), diagonal floored at 1e-6 (PRE:393-405)
            c = np.zeros((3, 3))
            for i in range(T - 1):
                if i + 1 > s_idx:
                    g = w[i] * dt[i]
                    E = exp_so3(g) if np.linalg.norm(g) > 1e-10 else np.eye(3)
                    B = (jr(g) if np.linalg.norm(g) > 1e-10 else np.eye(3)) * dt[i]
                    c = E.T @ c @ E + B @ (gyr_var * np.eye(3)) @ B.T
                cov[i + 1, :3, :3] = c
            d = np.arange(6)
            cov[:, d, d] = np.maximum(cov[:, d, d], 1e-6)
            # numeric Jacobians, PRE:352-379: gyro stamps shifted by -0.01; each gyro axis + 1e-4
            w_s = np.stack([_interp(gyr[:, a], gyr_t - K_DT, t) for a in range(3)], axis=1)
            Rs = _cumulative(exp_so3(w_s[:-1] * dt[:, None]), s_idx)
            dRdt = Rot.from_matrix(np.einsum("nji,njk->nik", R, Rs)).as_rotvec() / K_DT
            for a in range(3):
                wb = w.copy()
                wb[:, a] += K_BW
                Rb = _cumulative(exp_so3(wb[:-1] * dt[:, None]), s_idx)
                dRdbw[:, :, a] = Rot.from_matrix(np.einsum("nji,njk->nik", R, Rb)).as_rotvec() / K_BW
        self.meas = []
        for i in range(nq):
            sel = np.flatnonzero(lid == i)
            sel = sel[np.argsort(pos[sel])]
            self.meas.append([dict(delta_R=R[k], delta_p=np.zeros(3), dt=t[k] - start, dt_sq_half=0.5 * (t[k] - start) ** 2, cov=cov[k].copy(),
                                   d_delta_R_d_bw=dRdbw[k].copy(), d_delta_R_d_t=dRdt[k].copy(), d_delta_p_d_bw=np.zeros((3, 3)),
                                   d_delta_p_d_bv=np.zeros((3, 3)), d_delta_p_d_t=np.zeros(3)) for k in sel])
        if rot_only:
            return
        vsel = np.flatnonzero(lid == nq + 1)
        vsel = vsel[np.argsort(pos[vsel])]
        Rv = R[vsel]
        k_dt = int(np.flatnonzero((lid == nq) & (pos == 1))[0])
        R_dt_start = R[k_dt]  # PRE:268
        velr = np.einsum("nij,nj->ni", Rv, vel)  # reprojectVelData, MATH:415-426
        tq, ql, qp = _merge([np.asarray(l, float) for l in time_lists])
        if bare:
            dp = self._trapezoid(vel_t, velr, tq)
            for k in range(len(tq)):
                if dp[k] is not None:
                    self.meas[ql[k]][qp[k]]["delta_p"] = dp[k]
            return
        # Jacobians of the re-projected velocities, MATH:428-483
        d_bv = Rv  # d_bv[n][a, :] = row a of R
        d_bw = np.einsum("naj,njc->nac", np.cross(vel[:, None, :], Rv), dRdbw[vsel])  # row a: (v x R[a, :]) . dR/dbw
        d_dt = (np.einsum("ji,nj->ni", R_dt_start, velr) - velr) / K_DT
        # posePreintLPM, PRE:524-667: a first pass on the time-shifted data gives the d/dt difference
        dp_shift = self._trapezoid(vel_t - K_DT, velr + K_DT * d_dt, tq)
        dp, jbv, jbw = self._trapezoid(vel_t, velr, tq, d_bv, d_bw)
        for k in range(len(tq)):
            if dp[k] is None:
                continue
            m = self.meas[ql[k]][qp[k]]
            prev = dp_shift[k] if dp_shift[k] is not None else np.zeros(3)
            m["d_delta_p_d_t"] = (prev - dp[k]) / K_DT  # PRE:646
            m["delta_p"] = dp[k]
            for a in range(3):
                m["cov"][3 + a, 3 + a] = (tq[k] - start) * vel_var
            m["d_delta_p_d_bv"], m["d_delta_p_d_bw"] = jbv[k], jbw[k]

    def _trapezoid(self, vt, vd, tq, d_bv=None, d_bw=None):
        """posePreintLPMPartial / posePreintLPM main loop (PRE:552-741): trapezoid integral of the piecewise-linear data from
        `start` to every query >= start.  Reference detail kept: the first partial segment uses the data value at the segment's
        left NODE as d_0 (PRE:571-575), not the interpolated value at `start`."""
        start = self.start
        n = len(vt)
        want_j = d_bv is not None
        out = [None] * len(tq)
        jv = [None] * len(tq)
        jw = [None] * len(tq)
        if not np.any(tq >= start):
            raise ValueError("LPM: the start_time is not in the query domain")  # PRE:680
        p0 = 0
        while vt[p0 + 1] < start:
            p0 += 1
            if p0 == n - 1:
                raise ValueError("LPM: the start_time is not in the data domain")  # PRE:686
        res = np.zeros((len(tq), 3))
        rbv = np.zeros((len(tq), 3, 3))
        rbw = np.zeros((len(tq), 3, 3))
        for a in range(3):
            d = vd[:, a]
            p = p0
            t0, t1, d0, d1 = start, vt[p + 1], d[p], d[p + 1]
            alpha = (d[p + 1] - d[p]) / (vt[p + 1] - vt[p])
            beta = d[p] - alpha * vt[p]
            acc = 0.0
            if want_j:
                ratio = (start - vt[p]) / (vt[p + 1] - vt[p])
                g0w = ratio * d_bw[p + 1, a] + (1 - ratio) * d_bw[p, a]
                g0v = ratio * d_bv[p + 1, a] + (1 - ratio) * d_bv[p, a]
                accv, accw = np.zeros(3), np.zeros(3)
            for k, ti in enumerate(tq):
                if ti < start:
                    continue
                if ti > vt[0]:
                    while not (vt[p] <= ti <= vt[p + 1]) and p < n - 2:
                        acc += (t1 - t0) * (d0 + d1) / 2.0
                        if want_j:
                            accv = accv + (t1 - t0) / 2.0 * (g0v + d_bv[p + 1, a])
                            accw = accw + (t1 - t0) / 2.0 * (g0w + d_bw[p + 1, a])
                        p += 1
                        t0, t1, d0, d1 = vt[p], vt[p + 1], d[p], d[p + 1]
                        alpha = (d1 - d0) / (t1 - t0)
                        beta = d0 - alpha * t0
                        if want_j:
                            g0v, g0w = d_bv[p, a], d_bw[p, a]
                res[k, a] = acc + (ti - t0) * (d0 + (alpha * ti + beta)) / 2.0
                if want_j:
                    ratio = (ti - vt[p]) / (vt[p + 1] - vt[p])
                    g1w = ratio * d_bw[p + 1, a] + (1 - ratio) * d_bw[p, a]
                    g1v = ratio * d_bv[p + 1, a] + (1 - ratio) * d_bv[p, a]
                    rbv[k, a] = accv + (ti - t0) / 2.0 * (g0v + g1v)
                    rbw[k, a] = accw + (ti - t0) / 2.0 * (g0w + g1w)
        for k, ti in enumerate(tq):
            if ti >= start:
                out[k], jv[k], jw[k] = res[k], rbv[k], rbw[k]
        return (out, jv, jw) if want_j else out


# ----------------------------------------------------------------------------------------------------- UGPM (Se3Integrator)

def _slice(t, d, lo, hi):  # GyroVelData::get, TYPES:141-223: lo < t < hi, strictly
    if not lo <= hi:
        raise ValueError("The argument of GyroVelData::Get are not consistent")  # TYPES:160
    m = (t > lo) & (t < hi)
    return t[m], d[m]


def _unwrap_tables(lpm, S, overlap):
    """PRE:1214-1263: rotation vectors of the two query lists relative to the start stamp, continuous across 2 pi."""
    start_R = lpm.meas[2][0]["delta_R"]
    r = np.zeros((2, S, 3))
    for rng in (range(overlap, S), range(overlap - 1, -1, -1)):
        rev = [0, 0]
        prev = [np.zeros(3), np.zeros(3)]
        for i in rng:
            for j in range(2):
                tr = log_so3(start_R.T @ lpm.meas[j][i]["delta_R"])
                cands = [add_n_2pi(tr, rev[j] + q) for q in (-1, 0, 1)]
                best = int(np.argmin([np.linalg.norm(prev[j] - c) for c in cands]))  # getClosest, MATH:399-412 (first minimum)
                prev[j] = cands[best]
                rev[j] += best - 1
                r[j, i] = prev[j]
    return r


class Se3:
    """Se3Integrator (PRE:747-1494) + the bias-prior inflation of VelPreintegration::get (PRE:1734-1757)."""

    def __init__(self, win, duration, state_freq=50.0, overlap=8, correlate=True, gyr_bias=(0, 0, 0), vel_bias=(0, 0, 0), fit_tol=1e-15):
        gt_all, g_all = np.asarray(win["gyr_t"], float), np.asarray(win["gyr"], float)
        vt_all, v_all = np.asarray(win["vel_t"], float), np.asarray(win["vel"], float)
        gyr_var, vel_var, a = float(win["gyr_var"]), float(win["vel_var"]), float(win["start_t"])
        gyr_bias, vel_bias = np.asarray(gyr_bias, float), np.asarray(vel_bias, float)
        vel_freq = (len(vt_all) - 1) / (vt_all[-1] - vt_all[0])  # PRE:766-771
        gyr_freq = (len(gt_all) - 1) / (gt_all[-1] - gt_all[0])
        f = min(max(state_freq, 5.0 / duration), min(vel_freq, gyr_freq))
        S = int(np.ceil(duration * f)) + 2 * overlap  # PRE:775
        x = a - overlap / f + np.arange(S) / f  # PRE:777-786
        gt, g = _slice(gt_all, g_all, x[0], x[-1])  # PRE:789
        vt, v = _slice(vt_all, v_all, x[0], x[-1])
        self.S, self.f, self.x, self.a, self.correlate = S, f, x, a, correlate
        self.nb_gyr, self.nb_vel = len(gt), len(vt)
        lists = [x, x + K_DT, np.array([a])]
        # ---- pass 1 (PRE:1198-1263): LPM seeds of the GP states
        lpm = Lpm(gt, g, vt, v, gyr_var, vel_var, x[0], lists, gyr_bias=gyr_bias, vel_bias=vel_bias)
        start_R = lpm.meas[2][0]["delta_R"]
        r01 = _unwrap_tables(lpm, S, overlap)
        s_dr = (r01[1] - r01[0]) / K_DT  # PRE:1231
        dp0 = np.array([m["delta_p"] for m in lpm.meas[0]])
        dp1 = np.array([m["delta_p"] for m in lpm.meas[1]])
        s_v = ((dp1 - dp0) / K_DT) @ start_R  # rows: start_R^T (dp1 - dp0) / dt, PRE:1232
        d_r_dt_local = jr_times(r01[0], s_dr)  # PRE:1235
        r_temp = r01[0]
        # ---- passes 2-5 (PRE:1265-1399): time-shifted data, then each gyro axis + 1e-4 (bare, rotation only); NO bias prior there
        lpm_s = Lpm(gt - K_DT, g, vt - K_DT, v, gyr_var, vel_var, x[0], lists)
        rs = _unwrap_tables(lpm_s, S, overlap)
        d_r_dt_local_shift = jr_times(rs[0], (rs[1] - rs[0]) / K_DT)  # PRE:1306
        delta_r_time = jr_times(rs[0], rs[0] - r_temp)  # PRE:1307
        d_r_bw_local_shift, delta_r_bw = [], []
        for ax in range(3):
            gb = g.copy()
            gb[:, ax] += K_BW
            lb = Lpm(gt, gb, vt, v, gyr_var, vel_var, x[0], lists, bare=True, rot_only=True)
            rb = _unwrap_tables(lb, S, overlap)
            d_r_bw_local_shift.append(jr_times(rb[0], (rb[1] - rb[0]) / K_DT))  # PRE:1371
            delta_r_bw.append(jr_times(rb[0], rb[0] - r_temp))  # PRE:1372
        # ---- hyper-parameters, PRE:1444-1476
        st = np.concatenate([s_dr, s_v], axis=1)  # (S, 6)
        mean = st.mean(axis=0)
        sf2 = np.maximum(((st - mean) ** 2).mean(axis=0), np.array([gyr_var] * 3 + [vel_var] * 3))
        sz2 = np.array([gyr_var] * 3 + [vel_var] * 3)
        l2 = (3.0 / f) ** 2
        st = st - mean
        self.mean, self.sf2, self.sz2, self.l2 = mean, sf2, sz2, l2
        # ---- Gram matrices, PRE:832-866
        Kinv, KKinv, KintKinv, var = [], [], [], []
        for c in range(6):
            K = se_kernel(x, x, l2, sf2[c])
            Ki = np.linalg.inv(K + sz2[c] * np.eye(S))
            Kinv.append(Ki)
            KKinv.append(K @ Ki)
            if c < 3:
                KintKinv.append(se_kernel_integral(a, x, x, l2, sf2[c]) @ Ki)
            vc = sf2[c] + sz2[c] - np.einsum("ij,ji->i", KKinv[c], K)
            vc = np.where(vc <= 0, sz2[c], vc)
            var.append(vc)
        state_var = np.concatenate(var)
        self.Kinv, self.KintKinv = Kinv, KintKinv
        wgt = [1.0 / np.sqrt(1000.0 * vc) for vc in var]  # GpNormCostFunction weights, COST:31 with the x1000 of PRE:853, 864
        wgt = [np.where(np.isnan(w_), 1.0, w_) for w_ in wgt]  # COST:40
        Jgp = [w_[:, None] * (KKinv[c] - np.eye(S)) for c, w_ in enumerate(wgt)]  # residual = w o ((K K^-1 - I) s), COST:55-57
        # ---- cross tables of the two cost functions, COST:183-190, 293-308
        A_g = [se_kernel(gt, x, l2, sf2[c]) @ Kinv[c] for c in range(3)]  # K_s K^-1
        Ai_g = [se_kernel_integral(a, gt, x, l2, sf2[c]) @ Kinv[c] for c in range(3)]  # K_s_int K^-1
        Ai_v = [se_kernel_integral(a, vt, x, l2, sf2[c]) @ Kinv[c] for c in range(3)]
        A_v = [se_kernel(vt, x, l2, sf2[3 + c]) @ Kinv[3 + c] for c in range(3)]
        g_meas, v_meas = g - gyr_bias, v - vel_bias  # PRE:798-811
        sv = 1.0 / np.sqrt(vel_var)
        ng, nv = len(gt), len(vt)

        def rot_rr(sr):  # r, dr at the gyro stamps, COST:211-224
            r = np.stack([Ai_g[c] @ sr[c] for c in range(3)], axis=1) + np.outer(gt - a, mean[:3])
            dr = np.stack([A_g[c] @ sr[c] for c in range(3)], axis=1) + mean[:3]
            return r, dr

        def rot_res(sflat):
            sr = sflat.reshape(3, S)
            r, dr = rot_rr(sr)
            return np.concatenate([Jgp[c] @ sr[c] for c in range(3)] + [(jr_times(r, dr) - g_meas).ravel()])  # un-weighted, COST:250

        def rot_jac_data(sflat):  # d(J_r(r) dr)/ds: [3 ng, 3 S], rows sample-major (3 i + axis) like COST:243
            sr = sflat.reshape(3, S)
            r, dr = rot_rr(sr)
            h = 1e-30
            J = np.zeros((ng, 3, 3, S))
            Jdr = jr(r)  # d/d(dr)
            for c in range(3):
                rc = r.astype(complex)
                rc[:, c] += 1j * h
                d_r = np.imag(jr_times(rc, dr)) / h  # complex step: d res / d r_c, exact to rounding
                J[:, :, c, :] = d_r[:, :, None] * Ai_g[c][:, None, :] + Jdr[:, :, c][:, :, None] * A_g[c][:, None, :]
            return J.reshape(3 * ng, 3 * S)

        def rot_jac(sflat):
            top = np.zeros((3 * S, 3 * S))
            for c in range(3):
                top[c * S:(c + 1) * S, c * S:(c + 1) * S] = Jgp[c]
            return np.vstack([top, rot_jac_data(sflat)])

        def vel_parts(sr):
            r = np.stack([Ai_v[c] @ sr[c] for c in range(3)], axis=1) + np.outer(vt - a, mean[:3])  # COST:338, 352
            return r, Rot.from_rotvec(-r).as_matrix()  # Exp(-r), COST:353

        def vel_res(svflat, RT):
            sv_ = svflat.reshape(3, S)
            vv = np.stack([A_v[c] @ sv_[c] for c in range(3)], axis=1) + mean[3:]
            return np.concatenate([Jgp[3 + c] @ sv_[c] for c in range(3)] + [((np.einsum("nij,nj->ni", RT, vv) - v_meas) * sv).ravel()])  # COST:381

        def vel_jac_v(RT):  # d res / d s_vel: [3 nv, 3 S]
            J = np.zeros((nv, 3, 3, S))
            for c in range(3):
                J[:, :, c, :] = sv * RT[:, :, c][:, :, None] * A_v[c][:, None, :]  # COST:375
            return J.reshape(3 * nv, 3 * S)

        def vel_jac_r(sr, svflat):  # d res / d s_dr: [3 nv, 3 S], COST:362-370
            r, RT = vel_parts(sr)
            sv_ = svflat.reshape(3, S)
            vv = np.stack([A_v[c] @ sv_[c] for c in range(3)], axis=1) + mean[3:]
            tmp = np.einsum("nij,nj->ni", RT, vv)
            Jr_ = jr(r)
            d_res_d_r = np.einsum("nij,njk->nik", np.stack([skew(t_) for t_ in tmp]), Jr_)
            J = np.zeros((nv, 3, 3, S))
            for c in range(3):
                J[:, :, c, :] = sv * d_res_d_r[:, :, c][:, :, None] * Ai_v[c][:, None, :]
            return J.reshape(3 * nv, 3 * S)

        s_r0 = st[:, :3].T.copy().ravel()
        s_v0 = st[:, 3:].T.copy().ravel()
        # ---- state correlation from the Jacobians at the LPM-initialised state, PRE:886-940, 1478-1492
        if correlate:
            J = np.zeros((3 * ng + 3 * nv, 6 * S))
            J[:3 * ng, :3 * S] = rot_jac_data(s_r0)
            _, RT0 = vel_parts(s_r0.reshape(3, S))
            J[3 * ng:, :3 * S] = vel_jac_r(s_r0.reshape(3, S), s_v0)
            J[3 * ng:, 3 * S:] = vel_jac_v(RT0)
            Cm = np.linalg.inv(J.T @ J + 1e-5 * np.eye(6 * S))
            dsc = np.sqrt(state_var) / np.sqrt(np.diag(Cm))
            self.state_cor = dsc[:, None] * Cm * dsc[None, :]
        self.state_var = state_var
        # ---- the two GP fits (PRE:943-967): Ceres minimises 1/2 sum r^2; any exact least-squares solver finds the same minimiser
        kw = dict(method="trf", x_scale="jac", ftol=fit_tol, xtol=fit_tol, gtol=fit_tol, max_nfev=200)
        sol_r = least_squares(rot_res, s_r0, jac=rot_jac, **kw)
        s_r = sol_r.x.reshape(3, S)
        _, RT = vel_parts(s_r)
        Jv = np.vstack([np.kron(np.eye(3), np.ones((S, S))) * 0.0, vel_jac_v(RT)])
        for c in range(3):
            Jv[c * S:(c + 1) * S, c * S:(c + 1) * S] = Jgp[3 + c]
        sol_v = least_squares(lambda z: vel_res(z, RT), s_v0, jac=lambda z: Jv, **kw)
        s_v_ = sol_v.x.reshape(3, S)
        self.s_r, self.s_v = s_r, s_v_
        self.cost_rot, self.cost_vel = float(sol_r.cost), float(sol_v.cost)
        self.init_r, self.init_v = s_r0.reshape(3, S), s_v0.reshape(3, S)
        # ---- finishStateDiff, PRE:1401-1441
        dt_state = x - a
        state_r = np.stack([KintKinv[c] @ s_r[c] for c in range(3)], axis=1) + np.outer(dt_state, mean[:3])
        d_d_r_dt = np.zeros((3, S))
        d_state_bw = [np.zeros((S, 3)) for _ in range(3)]
        for i in range(S):
            Ji = jr_inv(state_r[i])
            d_r = Ji @ d_r_dt_local[i]
            tr = state_r[i] + Ji @ delta_r_time[i]
            d_d_r_dt[:, i] = (jr_inv(tr) @ d_r_dt_local_shift[i] - d_r) / K_DT
            for ax in range(3):
                trw = state_r[i] + Ji @ delta_r_bw[ax][i]
                dd = (jr_inv(trw) @ d_r_bw_local_shift[ax][i] - d_r) / K_BW
                for c in range(3):
                    d_state_bw[c][i, ax] = dd[c]
        self.d_d_r_dt, self.d_state_bw = d_d_r_dt, d_state_bw
        # ---- inference tables, PRE:978-1060
        self.alpha = [Kinv[c] @ (s_r[c] if c < 3 else s_v_[c - 3]) for c in range(6)]
        d_state_r_bw = [KintKinv[c] @ d_state_bw[c] for c in range(3)]  # PRE:1009-1017 (net effect of the 12-step loop)
        start_r_dt = np.array([(se_kernel_integral(a, [a + K_DT], x, l2, sf2[c]) @ self.alpha[c])[0] + K_DT * mean[c] for c in range(3)])
        R_dt_start = exp_so3(start_r_dt)
        mv = mean[3:]
        self.d_vel_bv = [np.zeros((S, 3)) for _ in range(3)]
        self.d_vel_bw = [np.zeros((S, 3)) for _ in range(3)]
        self.d_vel_dt = np.zeros((3, S))
        for i in range(S):
            Ri = exp_so3(state_r[i])
            vi = s_v_[:, i] + mv
            dbw = np.stack([d_state_r_bw[c][i] for c in range(3)])
            dvbw = -skew(vi) @ jr(-state_r[i]) @ dbw  # PRE:1048
            for c in range(3):
                self.d_vel_bv[c][i] = Ri[c]
                self.d_vel_bw[c][i] = dvbw[c]
            self.d_vel_dt[:, i] = (R_dt_start.T @ vi - vi) / K_DT  # PRE:1054-1058

    def get(self, t, vel_bias_std=0.0, gyr_bias_std=0.0):  # PRE:1069-1153
        S, x, a, l2 = self.S, self.x, self.a, self.l2
        dt = t - a
        r, p = np.zeros(3), np.zeros(3)
        d_r_dt, d_r_dw = np.zeros(3), np.zeros((3, 3))
        d_p_dt, d_p_dw, d_p_dv = np.zeros(3), np.zeros((3, 3)), np.zeros((3, 3))
        ks_all = np.zeros((6, 6 * S))
        var_vec = np.zeros(6)
        for c in range(6):
            ks = se_kernel_integral(a, [t], x, l2, self.sf2[c])[0]
            kk = ks @ self.Kinv[c]
            ks_all[c, c * S:(c + 1) * S] = kk
            var_vec[c] = kss_int(a, t, l2, self.sf2[c]) - kk @ ks
            if var_vec[c] <= 0:
                var_vec[c] = dt * dt * self.sz2[c]
            if c < 3:
                r[c] = ks @ self.alpha[c] + dt * self.mean[c]
                d_r_dw[c] = kk @ self.d_state_bw[c]
                d_r_dt[c] = kk @ self.d_d_r_dt[c]
            else:
                ks_dt = se_kernel_integral_dt(a, [t], x, l2, self.sf2[c])[0]
                p[c - 3] = ks @ self.alpha[c] + dt * self.mean[c]
                d_p_dw[c - 3] = kk @ self.d_vel_bw[c - 3]
                d_p_dv[c - 3] = kk @ self.d_vel_bv[c - 3]
                d_p_dt[c - 3] = ks_dt @ self.alpha[c] + kk @ self.d_vel_dt[c - 3]
        Jr_ = jr(r)
        cov = ks_all @ (self.state_cor if self.correlate else np.diag(self.state_var)) @ ks_all.T
        d = np.sqrt(var_vec) / np.sqrt(np.diag(cov))
        cov = d[:, None] * cov * d[None, :]
        c00 = Jr_ @ cov[:3, :3] @ Jr_.T
        c03 = Jr_ @ cov[:3, 3:]
        cov[:3, :3], cov[:3, 3:], cov[3:, :3] = c00, c03, c03.T
        out = dict(delta_R=exp_so3(r), delta_p=p, dt=dt, dt_sq_half=0.5 * dt * dt, cov=cov, d_delta_R_d_bw=Jr_ @ d_r_dw, d_delta_R_d_t=Jr_ @ d_r_dt,
                   d_delta_p_d_bw=d_p_dw, d_delta_p_d_bv=d_p_dv, d_delta_p_d_t=d_p_dt)
        return inflate(out, vel_bias_std, gyr_bias_std)


def inflate(m, vel_bias_std, gyr_bias_std):  # VelPreintegration::get, PRE:1744-1757
    if vel_bias_std > 0.0 or gyr_bias_std > 0.0:
        J = np.zeros((6, 6))
        J[:3, :3] = np.eye(3)
        J[3:, :3] = m["d_delta_p_d_bw"]
        J[3:, 3:] = m["d_delta_p_d_bv"]
        b = np.diag([gyr_bias_std ** 2] * 3 + [vel_bias_std ** 2] * 3)
        m = dict(m)
        m["cov"] = m["cov"] + J @ b @ J.T
    return m


def preintegrate(win, infer_t=None, type=1, min_freq=500.0, state_freq=50.0, correlate=True, overlap=8, gyr_bias=(0, 0, 0), vel_bias=(0, 0, 0),
                 vel_bias_std=0.0, gyr_bias_std=0.0):
    """Same contract as oracle.ugpm.preintegrate: (list of PreintMeas dicts, diag).  type 1 = UGPM (PRE:1540-1566), 0 = LPM (PRE:1567-1580)."""
    q = np.atleast_1d(np.asarray([win["end_t"]] if infer_t is None else infer_t, float))
    if type == 1:
        se3 = Se3(win, float(q.max() - win["start_t"]), state_freq=state_freq, overlap=overlap, correlate=correlate, gyr_bias=gyr_bias, vel_bias=vel_bias)
        res = [se3.get(float(t), vel_bias_std, gyr_bias_std) for t in q]
        return res, dict(nb_state=se3.S, nb_gyr=se3.nb_gyr, nb_vel=se3.nb_vel, cost_rot=se3.cost_rot, cost_vel=se3.cost_vel, state_freq=se3.f, se3=se3)
    lpm = Lpm(win["gyr_t"], win["gyr"], win["vel_t"], win["vel"], win["gyr_var"], win["vel_var"], win["start_t"], [q], gyr_bias=gyr_bias,
              vel_bias=vel_bias, min_freq=min_freq)
    return [inflate(m, vel_bias_std, gyr_bias_std) for m in lpm.meas[0]], dict()
