"""Sharded-source co-registration: ONE large source cloud against ONE map, source points split over the ranks of a
torch.distributed group, the map replicated (second row).

Every rank holds a registration object for its own shard (an `ApdGicp` on its GPU) and evaluates the partial H, b and error there;
the only exchange is a 43-double all-reduce per linearisation and a 1-double all-reduce per LM trial (RCCL over xGMI with the "nccl"
backend, gloo on CPU).  Every rank then performs the identical 6x6 solve and SO(3) update in the same arithmetic, so the poses stay
bit-identical on all ranks without a broadcast.  The optimiser shell below is the host-side mirror of
fast_gicp/gicp/impl/lsq_registration_impl.hpp:55-173 (file:line cited per step); it contains no per-point numerics.

`reg` must provide linearize(T) -> (error, H[6,6], b[6]) and compute_error(T) -> error for ITS shard, with the cluster weight
normalised by the GLOBAL source size (ApdGicp: set_params(cl_weight_points=N_total)).
"""
import numpy as np


def _all_reduce_sum(vec, group, device=None):
    import torch
    import torch.distributed as dist

    if group is None and not (dist.is_available() and dist.is_initialized()):
        return vec
    t = torch.from_numpy(np.ascontiguousarray(vec, np.float64))
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def so3_exp_matrix(omega):
    """so3_exp (so3.hpp:59-78) -> Quaterniond -> rotation matrix."""
    theta_sq = float(omega @ omega)
    if theta_sq < 1e-10:
        theta_quad = theta_sq * theta_sq
        imag = 0.5 - 1.0 / 48.0 * theta_sq + 1.0 / 3840.0 * theta_quad
        real = 1.0 - 1.0 / 8.0 * theta_sq + 1.0 / 384.0 * theta_quad
    else:
        theta = np.sqrt(theta_sq)
        imag = np.sin(0.5 * theta) / theta
        real = np.cos(0.5 * theta)
    w, x, y, z = real, imag * omega[0], imag * omega[1], imag * omega[2]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz, txx, txy, txz, tyy, tyz, tzz = tx * w, ty * w, tz * w, tx * x, ty * x, tz * x, ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy], [txy + twz, 1 - (txx + tzz), tyz - twx], [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def ldlt6_solve(A, rhs):
    """Eigen::LDLT<Matrix6d>(A).solve(rhs) as LsqRegistration uses it (lsq_registration_impl.hpp:112, 137): L D L^T with symmetric
    diagonal pivoting, zero pivots skipped (a singular H -- e.g. no correspondence at all -- gives a zero step, not an exception).
    The same operation order as ldlt6_solve in go-rio_amd/csrc/apd_kernels.hip, so a sharded run and a single-handle run take
    identical steps from identical H, b."""
    A = np.array(A, dtype=np.float64)
    n = A.shape[0]
    perm = list(range(n))
    for k in range(n):
        piv = k + int(np.argmax(np.abs(np.diag(A)[k:])))
        if abs(A[piv, piv]) <= abs(A[k, k]):
            piv = k  # first maximum, like the strict '>' scan of the kernel
        if piv != k:
            A[[k, piv], :] = A[[piv, k], :]
            A[:, [k, piv]] = A[:, [piv, k]]
            perm[k], perm[piv] = perm[piv], perm[k]
        d = A[k, k]
        if d == 0.0:
            continue
        col = A[:, k].copy()
        for i in range(k + 1, n):
            l = col[i] / d
            for j in range(k + 1, i + 1):
                A[i, j] -= l * col[j]
            A[i, k] = l
        for i in range(k + 1, n):
            for j in range(i + 1, n):
                A[i, j] = A[j, i]
    y = np.array([rhs[perm[i]] for i in range(n)], dtype=np.float64)
    for i in range(n):
        for j in range(i):
            y[i] -= A[i, j] * y[j]
    for i in range(n):
        y[i] = y[i] / A[i, i] if A[i, i] != 0.0 else 0.0
    for i in range(n - 1, -1, -1):
        for j in range(i + 1, n):
            y[i] -= A[j, i] * y[j]
    x = np.zeros(n)
    for i in range(n):
        x[perm[i]] = y[i]
    return x


def _delta(d):
    D = np.eye(4)
    D[:3, :3] = so3_exp_matrix(d[:3])  # rotation block first (lsq_registration_impl.hpp:117-119, 140-142)
    D[:3, 3] = d[3:]
    return D


def _is_converged(delta, rot_eps, trans_eps):  # lsq_registration_impl.hpp:83-92
    r = np.abs(delta[:3, :3] - np.eye(3)).max() / rot_eps
    t = np.abs(delta[:3, 3]).max() / trans_eps
    return max(r, t) < 1


def align_sharded(reg, guess=None, max_iterations=64, rotation_epsilon=2e-3, transformation_epsilon=5e-4, optimizer="LM", lm_max_iterations=10,
                  lm_init_lambda_factor=1e-9, group=None, device=None):
    """LsqRegistration::computeTransformation over a sharded source.  Returns dict(T float32 4x4, H, converged, nr_iterations, n_linearize)."""
    x0 = np.eye(4) if guess is None else np.asarray(guess, np.float32).astype(np.float64)
    lam, converged, nr_iterations, n_lin = -1.0, False, 0, 0
    Hfin = np.eye(6)
    for it in range(max_iterations):
        nr_iterations = it
        err, H, b = reg.linearize(x0)  # partial sums of this rank's shard
        red = _all_reduce_sum(np.concatenate([H.ravel(), b, [err]]), group, device)  # THE collective: 43 doubles
        H, b, y0 = red[:36].reshape(6, 6), red[36:42], float(red[42])
        n_lin += 1
        ok = False
        if optimizer == "GN":  # lsq_registration_impl.hpp:107-123
            delta = _delta(ldlt6_solve(H, -b))
            x0 = delta @ x0
            Hfin, ok = H, True
        else:  # lsq_registration_impl.hpp:127-173
            if lam < 0.0:
                lam = lm_init_lambda_factor * np.abs(np.diag(H)).max()
            nu = 2.0
            for _ in range(lm_max_iterations):
                d = ldlt6_solve(H + lam * np.eye(6), -b)
                delta = _delta(d)
                xi = delta @ x0
                yi = float(_all_reduce_sum(np.array([reg.compute_error(xi)]), group, device)[0])
                rho = (y0 - yi) / float(d @ (lam * d - b))
                if rho < 0:
                    if _is_converged(delta, rotation_epsilon, transformation_epsilon):
                        ok = True
                        break
                    lam, nu = nu * lam, 2 * nu
                    continue
                x0 = xi
                lam = lam * max(1.0 / 3.0, 1 - (2 * rho - 1) ** 3)
                Hfin, ok = H, True
                break
        if not ok:
            break  # "lm not converged!!" (lsq_registration_impl.hpp:71-74)
        converged = _is_converged(delta, rotation_epsilon, transformation_epsilon)
        if converged:
            break
    return dict(T=x0.astype(np.float32), H=Hfin, converged=bool(converged), nr_iterations=nr_iterations, n_linearize=n_lin)
