"""Seeded synthetic inputs for the hot path: 4D-radar-like scans and gyro / ego-velocity windows.

The reference ships no radar data (only two LiDAR PCDs under ndt_omp/data) and no IMU recordings, so tests and
bench.py use these generators.  Scene and noise follow the sensor model APD-GICP itself assumes
(fast_apdgicp_impl.hpp:194-197: sigma_r = dist_var * r / 400, sigma_az = r sin(az_var deg), sigma_el = r sin(el_var deg)),
the field of view comes from launch/ntu_loop3.launch (sc_azimuth_range 56.5 deg), and cluster labels are written to
`normal_x` the way preprocessing_nodelet_ntu.cpp:561-568 does (small integers stored as float, 0 = unclustered).
The IMU window follows radar_graph_slam_nodelet.cpp:465-512 (gyro var 1.74532925e-3, ego-velocity var 1e-6,
samples from start-0.5 s to end+1.0 s).
"""
import numpy as np

BASE_SEED = 20250704

# ground truth motion between source and target scans: t = (0.40, -0.10, 0.02) m, rpy = (0.3, -0.2, 2.0) deg
GT_TRANSLATION = np.array([0.40, -0.10, 0.02])
GT_RPY_DEG = np.array([0.3, -0.2, 2.0])


def rpy_to_matrix(rpy_deg):
    r, p, y = np.deg2rad(rpy_deg)
    cx, sx, cy, sy, cz, sz = np.cos(r), np.sin(r), np.cos(p), np.sin(p), np.cos(y), np.sin(y)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def gt_transform(translation=GT_TRANSLATION, rpy_deg=GT_RPY_DEG):
    T = np.eye(4)
    T[:3, :3] = rpy_to_matrix(rpy_deg)
    T[:3, 3] = translation
    return T


def _scene(rng):
    """Ground plane z = -1.5 m plus 12 axis-aligned boxes / walls inside x in [1, 120] m."""
    boxes = []
    for i in range(12):
        cx = rng.uniform(8.0, 110.0)
        cy = rng.uniform(-45.0, 45.0)
        if i % 3 == 0:  # long wall
            sx, sy, sz = rng.uniform(10, 30), rng.uniform(0.3, 0.8), rng.uniform(2.5, 6.0)
        elif i % 3 == 1:  # building block
            sx, sy, sz = rng.uniform(4, 12), rng.uniform(4, 12), rng.uniform(3, 10)
        else:  # vehicle sized
            sx, sy, sz = rng.uniform(1.5, 5), rng.uniform(1.5, 2.5), rng.uniform(1.4, 3.0)
        lo = np.array([cx - sx / 2, cy - sy / 2, -1.5])
        hi = np.array([cx + sx / 2, cy + sy / 2, -1.5 + sz])
        boxes.append((lo, hi))
    return boxes


def _raycast(dirs, origin, boxes, r_min=1.0, r_max=120.0):
    """Nearest hit range and object id (0 = ground, 1.. = boxes) for rays origin + r * dir; inf when nothing is hit."""
    n = dirs.shape[0]
    best = np.full(n, np.inf)
    obj = np.zeros(n, np.int32)
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = (-1.5 - origin[2]) / dirs[:, 2]
    ok = (dirs[:, 2] < 0) & (tg >= r_min) & (tg <= r_max)
    best[ok] = tg[ok]
    for bi, (lo, hi) in enumerate(boxes):
        with np.errstate(divide="ignore", invalid="ignore"):
            t1 = (lo[None, :] - origin[None, :]) / dirs
            t2 = (hi[None, :] - origin[None, :]) / dirs
        tn = np.nanmax(np.minimum(t1, t2), axis=1)
        tf = np.nanmin(np.maximum(t1, t2), axis=1)
        hit = (tn <= tf) & (tn >= r_min) & (tn <= r_max) & (tn < best)
        best[hit] = tn[hit]
        obj[hit] = bi + 1
    return best, obj


def radar_scan(n_points, seed, sensor_pose=None, scene_seed=BASE_SEED, dist_var=0.86, az_var=0.5, el_var=1.0, noise=True):
    """One synthetic radar scan in the SENSOR frame.

    Returns (xyz float32 [n,3], label float32 [n]).  `sensor_pose` (4x4, world <- sensor) places the sensor in the fixed
    scene; `scene_seed` fixes the scene, `seed` the sampling / noise stream.
    """
    rng = np.random.default_rng(seed)
    boxes = _scene(np.random.default_rng(scene_seed))
    pose = np.eye(4) if sensor_pose is None else np.asarray(sensor_pose, float)
    R, origin = pose[:3, :3], pose[:3, 3]
    pts, labels = [], []
    need = n_points
    while need > 0:
        nb = max(4096, int(need * 1.6))
        az = np.deg2rad(rng.uniform(-56.5, 56.5, nb))
        el = np.deg2rad(rng.uniform(-22.5, 22.5, nb))
        d_s = np.stack([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)], axis=1)
        d_w = d_s @ R.T
        r, obj = _raycast(d_w, origin, boxes)
        ok = np.isfinite(r)
        r, obj, az, el = r[ok], obj[ok], az[ok], el[ok]
        if noise:
            r = r + rng.normal(0.0, 1.0, r.shape) * (dist_var * r / 400.0)
            az = az + rng.normal(0.0, 1.0, r.shape) * np.sin(np.deg2rad(az_var))
            el = el + rng.normal(0.0, 1.0, r.shape) * np.sin(np.deg2rad(el_var))
        p = np.stack([r * np.cos(el) * np.cos(az), r * np.cos(el) * np.sin(az), r * np.sin(el)], axis=1)
        pts.append(p[:need])
        labels.append(obj[:need])
        need -= min(need, p.shape[0])
    xyz = np.concatenate(pts).astype(np.float32)
    lab = (np.concatenate(labels) % 16).astype(np.float32)
    return xyz, lab


def scan_pair(n_src, n_tgt, seed, T_gt=None, noise_scale=1.0):
    """Source / target scans of the same scene; aligning source onto target recovers `T_gt` (guess = identity).

    The target is an independent sample (different rng stream) of the scene seen from the source sensor pose, then
    moved by T_gt, as SURVEY 8d prescribes.  `noise_scale` scales the three sensor sigmas (1.0 = the APD model).  Returns (src_xyz, src_label, tgt_xyz, tgt_label, T_gt).
    """
    T = gt_transform() if T_gt is None else np.asarray(T_gt, float)
    kw = dict(dist_var=0.86 * noise_scale, az_var=0.5 * noise_scale, el_var=1.0 * noise_scale)
    src_xyz, src_lab = radar_scan(n_src, seed=seed * 2 + 1, **kw)
    t_xyz, tgt_lab = radar_scan(n_tgt, seed=seed * 2 + 2, **kw)
    tgt_xyz = (t_xyz.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    return src_xyz, src_lab, tgt_xyz, tgt_lab, T


def local_map(n_points, seed, n_scans=6, T_gt=None):
    """A local map of `n_points` built from `n_scans` accumulated scans (C3: 100k-pt map), moved by T_gt."""
    T = gt_transform() if T_gt is None else np.asarray(T_gt, float)
    rng = np.random.default_rng(seed)
    per = -(-n_points // n_scans)
    pts, labs = [], []
    for s in range(n_scans):
        pose = np.eye(4)
        pose[:3, 3] = [0.8 * s, rng.uniform(-0.2, 0.2), 0.0]
        pose[:3, :3] = rpy_to_matrix([0, 0, rng.uniform(-1.0, 1.0)])
        xyz, lab = radar_scan(per, seed=seed * 16 + s + 3, sensor_pose=pose)
        world = xyz.astype(np.float64) @ pose[:3, :3].T + pose[:3, 3]
        pts.append(world)
        labs.append(lab)
    world = np.concatenate(pts)[:n_points]
    lab = np.concatenate(labs)[:n_points]
    xyz = (world @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    return xyz, lab.astype(np.float32)


# --------------------------------------------------------------------------- IMU / ego-velocity windows

GYR_VAR = 1.74532925e-03  # radar_graph_slam_nodelet.cpp:476
VEL_VAR = 1e-6  # radar_graph_slam_nodelet.cpp:493


def omega_true(t):
    return np.stack([0.10 * np.sin(2 * np.pi * 0.7 * t), 0.05 * np.cos(2 * np.pi * 0.4 * t), 0.30 + 0.10 * np.sin(2 * np.pi * 0.2 * t)], axis=-1)


def vel_true(t):
    return np.stack([5.0 + np.sin(t), 0.2 * np.sin(2 * t), 0.05 * np.ones_like(t)], axis=-1)


def imu_window(seed, start_t=10.0, duration=1.0, gyr_hz=200.0, vel_hz=200.0, noise=True, omega_fn=omega_true, vel_fn=vel_true,
               gyr_var=GYR_VAR, vel_var=VEL_VAR):
    """One pre-integration window as the back end assembles it (RGS:465-512).

    Returns dict(gyr_t, gyr [n,3], vel_t, vel [n,3], gyr_var, vel_var, start_t, end_t); samples span
    [start-0.5, end+1.0] like the nodelet's queues (RGS:466, 482).
    """
    rng = np.random.default_rng(seed)
    end_t = start_t + duration
    t0, t1 = start_t - 0.5, end_t + 1.0
    gyr_t = t0 + np.arange(int(round((t1 - t0) * gyr_hz)) + 1) / gyr_hz
    vel_t = t0 + np.arange(int(round((t1 - t0) * vel_hz)) + 1) / vel_hz
    gyr = omega_fn(gyr_t)
    vel = vel_fn(vel_t)
    if noise:
        gyr = gyr + rng.normal(0.0, np.sqrt(gyr_var), gyr.shape)
        vel = vel + rng.normal(0.0, np.sqrt(vel_var), vel.shape)
    return dict(gyr_t=gyr_t, gyr=np.ascontiguousarray(gyr), vel_t=vel_t, vel=np.ascontiguousarray(vel), gyr_var=gyr_var, vel_var=vel_var,
                start_t=start_t, end_t=end_t)
