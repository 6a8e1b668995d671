"""Scan-to-submap target assembly (scan_matching_odometry_nodelet.cpp:602-618): keyframe clouds transformed by
their relative poses, concatenated, optionally voxel-grid downsampled, set as the registration target -- device path against the CPU
restatement.  Every output coordinate and label must be BIT-EXACT (the arithmetic is a fixed sequence of double / float operations),
and registering against the assembled target must equal registering against the same cloud set through setInputTarget."""
import importlib

import numpy as np
import pytest

synth = importlib.import_module("go-rio_amd.synth")


def _keyframes(n_frames=5, n=3000):
    frames, odoms = [], []
    for k in range(n_frames):
        odom = synth.gt_transform([0.6 * k, 0.05 * k, 0.01 * k], [0.1 * k, -0.05 * k, 1.2 * k])
        xyz, lab = synth.radar_scan(n + 37 * k, seed=500 + k, sensor_pose=odom)
        frames.append((xyz, lab))
        odoms.append(odom)
    # one relative pose per keyframe but the newest (SMO:605-606 compute theirs as odom_i^-1 * odom_newest in the nodelet's odometry
    # convention; the ABI applies whatever matrix the caller hands over).  Here: sensor-frame points of keyframe i -> newest sensor frame.
    rel = [np.linalg.inv(odoms[-1]) @ odoms[i] for i in range(n_frames - 1)]
    return frames[:-1], rel, frames[-1]


def test_oracle_submap_properties(oracle_apd):
    """CPU: NONE = concatenation of the transformed clouds (non-finite points dropped); VOXELGRID: one point per occupied voxel,
    ascending voxel order, centroid inside its voxel, labels in {0, 1}."""
    frames, rel, _ = _keyframes()
    frames[1][0][5, 0] = np.nan
    x, l = oracle_apd.submap_assemble(frames, rel, 0.0)
    assert x.shape[0] == sum(f[0].shape[0] for f in frames) - 1
    k = frames[0][0].shape[0]
    want = (rel[0][:3, :3] @ frames[0][0][7].astype(np.float64) + rel[0][:3, 3]).astype(np.float32)
    assert np.allclose(x[7], want, atol=1e-5) and l[7] == frames[0][1][7] and np.array_equal(l[:k], frames[0][1])
    leaf = 0.5
    xv, lv = oracle_apd.submap_assemble(frames, rel, leaf)
    assert 0 < xv.shape[0] < x.shape[0] and set(np.unique(lv)) <= {0.0, 1.0}
    inv = np.float32(1.0) / np.float32(leaf)
    vox = np.floor(xv * inv).astype(np.int64)
    assert len({tuple(v) for v in vox}) == xv.shape[0]  # one centroid per voxel
    src_vox = {tuple(v) for v in np.floor(x * inv).astype(np.int64)}
    assert {tuple(v) for v in vox} <= src_vox


@pytest.mark.gpu
@pytest.mark.parametrize("leaf", [0.0, 0.5, 0.1])
def test_submap_assembly_bit_exact(gpu, gorio, oracle_apd, leaf):
    frames, rel, _ = _keyframes()
    frames[2][0][11, 1] = np.inf  # a non-finite point is dropped by PassThrough and VoxelGrid alike
    xo, lo = oracle_apd.submap_assemble(frames, rel, leaf)
    g = gorio.ApdGicp(corr_dist_threshold=2.0)
    n = g.setInputTargetSubmap(frames, rel, voxel_leaf=leaf)
    assert n == xo.shape[0]
    xg, lg = g.getTargetPoints()
    assert np.array_equal(xg, xo) and np.array_equal(lg, lo)


@pytest.mark.gpu
def test_register_against_submap(gpu, gorio, oracle_apd, pose_err):
    """The newest scan against the submap of its predecessors: identical to setInputTarget with the oracle-assembled cloud, and the
    recovered pose is the identity (every keyframe was moved into the newest frame)."""
    frames, rel, newest = _keyframes(n_frames=6, n=4000)
    for leaf in (0.0, 0.2):
        xo, lo = oracle_apd.submap_assemble(frames, rel, leaf)
        kw = dict(corr_dist_threshold=2.0, search=1, transformation_epsilon=0.01)
        a = gorio.ApdGicp(**kw)
        a.setInputTargetSubmap(frames, rel, voxel_leaf=leaf)
        a.setInputSource(*newest)
        ra = a.align()
        b = gorio.ApdGicp(**kw)
        b.setInputTarget(xo, lo)
        b.setInputSource(*newest)
        rb = b.align()
        assert np.array_equal(ra["T"], rb["T"]) and ra["n_linearize"] == rb["n_linearize"]
        te, re = pose_err(np.eye(4), ra["T"])
        assert ra["converged"] and te < 0.05 and re < np.deg2rad(1.0), (leaf, te, re)
    # a shared submap: the assembled target can be handed to other handles like any target
    c = gorio.ApdGicp(**kw)
    c.setInputTargetShared(a)
    c.setInputSource(*newest)
    assert np.array_equal(c.align()["T"], ra["T"])


@pytest.mark.gpu
def test_submap_leaf_too_small_falls_back_like_pcl(gpu, gorio, oracle_apd):
    """A leaf so small that the voxel count overflows int32: PCL warns and returns the input cloud; so do both sides here."""
    frames, rel, _ = _keyframes(n_frames=3, n=500)
    xo, lo = oracle_apd.submap_assemble(frames, rel, 1e-3)
    assert xo.shape[0] == sum(f[0].shape[0] for f in frames)
    g = gorio.ApdGicp()
    assert g.setInputTargetSubmap(frames, rel, voxel_leaf=1e-3) == xo.shape[0]
    xg, lg = g.getTargetPoints()
    assert np.array_equal(xg, xo) and np.array_equal(lg, lo)
