"""GPU tests of ONE RANK'S SHARE of BASELINE.json configs[4] (C5: 512 scans against a shared 1 M-point map on 8 GPUs): a batch of
16 384-point scans registered against ONE 1 000 000-point map that is uploaded, indexed and k-NN'd once on the GPU and referenced by
every handle of the batch (gorio_apd_set_target_shared).

The oracle finishes a 1 M-point map in minutes, not seconds, so at the full size the tests use the size-independent properties the
path offers (the same checks against the CPU oracle run at 100 k in test_configs_gpu.py / test_apd_gpu.py):
  * the exact pruned search returns what the exhaustive kernel returns on the same resident map, bit for bit (indices, squared
    distances), and therefore the same H / b;
  * every scan of the batch -- the points one keyframe contributed to the map, moved rigidly by a known transform -- is registered
    back to 1e-4 m / 1e-4 rad (the reference's registration test accepts 0.05 m / 1 degree, gicp_test.cpp:149-150);
  * a handle that references the shared map gives the bits of a handle that owns a private copy of it;
  * align_batch over the shared map == the same handles aligned one by one.
The multi-rank side of C5 (batch sharding has no collective; the sharded-source mode has one) is covered by test_sharded.py.
"""
import importlib

import numpy as np
import pytest

synth = importlib.import_module("go-rio_amd.synth")
pytestmark = pytest.mark.gpu

MAP_POINTS = 1_000_000
N_SCANS = MAP_POINTS // 16384


TIGHT_LM = dict(corr_dist_threshold=2.0, transformation_epsilon=1e-6, rotation_epsilon=1e-7)  # the optimiser the product ships with


@pytest.fixture(scope="module")
def c5(gpu, gorio):
    tx, tl = synth.local_map(MAP_POINTS, seed=synth.BASE_SEED + 77, n_scans=N_SCANS)
    per = -(-MAP_POINTS // N_SCANS)
    scans, answers = [], []
    for q in range(4):
        # known answer at full size: the points one keyframe contributed to the map, moved rigidly away (the C3 test's construction)
        k = q * 13 % N_SCANS
        pick = np.arange(k * per, k * per + 16384)
        T = synth.gt_transform([0.25 - 0.05 * q, -0.15 + 0.04 * q, 0.04], [0.2, -0.1, 0.8 - 0.3 * q])
        Ti = np.linalg.inv(T)
        sx = (tx[pick].astype(np.float64) @ Ti[:3, :3].T + Ti[:3, 3]).astype(np.float32)
        scans.append((sx, tl[pick].copy()))
        answers.append(T)
    owner = gorio.ApdGicp(search=1, **TIGHT_LM)
    owner.setInputTarget(tx, tl)
    owner.setInputSource(*scans[0])
    owner.calculateCovariances()  # the one index build + k-NN of the map
    return dict(map=(tx, tl), scans=scans, guesses=np.tile(np.eye(4, dtype=np.float32), (4, 1, 1)), answers=answers, owner=owner)


def test_c5_pruned_search_equals_exhaustive_on_the_1m_map(gpu, gorio, c5):
    g = c5["owner"]
    T0 = c5["guesses"][0].astype(np.float64)
    err_p, H_p, b_p = g.linearize(T0)
    corr_p, sqd_p = g.getCorrespondences()
    g.set_params(search=0)  # the same resident map and covariances through nn_search_kernel (16 384 x 1 000 000 distances)
    try:
        err_b, H_b, b_b = g.linearize(T0)
        corr_b, sqd_b = g.getCorrespondences()
    finally:
        g.set_params(search=1)
    assert (corr_p >= 0).sum() > 8000
    assert np.array_equal(corr_p, corr_b) and np.array_equal(sqd_p[corr_p >= 0], sqd_b[corr_b >= 0])
    assert np.array_equal(H_p, H_b) and np.array_equal(b_p, b_b) and err_p == err_b


def test_c5_batch_on_shared_map_recovers_every_pose_and_equals_single_aligns(gpu, gorio, c5, pose_err):
    owner = c5["owner"]
    objs = [owner]
    for sx, sl in c5["scans"][1:]:
        o = gorio.ApdGicp(search=1, **TIGHT_LM)
        o.setInputTargetShared(owner)
        o.setInputSource(sx, sl)
        objs.append(o)
    res = gorio.align_batch(objs, c5["guesses"])
    for r, ans in zip(res, c5["answers"]):
        te, re = pose_err(ans, r["T"])
        assert r["converged"] and te < 1e-4 and re < 1e-4, (te, re)
    for o, g, r in zip(objs, c5["guesses"], res):  # one by one: the same bits
        single = o.align(g)
        assert np.array_equal(single["T"], r["T"]) and np.array_equal(single["H"], r["H"])


def test_c5_shared_map_equals_private_copy(gpu, gorio, c5):
    tx, tl = c5["map"]
    sx, sl = c5["scans"][2]
    shared = gorio.ApdGicp(search=1, **TIGHT_LM)
    shared.setInputTargetShared(c5["owner"])
    shared.setInputSource(sx, sl)
    private = gorio.ApdGicp(search=1, **TIGHT_LM)
    private.setInputTarget(tx, tl)
    private.setInputSource(sx, sl)
    a, b = shared.align(c5["guesses"][2]), private.align(c5["guesses"][2])
    assert np.array_equal(a["T"], b["T"]) and np.array_equal(a["H"], b["H"])
    ca, _ = shared.getCorrespondences()
    cb, _ = private.getCorrespondences()
    assert np.array_equal(ca, cb)


def test_pruned_search_on_a_target_beyond_one_mask_pass(gpu, gorio):
    """The group mask of the three-level search covers 2048 tile groups (4.2 M points) per pass; a 4.5 M-point target needs two passes.
    Exhaustive and pruned search must still agree bit for bit (correspondences, squared distances) -- covariances are injected so that
    only the searches run at this size."""
    rng = np.random.default_rng(7)
    m = 4_500_000
    tx = np.stack([rng.uniform(0, 400, m), rng.uniform(-200, 200, m), rng.uniform(-3, 12, m)], axis=1).astype(np.float32)
    sx = (tx[rng.choice(m, 4096, replace=False)] + rng.normal(0, 0.3, (4096, 3))).astype(np.float32)
    g = gorio.ApdGicp(corr_dist_threshold=2.0, search=1)
    g.setInputTarget(tx, np.zeros(m, np.float32))
    g.setInputSource(sx, np.zeros(4096, np.float32))
    g.setTargetCovariances(np.tile(np.eye(4), (m, 1, 1)))
    g.setSourceCovariances(np.tile(np.eye(4), (4096, 1, 1)))
    _, H_p, b_p = g.linearize(np.eye(4))
    corr_p, sqd_p = g.getCorrespondences()
    g.set_params(search=0)
    _, H_b, b_b = g.linearize(np.eye(4))
    corr_b, sqd_b = g.getCorrespondences()
    assert (corr_b >= 0).sum() > 3000
    assert np.array_equal(corr_p, corr_b) and np.array_equal(sqd_p[corr_p >= 0], sqd_b[corr_b >= 0])
    assert np.array_equal(H_p, H_b) and np.array_equal(b_p, b_b)
