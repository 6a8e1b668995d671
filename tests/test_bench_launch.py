"""bench.py's launch contract: `--gpus N` must really run N ranks (or refuse), a launcher's WORLD_SIZE must agree with the flag,
and nothing runs without a GPU (no CPU fallback).  The N > 1 data path itself (weak-scaling batch shards, barrier, max-over-ranks
time, one JSON line from rank 0) is rehearsed on ONE GPU with two ranks over gloo in the gpu-marked test."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=600):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def _no_gpu():
    import torch

    return torch.cuda.device_count() == 0


@pytest.mark.skipif(not _no_gpu(), reason="needs a machine WITHOUT GPUs (the authoring container)")
def test_gpus_flag_refuses_to_run_fewer_ranks_than_asked():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and "needs 2 visible GPUs" in r.stderr and r.stdout.strip() == ""


@pytest.mark.skipif(not _no_gpu(), reason="needs a machine WITHOUT GPUs (the authoring container)")
def test_no_gpu_no_number():
    r = _run(["--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and r.stdout.strip() == ""  # fails loudly, prints no throughput line


def test_launcher_world_size_must_match_flag():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "must agree" in r.stderr and r.stdout.strip() == ""


@pytest.mark.gpu
def test_two_ranks_rehearsal_on_one_gpu(gpu):
    """`bench.py --gpus 2` with no launcher: the script itself starts two ranks (here both on device 0, reduction tensors over gloo);
    rank 0 prints ONE line with n_gpus = 2 and the work of both ranks."""
    r = _run(["--gpus", "2", "--all-ranks-on-device", "0", "--dist-backend", "gloo", "--pairs", "4", "--steps", "2", "--warmup", "1",
              "--no-cpu-baseline", "--no-exhaustive"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["pairs_per_gpu"] == 4
    assert out["value"] == pytest.approx(2 * 4 * 20 * 2 / (out["ms_per_step"] * 1e-3 * 2), rel=1e-6)  # both ranks' linearisations / max time
    assert out["check"]["ok"]


@pytest.mark.gpu
def test_rccl_coregistration_across_two_gpus(gpu):
    """BASELINE configs[4] / SURVEY 8(e) row 2 on real hardware: two ranks, one GPU each, one source sharded over them behind the C ABI, one
    ncclAllReduce per linearisation.  Skips on a one-GPU box (two RCCL ranks cannot share a device); on a node with >= 2 GPUs it checks
    that RCCL itself reports two ranks, that the collective ran once per linearisation, that both ranks end on bit-identical poses and
    that the pose equals the unsharded align's to 1e-6."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: two RCCL ranks cannot share one device")
    r = _run(["--gpus", "2", "--coreg-only", "--coreg-points", "65536", "--coreg-map-points", "200000", "--iters", "10"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    c = json.loads(lines[0])["coreg"]
    assert c["ok"] and c["ranks_seen_by_rccl"] == 2 and c["pose_identical_on_every_rank"]
    assert c["allreduce_per_linearisation"] == 1.0 and c["allreduce_count"] == 10 * c["aligns_timed"]
    assert c["max_abs_diff_vs_unsharded"] < 1e-6
