"""CPU: bench.py's N-rank contract (step contract of train.py:176-188 under data parallelism).
`python bench.py --gpus N` without a launcher must really start N ranks and prove the collective backend connected
N of them (n_ranks_rccl); fewer GPUs than asked, or a launcher/flag mismatch, must fail loudly."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def _line(r):
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout[-400:], r.stderr[-800:])    # printed exactly once, by exactly one of watchdog / main path
    return json.loads(lines[0])


def test_self_launch_two_ranks_dry_run():
    r = _run(["--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-800:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                        # rank 0's line only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_rccl"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["dry_run"] is True
    assert d["scaling"] == "weak" and d["value"] > 0


def test_self_launch_eight_ranks_dry_run():
    """configs[3]'s world size: `bench.py --gpus 8` starts EIGHT ranks, the backend connects eight of them, the (stub) gradient
    exchange sums over eight, and the train leg reports the global batch 512 = 64 x 8."""
    r = _run(["--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1"], timeout=600)
    assert r.returncode == 0, r.stderr[-800:]
    d = _line(r)
    assert d["n_gpus"] == 8 and d["n_ranks_rccl"] == 8 and d["scaling"] == "weak"
    assert d["train_step"] == {"parallelism": "native communicator", "grad_sum": 36.0, "batch_per_gpu": 64, "global_batch": 512}


@pytest.mark.skipif(torch.cuda.device_count() >= 2, reason="needs a box with fewer than 2 GPUs")
def test_more_ranks_than_gpus_fails_loudly():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-train", "--no-cpu-baseline"], timeout=120)
    assert r.returncode != 0
    assert "--gpus 2 requested" in r.stderr and not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_launcher_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_train_leg_control_flow_two_ranks():
    """The data-parallel train leg's control flow (bench.guarded_dp_leg / agree_on_native_comm) under gloo with a stubbed
    trainer: all good; one rank without a native communicator -> every rank falls back together; one rank stalls -> the
    watchdog prints the headline line with the reason and the run exits NON-zero; the leg raises on one rank -> non-zero."""
    r = _run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-800:]
    assert _line(r)["train_step"] == {"parallelism": "native communicator", "grad_sum": 3.0, "batch_per_gpu": 64, "global_batch": 128}

    r = _run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"], {"UBD_BENCH_DRY_FAULT": "attach_fail"})
    assert r.returncode == 0, r.stderr[-800:]
    assert _line(r)["train_step"] == {"parallelism": "torch.distributed fallback", "grad_sum": 3.0, "batch_per_gpu": 64, "global_batch": 128}
    assert "native RCCL communicator unavailable" in r.stderr

    r = _run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"], {"UBD_BENCH_DRY_FAULT": "stall", "UBD_BENCH_TRAIN_TIMEOUT_S": "4"})
    assert r.returncode != 0
    d = _line(r)
    assert "did not finish within 4 s" in d["train_step"]["error"] and d["value"] > 0       # the headline figures survived

    r = _run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"], {"UBD_BENCH_DRY_FAULT": "leg_fail", "UBD_BENCH_TRAIN_TIMEOUT_S": "6"})
    assert r.returncode != 0
    assert "train_step" in _line(r)
