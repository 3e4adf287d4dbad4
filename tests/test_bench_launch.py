"""`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run and reaches the JSON line
(it used to raise SystemExit, so a driver that starts the N = 8 line like the N = 1 line got an error
instead of a number).  CPU: the rendezvous runs on gloo and no model is built (--launch-selftest); the GPU path behind it is
the one bench.py always had (RANK / LOCAL_RANK / WORLD_SIZE from the environment)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_gpus2_self_spawns_and_prints_one_json_line():
    r = run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--launch-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["launch_selftest"] is True
    assert d["max_over_ranks_s"] >= 0.02          # rank 1 sleeps 20 ms: the maximum over ranks, not rank 0's own time


def test_world_size_mismatch_is_an_error():
    r = run(["--gpus", "2", "--launch-selftest"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_single_rank_selftest():
    r = run(["--gpus", "1", "--launch-selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["n_gpus"] == 1


def test_a_non_finite_loss_aborts_the_bench():
    """Round 6: a NaN model runs 11 % faster on the MI355X -- bench.py must refuse to print a throughput for one."""
    import math

    import pytest

    import bench
    assert bench.require_finite_loss(0.75) == 0.75
    for bad in (float("nan"), float("inf"), -float("inf")):
        with pytest.raises(SystemExit) as e:
            bench.require_finite_loss(bad)
        assert "no throughput is reported" in str(e.value) and not math.isfinite(bad)
