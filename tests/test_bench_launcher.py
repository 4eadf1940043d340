"""bench.py --gpus N must start N ranks (the flag used to be parsed and ignored,
so `python bench.py --gpus 8` measured one GPU).  CPU test of the launcher: --dry-run forms a
gloo process group of the started ranks, deals the trees and prints the line -- no GPU work.
Mirrors how the reference's Engine creates `thread_count` executors itself
(/root/reference/src/engine.cpp:23-27)."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py")] + args, env=env,
                          capture_output=True, text=True, timeout=timeout)


def _last_json(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--backend", "gloo", "--dry-run", "--trees", "1001"])
    assert r.returncode == 0, r.stdout + r.stderr
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dry_run"] is True
    assert line["trees_per_rank"] == [501, 500] and line["value"] is None


def test_gpus_1_runs_in_process():
    r = _run(["--gpus", "1", "--dry-run"])
    assert r.returncode == 0, r.stdout + r.stderr
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["trees_per_rank"] == [1000]


def test_launcher_and_flag_must_agree():
    """Under an external launcher (WORLD_SIZE set) a different --gpus is an error, not a
    silently mislabelled run."""
    r = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "must agree" in (r.stdout + r.stderr)


def test_driver_style_launch_is_accepted():
    """The driver's own form: torch.distributed.run around bench.py --gpus N."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                        "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                        "29731", os.path.join(REPO, "bench.py"), "--gpus", "2", "--backend",
                        "gloo", "--dry-run"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert _last_json(r.stdout)["rccl_ranks"] == 2
