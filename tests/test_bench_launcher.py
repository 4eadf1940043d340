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


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def _last_json(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks_and_prints_one_line():
    r = _run(["--gpus", "2", "--dry-run", "--trees", "1001"])  # (--dry-run defaults to gloo)
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
                        str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2",
                        "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert _last_json(r.stdout)["rccl_ranks"] == 2


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_line_is_small():
    """the round-3 line had grown to 21 KB and the driver (8 KB tail of stdout)
    could not parse it.  The last line is built by final_line() from the full result; fed the
    full round-3 result (profiles/r03_bench_line.json, every `also` leg with its two rooflines
    and notes) it must stay under 4 KB, parse, and still carry the contract's fields."""
    bench = _bench_module()
    full = json.load(open(os.path.join(REPO, "profiles", "r03_bench_line.json")))
    assert len(json.dumps(full)) > 16000  # the canned input really is the oversized one
    full["host_pointer_trees_per_s"] = 851234.56789
    full["adapter_trees_per_s"] = 801234.56789
    full["reduced"] = {"trees_per_s": 1.1e6, "ms_per_step": 0.91, "all_reduce_doubles": 4098,
                       "collective": "one all_reduce over 8 rank(s)", "max_rel_err": 1e-15,
                       "note": "x" * 500}
    full["weak"] = {"value": 8e6, "unit": "trees/s", "ms_per_step": 1.0, "trees_per_gpu": 1000}
    line = bench.final_line(full)
    assert len(line) < 4096, len(line)
    d = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "parity_checked", "parity_max_rel_err",
                "small_batch_ms", "also"):
        assert key in d, key
    assert d["value"] == pytest.approx(full["value"], rel=1e-5)
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
    # the other legs: one table, the same columns for every leg (round 6)
    assert d["config"]["workload"] and len(d["also"]["rows"]) == len(full["also"])
    assert {"workload", "trees_per_s", "bound", "frac", "frac_algorithmic", "frac_executed", "pipe_busy",
            "hbm_frac"} <= set(d["also"]["keys"])
    assert all(len(r) == len(d["also"]["keys"]) for r in d["also"]["rows"])
    # and a line that cannot be made small is refused, not printed
    full["config"]["workload"] = "w" * 5000
    with pytest.raises(AssertionError, match="bytes"):
        bench.final_line(full)
