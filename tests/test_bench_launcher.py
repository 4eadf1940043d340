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


def test_one_roofline_convention_on_every_leg():
    """4-state and 20-state legs carry the same roofline keys.  The executed
    matrix-instruction counts follow from the tree size alone and are pinned to the SQ_INSTS_MFMA
    figures of the committed counter tables (385 / 628 per tile job of a 27-taxon tree; the
    log-likelihood kernel's 208 + its root reductions); `bound` names the larger of the HBM
    fraction and the executed matrix fraction, `frac` is the contract's figure for it; a leg whose
    kernel instantiation has no committed counters gets no borrowed `pipe_busy`."""
    bench = _bench_module()
    n, P, K = 27, 934, 4
    per_tile = 512.0 * 78
    assert bench.executed_flops_per_tree("gradient_walk_lut_kernel", n, P, K) == 385 * per_tile
    assert bench.executed_flops_per_tree("gradient_walk_lut_fused_kernel", n, P, K) == 385 * per_tile
    assert bench.executed_flops_per_tree("gradient_walk_kernel", n, P, K) == 628 * per_tile
    assert bench.executed_flops_per_tree("loglik_mfma_kernel", n, P, K) == (208 + 5) * 512.0 * 59
    assert bench.executed_flops_per_tree("gradient_hbm_kernel", n, P, K) is None
    assert bench.executed_flops_per_tree("gradient_walk_kernel", n, P, 8) is None
    keys = {"kernel", "kernel_ms", "units_per_launch", "bound", "achieved", "peak", "unit", "frac",
            "frac_algorithmic", "frac_executed", "pipe_busy", "hbm_frac", "traffic", "peak_measured",
            "peak_datasheet"}
    f_ll, f_g = bench.algorithmic_flops(n, P, K)
    b_ll, b_g = bench.algorithmic_bytes(n, P, K)
    r4 = bench.roofline("gradient_walk_lut_kernel", 0.77, 1000, f_g, b_g, 153.7e6, "x", shape=(n, P, K))
    assert keys <= set(r4) and r4["bound"] == "mfma" and r4["frac"] == r4["frac_algorithmic"]
    assert abs(r4["frac_algorithmic"] - 0.511) < 0.002 and abs(r4["frac_executed"] - 0.281) < 0.002
    assert 0.0 < r4["hbm_frac"] < 0.05 and r4["traffic_source"].startswith("static: ")
    assert r4["pipe_busy"] is None or 0.3 < r4["pipe_busy"] < 1.0
    none = bench.roofline("gradient_walk_lut_kernel", 0.26, 1000, 1e6, 1e6, shape=(69, 238, 1), pipe=False)
    assert none["pipe_busy"] is None and none["hbm_frac"] is None and keys - {"traffic"} <= set(none) | {"traffic"}
    # a streamed (20-state) leg: the same keys; HBM binds when its fraction exceeds the executed one
    f_ll20, f_g20 = bench.algorithmic_flops(512, 50000, 4, s=20)
    pre = bench.streamed_roofline("aa_pre_wg_kernel<2>", 42.1, 8, "aa_pre_wg_kernel<2>",
                                  1.0e11, 1.0e10, f_g20 - f_ll20)
    assert keys <= set(pre)
    if pre["traffic"]:
        assert pre["bound"] in ("hbm", "mfma")
        assert pre["frac"] == (pre["hbm_frac"] if pre["bound"] == "hbm" else pre["frac_algorithmic"])
    shard = bench.streamed_roofline("aa_post_wg_kernel<2,false> (+ aa_root_kernel)", 0.5, 1,
                                    "aa_post_wg_kernel<2, false>", 1.0e10, 1.0e9, f_ll20 / 8,
                                    traffic_scale=0.125)
    assert shard["pipe_busy"] is None  # (another instantiation than the profiled one)
    if shard["traffic"]:
        assert "share of the profiled alignment" in shard["traffic_source"]
