#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X phylogenetic engine.

Metric (BASELINE.json): tree log-likelihoods+gradients / second, batched.
One "step" = one Engine::Gradients call (libsbn `phylo_gradients` semantics: per-tree
log-likelihood + branch-length gradient + site-model gradient) over a batch of trees whose
inputs are already resident in HBM, followed -- when more than one GPU takes part -- by the
single RCCL collective that returns every rank's per-tree results (all_gather).

Workload of the headline line (north_star / BASELINE.json configs[1]): the reference's DS1
alignment (27 taxa, 1949 sites -> 934 site patterns) x 1000 trees (the 100 topologies of
DS1.100_topologies.nwk x 10 synthetic branch-length draws, Exp(mean 0.1)), JC69 + the
reference's 4-category discrete rate model ("weibull+4", shape 1.0).  STRONG scaling: the
1000 trees are dealt to the ranks in contiguous blocks (1000 / N per GPU), as the
reference's FatBeagleParallelize deals a tree collection to its FatBeagles; a weak-scaled
figure (1000 trees per GPU) rides along in "weak".

Output.  The LAST line of stdout is ONE compact JSON object (< 4 KB, asserted: final_line) with
the contract's fields, the dominant kernel's `roofline`, `cpu_baseline`, the parity of the timed
outputs, `small_batch_ms` (the 125-tree step one GPU of eight runs, from a hipGraph), the
host-pointer / C++ adapter / fused-reduction legs, and one short entry per other configuration
in `also` (GTR+4G full and branch-only, log-likelihoods, the 1949-pattern shape, a 36-taxon arena shape, fluA rooted,
the 20-state 512 x 50 000 case).  Everything behind those numbers -- second rooflines, phase
tables, notes, samples -- is written to bench_also.json and printed before the last line as
`BENCH_ALSO <name> <json>` lines.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--trees T_total] [--mode gradient|loglik]

--gpus N > 1: one rank per GPU over RCCL.  Under a launcher (torch.distributed.run: WORLD_SIZE
set) N must equal WORLD_SIZE; without one, bench.py starts the N ranks itself as a child
`python -m torch.distributed.run --nproc-per-node N bench.py ...` (before anything touches the
GPU) and relays rank 0's JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
DATA = os.path.join(REPO, "tests", "golden", "data")  # the reference's own data files (fixtures)

import numpy as np  # noqa: E402

# Measured device-wide with tools/fp64_peak_probe.hip (profiles/r02_fp64_peak_probe.txt):
# v_mfma_f64_4x4x4_4b sustains 71.1 TFLOP/s at 8 waves per SIMD (64.4 at one), the 16+4 row mix
# of the 20-state kernels 70.7, v_mfma_f64_16x16x4 alone 49.1, v_fma_f64 57.5.  (Data sheet:
# 78.6.)  The roofline fractions below are against the MEASURED matrix-core figure.
FP64_PEAK_TFLOPS = 71.1
FP64_PEAK_SOURCE = ("measured: tools/fp64_peak_probe.hip, v_mfma_f64_4x4x4_4b at 8 waves/SIMD, "
                    "profiles/r02_fp64_peak_probe.txt (data sheet 78.6)")
HBM_PEAK_GBPS = 8000.0


def algorithmic_bytes(n, P, K, s=4):
    """PLV-streaming model, compact tips."""
    plv = K * P * s * 8
    return 2 * (n - 1) * plv + 4 * n * P, (10 * n - 14) * plv + 12 * n * P


def algorithmic_flops(n, P, K, s=4):
    """F_LL = (n-1) K P 4s^2, F_G = F_LL + (2n-2) K P (4s^2 + 2s^2 + 4s)."""
    f_ll = (n - 1) * K * P * 4 * s * s
    return f_ll, f_ll + (2 * n - 2) * K * P * (4 * s * s + 2 * s * s + 4 * s)


# ---------------------------------------------------------------------------------------
# Workloads, built through the product's own ingest (libsbn_amd/csrc/host: FASTA, Newick,
# site patterns, time trees)
# ---------------------------------------------------------------------------------------
def ds1_workload(T, seed=43):
    from libsbn_amd import _hostapi
    tc = _hostapi.TreeCollection.of_newick_file(os.path.join(DATA, "DS1.100_topologies.nwk"))
    tips, w, _ = tc.site_pattern(os.path.join(DATA, "DS1.fasta"))
    pids100 = np.stack(tc.parent_ids)
    reps = (T + len(pids100) - 1) // len(pids100)
    pids = np.tile(pids100, (reps, 1))[:T]
    rng = np.random.default_rng(seed)
    bls = rng.exponential(0.1, size=(T, pids.shape[1] + 1))
    bls[:, -1] = 0.0
    return tips, w, np.ascontiguousarray(pids), bls


def evolved_alignment(pid, bl, P, rng, states=4):
    """i.i.d. uniform root states evolved down one tree under the Jukes-Cantor-like model of
    `states` states (SURVEY 8d, S-DS1 with 1949 columns: every column its own pattern)."""
    nodes = len(pid) + 1
    n = (nodes + 2) // 2
    x = np.zeros((nodes, P), dtype=np.int32)
    x[nodes - 1] = rng.integers(0, states, P)
    for v in range(nodes - 2, -1, -1):
        p_same = 1.0 / states + (1 - 1.0 / states) * np.exp(-states / (states - 1.0) * bl[v])
        keep = rng.random(P) < p_same
        other = (x[pid[v]] + rng.integers(1, states, P)) % states
        x[v] = np.where(keep, x[pid[v]], other)
    return np.ascontiguousarray(x[:n])


def two_internal_children(parent_ids, n):
    """Number of nodes of a parent-id vector whose children include two internal nodes (a
    trifurcating root with k internal children counts k - 1)."""
    parent_ids = np.asarray(parent_ids)
    kids = np.bincount(parent_ids[n:][parent_ids[n:] >= 0], minlength=len(parent_ids) + 1)
    return int(np.maximum(kids - 1, 0).sum())


def random_unrooted_topology(n, rng):
    """uniform random-join topology, numbered as the reference numbers nodes (leaves keep their
    ids, internal nodes in post-order, children ordered by max leaf id: node.cpp:32-59,341-357)"""
    parts = [(i, i, None) for i in range(n)]  # (max leaf, tag, children)
    while len(parts) > 3:
        i, j = sorted(rng.choice(len(parts), size=2, replace=False))
        b = parts.pop(j)
        a = parts.pop(i)
        parts.append((max(a[0], b[0]), -1, (a, b)))
    root = (max(p[0] for p in parts), -1, tuple(parts))
    parent, next_id = {}, [n]
    stack = [(root, False)]
    ids = {}
    while stack:
        node, done = stack.pop()
        if node[2] is None:
            ids[id(node)] = node[1]
            continue
        kids = sorted(node[2], key=lambda c: c[0])
        if not done:
            stack.append((node, True))
            for c in reversed(kids):
                stack.append((c, False))
        else:
            me = next_id[0]
            next_id[0] += 1
            ids[id(node)] = me
            for c in kids:
                parent[ids[id(c)]] = me
    return np.array([parent[v] for v in range(next_id[0] - 1)], dtype=np.int32)


def flua_workload(T, seed=46):
    """fluA.fa + fluA.tree (BASELINE.json configs[3]); T > 1: replicas with jittered height
    ratios (S-flu of SURVEY 8d)."""
    from libsbn_amd import _hostapi
    tc = _hostapi.TreeCollection.of_newick_file(os.path.join(DATA, "fluA.tree"))
    tips, w, _ = tc.site_pattern(os.path.join(DATA, "fluA.fa"))
    dates = tc.dates_from_taxon_names()  # already max - date (taxon_name_munging.cpp:46-78)
    pid, bl0 = tc.parent_ids[0], tc.branch_lengths[0]
    h0, b0, r0 = _hostapi.time_tree_from_branch_lengths(pid, bl0, dates)
    rng = np.random.default_rng(seed)
    n = len(dates)
    pids = np.tile(pid, (T, 1))
    bls, hs, bs, rs = [], [], [], []
    for t in range(T):
        r = r0.copy()
        if t > 0:
            r[:-1] = np.clip(r0[:-1] * np.exp(rng.normal(0, 0.05, n - 2)), 1e-3, 0.999)
            r[-1] = r0[-1] * np.exp(rng.normal(0, 0.02))
        bl, h, b = _hostapi.time_tree_from_height_ratios(pid, dates, r)
        bls.append(bl), hs.append(h), bs.append(b), rs.append(r)
    rates = np.full((T, 2 * n - 2), 0.001)
    return (tips, w, np.ascontiguousarray(pids), np.stack(bls), rates, np.ones(T, np.int32),
            np.stack(hs), np.stack(bs), np.stack(rs))


def usable_cores():
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU
    quota (running more threads than the quota only adds throttling stalls)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return cores


def oracle():
    """The CPU oracle (tests/oracle_lib.py over oracle/liboracle.so): the checker of
    `parity_checked` and the `cpu_baseline` leg -- never part of what is timed as `value`."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_lib as O
    return O


def beagle_probe():
    """is the reference's arithmetic library on this box?  (dlopen only; the
    image has no BEAGLE, no network to get it, and the reference pins no commit of it.)"""
    import ctypes
    import ctypes.util
    tried = ["libhmsbeagle.so", "libhmsbeagle.so.1", "libhmsbeagle.so.21", "libhmsbeagle-cpu.so",
             "libhmsbeagle-cpu-sse.so"]
    found = ctypes.util.find_library("hmsbeagle")
    if found:
        tried.insert(0, found)
    for name in tried:
        try:
            ctypes.CDLL(name)
            return {"found": True, "library": name}
        except OSError:
            continue
    return {"found": False, "tried": tried}


def cpu_baseline(tips, w, pids, bls, params, mode, budget_s=15.0, reps=10):
    """The CPU oracle (a port of the reference's algorithm, NOT BEAGLE itself: BEAGLE
    is not available in this image) timed on the host cores with the reference's
    tree-level threading model (one worker per core, FatBeagleParallelize).  SURVEY 8(d):
    >= 3 warm-ups, the MEDIAN of >= 10 repetitions of the engine call, on all usable cores
    and on one."""
    O = oracle()
    cores = usable_cores()
    spec = O.make_spec(tips.shape[0], tips.shape[1], "JC69", "weibull+4")
    fn = O.unrooted_gradients if mode == "gradient" else O.unrooted_log_likelihoods

    def take(count):  # the batch, cycled
        idx = np.arange(count) % len(pids)
        return pids[idx], bls[idx], params[idx]

    def timed(count, threads):
        a, b, c = take(count)
        t0 = time.perf_counter()
        fn(spec, tips, w, a, b, c, False, threads)
        return time.perf_counter() - t0

    S = 4 * cores
    for _ in range(3):  # warm-ups (page faults, thread start)
        dt = timed(S, cores)
    rate = S / dt
    # one repetition = one engine call over S2 trees, sized so that `reps` of them fit the budget
    S2 = max(cores, int(rate * budget_s * 0.8 / reps) // cores * cores)
    times = sorted(timed(S2, cores) for _ in range(reps))
    med = 0.5 * (times[(reps - 1) // 2] + times[reps // 2])
    S1 = max(1, int(rate / cores * budget_s * 0.2 / 3))  # and on ONE core: 3 short repetitions
    one = sorted(timed(S1, 1) for _ in range(3))
    return {"value": S2 / med, "unit": "trees/s", "cores": cores, "kind": "port",
            "repetitions": reps, "value_min": S2 / times[-1], "value_max": S2 / times[0],
            "value_on_one_core": S1 / one[1],
            "beagle_probe": beagle_probe(),
            "sample": f"median of {reps} calls x {S2} trees (the headline batch, cycled), {mode} "
                      f"semantics, {cores} threads, {sum(times):.0f} s; one core: 3 calls x {S1}",
            "sample_detail": f"3 warm-up calls; {cores} OpenMP threads = usable cores (affinity "
                             "mask capped by the cgroup CPU quota; host has "
                             f"{os.cpu_count()} logical CPUs), one tree per thread, one "
                             "workspace per thread; CPU oracle = BEAGLE-equivalent algorithm "
                             "(cache-blocked over site patterns) in plain C -O3 -march=native, "
                             "not BEAGLE (dlopen probe for libhmsbeagle in beagle_probe)"}


FP64_PEAK_DATASHEET_TFLOPS = 78.6


def executed_flops_per_tree(kname, n, P, K, R=3):
    """Flops of the FP64 matrix instructions a 4-state walk kernel EXECUTES for one tree, from
    the launch geometry (pattern tiles per tree) x the matrix instructions of a tile job, which
    follow from the tree size alone (checked against SQ_INSTS_MFMA of the committed counter
    tables: 385 / 628 / 208 per job for a 27-taxon tree) x 512 flops per v_mfma_f64_4x4x4_4b.
      look-up walk (gradient_walk_lut*):   4 (n-2) R + 2 R + 1 + 3 (n-1)   (tip children: none)
      mask-tip walk (gradient_walk_kernel):  (7 n - 8) R + 2 R + 1 + 3 (n-1)
      loglik_mfma (R = 4):                 2 (n-1) R + R + 1
    None for kernels that do not run on the matrix cores or for more than four categories."""
    if K > 4:
        return None
    kp = 1 if K == 1 else (2 if K == 2 else 4)
    if kname.startswith("gradient_walk_lut"):
        per_job = 4 * (n - 2) * R + 2 * R + 1 + 3 * (n - 1)
    elif kname.startswith("gradient_walk_kernel"):
        per_job = (7 * n - 8) * R + 2 * R + 1 + 3 * (n - 1)
    elif kname.startswith("loglik_mfma"):
        R = 4
        per_job = 2 * (n - 1) * R + R + 1
    else:
        return None
    per_tile = R * (16 // kp)
    return ((P + per_tile - 1) // per_tile) * per_job * 512.0


def pipe_entry(kname, launch_trees=None):
    """profiles/pipe.json (tools/pipe_table.py, from the committed SQ-counter tables): how busy
    the vector pipe was during a launch of this kernel -- static, not measured in this run."""
    try:
        table = json.load(open(os.path.join(REPO, "profiles", "pipe.json")))
    except (OSError, ValueError):
        return None
    base = kname.split("<")[0].split(" ")[0]
    want_grad = "true" in kname
    hits = []
    for key, ent in table.items():
        if not key.startswith(base):
            continue
        if base.startswith("aa_post") and (("true" in key.split(" ")[0]) != want_grad):
            continue
        if launch_trees is not None and " tree(s) per launch" in key and \
                f"-- {launch_trees} tree(s) per launch" not in key:
            continue
        hits.append((ent.get("round", ""), key, ent))
    if not hits:
        return None
    hits.sort(reverse=True)
    return hits[0][2]


def unified_roofline(kname, k_ms, units, alg_flops_per_unit, exec_flops_per_unit, traffic=None,
                     traffic_source=None, model_bytes_per_unit=None, pipe_trees=None, pipe=True):
    """ONE convention for every leg, 4-state and 20-state.  Per launch of the
    dominant kernel, `kernel_ms` from HIP events on the call's stream:
      frac_algorithmic  SURVEY 8(d) algorithmic flops x units / time / MEASURED FP64 matrix peak
                        (71.1 TFLOP/s; `_datasheet`: against 78.6).  Can exceed 1 where the kernel
                        does not execute what the model counts (tip children are table look-ups);
      frac_executed     flops of the matrix instructions really executed / time / measured peak;
      pipe_busy         (16|64 x MFMA + 4 x other vector instructions) / SIMD cycles of the launch,
                        STATIC: from the committed SQ-counter tables (profiles/pipe.json);
      hbm_frac          HBM-side bytes per launch (PMC; STATIC: profiles/traffic.json) / time / 8 TB/s.
    `bound` names the resource that binds -- "hbm" when hbm_frac exceeds the executed matrix
    fraction, else "mfma" (the SIMD's FP64 pipe) -- and `achieved` / `peak` / `unit` / `frac` are
    the task contract's fields for it: algorithmic flops (or real bytes) per launch / time."""
    sec = k_ms * 1e-3
    alg = alg_flops_per_unit * units / sec / 1e12
    exe = exec_flops_per_unit * units / sec / 1e12 if exec_flops_per_unit else None
    hbm_frac = traffic / sec / 1e9 / HBM_PEAK_GBPS if traffic else None
    # (pipe=False: this leg runs another instantiation of the kernel than the one whose SQ
    # counters are committed -- no figure rather than a borrowed one)
    pipe = pipe_entry(kname, pipe_trees) if pipe else None
    out = {"kernel": kname, "kernel_ms": k_ms, "units_per_launch": units,
           "frac_algorithmic": alg / FP64_PEAK_TFLOPS,
           "frac_executed": exe / FP64_PEAK_TFLOPS if exe is not None else None,
           "pipe_busy": pipe["pipe_busy"] if pipe else None,
           "hbm_frac": hbm_frac,
           "algorithmic_TFLOPs": alg, "executed_TFLOPs": exe,
           "peak_measured": FP64_PEAK_TFLOPS, "peak_datasheet": FP64_PEAK_DATASHEET_TFLOPS,
           "frac_algorithmic_datasheet": alg / FP64_PEAK_DATASHEET_TFLOPS,
           "frac_executed_datasheet": exe / FP64_PEAK_DATASHEET_TFLOPS if exe is not None else None,
           "fp64_peak_source": FP64_PEAK_SOURCE,
           "pipe_busy_source": ("static: " + pipe["source"] + " via profiles/pipe.json") if pipe else None,
           "traffic": traffic,
           "traffic_source": ("static: " + traffic_source) if traffic_source else None,
           "algorithmic_flops_per_unit": alg_flops_per_unit,
           "executed_flops_per_unit": exec_flops_per_unit}
    if model_bytes_per_unit:
        out["hbm_model_bytes_per_unit"] = model_bytes_per_unit
        out["hbm_model_over_peak"] = model_bytes_per_unit * units / sec / 1e9 / HBM_PEAK_GBPS
    pipe_frac = out["frac_executed"] if out["frac_executed"] is not None else out["frac_algorithmic"]
    if hbm_frac is not None and hbm_frac >= pipe_frac:
        out.update(bound="hbm", achieved=traffic / sec / 1e9, peak=HBM_PEAK_GBPS, unit="GB/s", frac=hbm_frac)
    else:
        out.update(bound="mfma", achieved=alg, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s",
                   frac=out["frac_algorithmic"])
    return out


def roofline(kname, k_ms, units, flops_per_unit, bytes_per_unit, traffic=None,
             traffic_source=None, shape=None, exec_flops_per_unit=None, pipe=True):
    """A 4-state leg.  The walk kernels keep partial vectors in LDS: what binds them is the
    SIMD's FP64 pipe.  shape = (n, P, K): the executed matrix flops follow from the kernel that
    ran and the launch geometry (executed_flops_per_tree); whole-call legs pass their own sum."""
    if exec_flops_per_unit is None and shape is not None:
        exec_flops_per_unit = executed_flops_per_tree(kname, *shape)
    return unified_roofline(kname, k_ms, units, flops_per_unit, exec_flops_per_unit, traffic,
                            traffic_source, bytes_per_unit, pipe=pipe)


def traffic_entry(name):
    """profiles/traffic.json: HBM-side bytes per launch of a kernel from the PMC passes
    ((2 FETCH_SIZE + WRITE_SIZE) x 1024, separate rocprofv3 --pmc runs) -- not this run.
    Keys are kernel names as rocprofv3 prints them (template arguments included), with
    "|T=<trees per launch>" appended for the 20-state kernels; `name` matches a key exactly or
    as its beginning (the latest round's entry first)."""
    tpath = os.path.join(REPO, "profiles", "traffic.json")
    try:
        table = json.load(open(tpath))
    except (OSError, ValueError):
        return None
    if name in table:
        return table[name]
    base, _, suffix = name.partition("|")
    hits = [k for k in table if k.startswith(base) and k.endswith("|" + suffix if suffix else "")
            and ("|" in k) == bool(suffix)]
    hits.sort(key=lambda k: table[k].get("round", ""), reverse=True)
    return table[hits[0]] if hits else None


def streamed_roofline(kname, k_ms, evals, traffic_key, mfma_flops_per_eval, model_bytes_per_eval,
                      alg_flops_per_eval, traffic_scale=1.0):
    """A 20-state leg (partial vectors stream through HBM): the same fields as every other leg
    (unified_roofline).  HBM-side bytes: the PMC figure of a launch of this size where one was
    profiled, else the one-tree figure x the evaluations of this launch (said so in
    traffic_source); traffic_scale: this run's share of the profiled alignment (pattern shards)."""
    ent = traffic_entry(f"{traffic_key}|T={evals}")
    if ent:
        traffic, src = ent["hbm_bytes_per_launch"], f"profiles/traffic.json [{traffic_key}|T={evals}] ({ent.get('round', '')}), measured at this launch size"
    else:
        ent = traffic_entry(f"{traffic_key}|T=1") or traffic_entry(traffic_key)
        traffic = ent["hbm_bytes_per_launch"] * evals if ent else None
        src = (f"profiles/traffic.json [{traffic_key}] one-tree launch x {evals} evaluations "
               f"({ent.get('round', '')})" if ent else None)
    if traffic is not None and traffic_scale != 1.0:
        traffic *= traffic_scale
        src += f", x {traffic_scale:.4f} (this run's share of the profiled alignment)"
    return unified_roofline(kname, k_ms, evals, alg_flops_per_eval, mfma_flops_per_eval, traffic, src,
                            model_bytes_per_eval, pipe_trees=evals, pipe=traffic_scale == 1.0)


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def parity(pairs, tol=1e-10):
    """Largest relative error of (name, device result, oracle result[, tolerance]) pairs;
    raises when an output of a timed step differs from the CPU oracle by more than its
    tolerance (1e-10 unless the pair states another)."""
    worst = 0.0
    for pair in pairs:
        name, dev, ref = pair[:3]
        t = pair[3] if len(pair) > 3 else tol
        e = rel_err(dev, ref)
        if not e <= t:
            raise AssertionError(f"timed outputs differ from the oracle: {name} rel err {e:.3e} > {t}")
        worst = max(worst, e)
    return worst


LINE_LIMIT = 4096  # bytes: the driver keeps an 8 KB tail of stdout and parses its last line


def _sig(x, digits=6):
    """x with every float rounded to `digits` significant digits (NaN / inf -> None)."""
    if isinstance(x, (bool, str)) or x is None:
        return x
    if isinstance(x, (float, np.floating)):
        x = float(x)
        return float(f"{x:.{digits}g}") if np.isfinite(x) else None
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if d and k in d}


ROOFLINE_KEYS = ("kernel", "kernel_ms", "units_per_launch", "bound", "achieved", "peak", "unit",
                 "frac", "frac_algorithmic", "frac_executed", "pipe_busy", "hbm_frac", "traffic",
                 "peak_measured", "peak_datasheet")
ALSO_ROOFLINE_KEYS = ("bound", "frac", "frac_algorithmic", "frac_executed", "pipe_busy", "hbm_frac")


def final_line(full):
    """The ONE JSON line the driver parses (last line of stdout), from the full result: the
    contract's fields, the dominant kernel's roofline, the CPU baseline, parity, and one short
    entry per other leg.  Everything else (phase tables, notes, per-kernel second rooflines,
    probe lists) is in bench_also.json and on the `BENCH_ALSO ...` lines printed before it.
    Raises if the line would exceed LINE_LIMIT bytes (a 21 KB line did not parse)."""
    keep = ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step",
            "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "dry_run", "backend",
            "trees_per_rank", "patterns_per_rank", "parity", "parity_checked",
            "parity_max_rel_err", "small_batch_ms", "host_pointer_trees_per_s",
            "adapter_trees_per_s", "also_error")
    out = _pick(full, keep)
    if "step_ms_device" in full:
        out["step_ms_device"] = _pick(full["step_ms_device"],
                                      ("min", "median", "max", "kernel_median"))
    if "config" in full:
        out["config"] = _pick(full["config"], ("workload", "trees_total", "trees_per_gpu", "taxa",
                                               "patterns", "categories", "trees", "parallelism"))
    if full.get("roofline"):
        out["roofline"] = _pick(full["roofline"], ROOFLINE_KEYS)
    if full.get("cpu_baseline"):
        out["cpu_baseline"] = _pick(full["cpu_baseline"],
                                    ("value", "unit", "cores", "kind", "repetitions",
                                     "value_on_one_core", "sample"))
    if full.get("small_batch"):
        out["small_batch"] = _pick(full["small_batch"],
                                   ("trees", "ms_per_step", "graph_ms_per_step", "kernel_ms",
                                    "graph_with_event_per_replay_ms_per_step",
                                    "graph_speedup_vs_full_batch", "graph_full_batch_ms_per_step",
                                    "graph_full_over_small",
                                    "graph_with_collective_ms_per_step"))
    if full.get("weak"):
        out["weak"] = _pick(full["weak"], ("value", "ms_per_step", "trees_per_gpu"))
    if full.get("reduced"):
        out["reduced"] = _pick(full["reduced"], ("trees_per_s", "ms_per_step", "all_reduce_doubles",
                                                 "collective", "max_rel_err"))
    if full.get("gathered"):
        out["gathered"] = _pick(full["gathered"], ("trees_per_s", "ms_per_step", "collective"))
    if full.get("also"):
        # one row per other leg, the SAME columns for every leg (4-state and 20-state alike; the
        # column names once: fifteen legs as objects would not fit the line); null = not applicable
        keys = ("workload", "trees_per_s", "parity_max_rel_err") + ALSO_ROOFLINE_KEYS + ("full_over_shard",)
        rows = []
        for a in full["also"]:
            r = a.get("roofline") or {}
            row = [a.get("short", a.get("workload", ""))[:40], _sig(a.get("trees_per_s"), 4),
                   _sig(a.get("parity_max_rel_err"), 2)]
            row += [r.get("bound")] + [_sig(r.get(k), 3) for k in ALSO_ROOFLINE_KEYS[1:]]
            row.append(_sig(a.get("full_over_shard"), 3))
            rows.append(row)
        out["also"] = {"keys": list(keys), "rows": rows}
    out["detail"] = "bench_also.json"
    line = json.dumps(_sig(out), separators=(",", ":"))
    if len(line) >= LINE_LIMIT:
        raise AssertionError(f"bench line is {len(line)} bytes (limit {LINE_LIMIT}): move "
                             "fields to bench_also.json")
    json.loads(line)
    return line


def emit(full):
    """bench_also.json (next to bench.py, and under gpurun_out/ when that exists) + the same
    content as `BENCH_ALSO <name> <json>` lines on stdout (one per leg, none starting with a
    brace) + the compact last line."""
    import ctypes
    for path in (os.path.join(REPO, "bench_also.json"),
                 os.path.join(REPO, "gpurun_out", "bench_also.json")):
        try:
            if os.path.isdir(os.path.dirname(path)):
                with open(path, "w") as fh:
                    json.dump(full, fh, indent=1)
        except OSError:
            pass
    # RCCL prints a version banner through C stdio, which is flushed at exit when stdout is a
    # file: flush it now so that the JSON line is the last line.
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    for key in ("roofline", "step_ms_device", "small_batch", "weak", "gathered", "reduced",
                "host_pointer", "adapter", "cpu_baseline", "config", "rank_devices"):
        if full.get(key) is not None:
            print("BENCH_ALSO", key, json.dumps(_sig(full[key], 8)), flush=True)
    for leg in full.get("also") or []:
        print("BENCH_ALSO", "leg", json.dumps(_sig(leg, 8)), flush=True)
    print(final_line(full), flush=True)  # the ONE JSON line, last thing on stdout


class Timed:
    """Times `steps` calls of fn() after `warmup` calls; kernel time from the engine's HIP
    events around its dominant kernel(s); optionally a second, separate pass with the call cut
    into phases (mi_engine_profile_begin_phases: a few more events per call, so never inside
    the timed region)."""

    def __init__(self, torch, eng, stream):
        self.torch, self.eng, self.stream = torch, eng, stream

    def run(self, fn, steps, warmup):
        torch, eng = self.torch, self.eng
        for _ in range(warmup):
            fn()
        eng.check_status(self.stream)
        torch.cuda.synchronize()
        eng.profile_begin(steps)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        k = eng.profile_collect(steps)
        eng.check_status(self.stream)
        return 1e3 * dt, (float(np.mean(k)) if k else float("nan"))

    def phases(self, fn, calls=2):
        """median [set-up, post-order, pre-order / main walk, rest] ms and the evaluations of
        the first walk launch"""
        self.eng.profile_begin_phases(calls)
        for _ in range(calls):
            fn()
        self.torch.cuda.synchronize()
        _, ph, first = self.eng.profile_collect_phases(calls)
        return [float(np.median([p[k] for p in ph])) for k in range(4)], first


def also_workloads(torch, dev, L, steps):
    """The other BASELINE.json configurations, one GPU, short runs (rank 0 only).  Every leg
    compares a sample of its last timed step with the CPU oracle (`parity_checked` trees,
    tolerance 1e-10 unless stated)."""
    out = []
    stream = torch.cuda.current_stream().cuda_stream
    O = oracle()
    cores = usable_cores()

    def dev_arrays(*arrays):
        return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrays]

    def host(t, count=None):
        return (t if count is None else t[:count]).cpu().numpy()

    # --- DS1 x 1000: GTR+weibull+4 (configs[2]), log-likelihoods, the 1949-pattern shape
    T = 1000
    tips, w, pids, bls = ds1_workload(T)
    n, P = tips.shape
    N, K = 2 * n - 1, 4
    rng = np.random.default_rng(45)
    gtr = np.hstack([rng.dirichlet(10 * np.ones(6), T), rng.dirichlet(10 * np.ones(4), T),
                     np.ones((T, 2))])
    jc = np.ones((T, 2))
    d_pid, d_bl, d_gtr, d_jc = dev_arrays(pids, bls, gtr, jc)
    ll = torch.empty(T, dtype=torch.float64, device=dev)
    g = torch.empty((T, N), dtype=torch.float64, device=dev)
    site = torch.empty(T, dtype=torch.float64, device=dev)
    sub = torch.empty((T, 8), dtype=torch.float64, device=dev)
    b_ll, b_g = algorithmic_bytes(n, P, K)
    f_ll, f_g = algorithmic_flops(n, P, K)
    eng = L.Engine(L.PhyloModelSpecification("GTR", "weibull+4", "strict"), tips, w,
                   device=dev.index)
    eng.reserve(T, True)
    tm = Timed(torch, eng, stream)
    S = 4  # trees of each leg that are compared with the oracle
    ospec = O.make_spec(n, P, "GTR", "weibull+4")
    # the engine forms I + V expm1(L t) V^-1; the oracle's default is
    # BEAGLE's V exp(L t) V^-1, whose rounding the 2e-6 finite-difference step amplifies
    O.set_transition_mode(1)
    try:
        og = O.unrooted_gradients(ospec, tips, w, pids[:S], bls[:S], gtr[:S], False, min(S, cores))
    finally:
        O.set_transition_mode(0)
    for label, s_ptr, u_ptr, bytes_, flops, full in (
            ("full phylo_gradients (reference semantics: 16 finite-difference passes + "
             "perturbed-model site pass)", site.data_ptr(), sub.data_ptr(),
             2 * b_g + 16 * b_ll, 2 * f_g + 16 * f_ll, True),
            ("logL + branch-length gradient only (configs[2] as worded)", None, None, b_g, f_g,
             False)):
        short = "DS1x1000 GTR+G4 " + ("full gradients (FD)" if full else "logL+branch gradient")
        ms, k_ms = tm.run(lambda: eng.gradients_device(
            stream, T, d_pid.data_ptr(), d_bl.data_ptr(), d_gtr.data_ptr(), ll.data_ptr(),
            g.data_ptr(), s_ptr, u_ptr), steps, 1)
        kname = eng.last_call_info()[0]
        pairs = [("logL", host(ll, S), og["log_likelihood"]),
                 ("branch gradient", host(g, S), og["branch_lengths"])]
        if full:
            # central differences with step 1e-6 of two FP64 log-likelihoods ~ -7e3: both sides
            # carry ~1e-16 * 7e3 / 2e-6 = 4e-7 of absolute rounding noise
            pairs += [("site gradient", host(site, S), og["site_model"], 1e-8),
                      ("substitution gradient (finite differences)", host(sub, S),
                       og["substitution_model"], 1e-6)]
        err = parity(pairs)
        # the whole call against the FP64 roofline: every pass of the call is a walk kernel
        # the whole call against the FP64 pipe: every pass of the call is a walk kernel
        x_g, x_ll = executed_flops_per_tree(kname, n, P, K), executed_flops_per_tree("loglik_mfma", n, P, K)
        r = roofline(kname, ms, T, flops, bytes_,
                     exec_flops_per_unit=(2 * x_g + 16 * x_ll) if full else x_g)
        r["note"] = "whole call (all its walk passes) over the step time"
        out.append({"workload": f"DS1 27 taxa x {P} patterns x {T} trees, GTR+weibull+4, " + label,
                    "short": short, "trees_per_s": T / (ms * 1e-3), "ms_per_step": ms,
                    "kernel": kname,
                    "kernel_ms": k_ms,
                    "kernel_note": "HIP events around the main gradient pass only; the call also "
                                   "runs the finite-difference / site passes",
                    "roofline": r, "parity_checked": S, "parity_max_rel_err": err,
                    "parity_note": "vs CPU oracle (expm1 transition form): logL, branch gradient "
                                   "1e-10" + ("; site 1e-8, substitution 1e-6 (finite-difference "
                                              "noise of both sides)" if full else "")})
    assert bool(torch.isfinite(sub).all())
    eng.close()
    eng = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w,
                   device=dev.index)
    eng.reserve(T, False)
    tm = Timed(torch, eng, stream)
    ms, k_ms = tm.run(lambda: eng.log_likelihoods_device(
        stream, T, d_pid.data_ptr(), d_bl.data_ptr(), d_jc.data_ptr(), ll.data_ptr()), steps, 1)
    kname = eng.last_call_info()[0]
    S = 16
    oll = O.unrooted_log_likelihoods(O.make_spec(n, P, "JC69", "weibull+4"), tips, w, pids[:S],
                                     bls[:S], jc[:S], False, min(S, cores))
    err = parity([("logL", host(ll, S), oll)])
    ent = traffic_entry(kname.split("<")[0])
    out.append({"workload": f"DS1 27 taxa x {P} patterns x {T} trees, JC69+weibull+4, log_likelihoods",
                "short": "DS1x1000 JC69+G4 log_likelihoods", "trees_per_s": T / (ms * 1e-3),
                "ms_per_step": ms, "kernel": kname,
                "kernel_ms": k_ms,
                "roofline": roofline(kname, k_ms, T, f_ll, b_ll,
                                     ent["hbm_bytes_per_launch"] if ent else None,
                                     "profiles/traffic.json (PMC, 1000-tree launch), not this run",
                                     shape=(n, P, K)),
                "parity_checked": S, "parity_max_rel_err": err})
    eng.close()
    # S-DS1, 1949 patterns (every column of a DS1-sized alignment its own pattern)
    rng = np.random.default_rng(44)
    tips2 = evolved_alignment(pids[0], bls[0], 1949, rng)
    w2 = np.ones(1949)
    eng = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips2, w2,
                   device=dev.index)
    eng.reserve(T, True)
    tm = Timed(torch, eng, stream)
    ms, k_ms = tm.run(lambda: eng.gradients_device(
        stream, T, d_pid.data_ptr(), d_bl.data_ptr(), d_jc.data_ptr(), ll.data_ptr(),
        g.data_ptr(), site.data_ptr(), None), steps, 1)
    kname = eng.last_call_info()[0]
    b2_ll, b2_g = algorithmic_bytes(n, 1949, K)
    f2_ll, f2_g = algorithmic_flops(n, 1949, K)
    S = 8
    og = O.unrooted_gradients(O.make_spec(n, 1949, "JC69", "weibull+4"), tips2, w2, pids[:S],
                              bls[:S], jc[:S], False, min(S, cores))
    err = parity([("logL", host(ll, S), og["log_likelihood"]),
                  ("branch gradient", host(g, S), og["branch_lengths"]),
                  ("site gradient", host(site, S), og["site_model"])])
    out.append({"workload": f"S-DS1 27 taxa x 1949 patterns (synthetic, evolved under JC) x {T} "
                            "trees, JC69+weibull+4, phylo_gradients",
                "short": "S-DS1 1949 patterns x1000 gradients", "trees_per_s": T / (ms * 1e-3),
                "ms_per_step": ms, "kernel": kname,
                "kernel_ms": k_ms, "roofline": roofline(kname, k_ms, T, f2_g, b2_g, shape=(n, 1949, K)),
                "parity_checked": S, "parity_max_rel_err": err})
    eng.close()

    # --- a standard alignment's SHAPE beyond what LDS holds (DS3: 36 taxa, 1812 patterns): random
    # topologies, synthetic alignment -- the look-up walk's arena variant with wide pattern tiles
    rng = np.random.default_rng(46)
    n3, P3 = 36, 1812
    top3 = np.stack([random_unrooted_topology(n3, rng) for _ in range(50)])
    pid3 = np.ascontiguousarray(np.tile(top3, (T // 50 + 1, 1))[:T]).astype(np.int32)
    bl3 = rng.exponential(0.1, size=(T, 2 * n3 - 2))
    bl3[:, -1] = 0
    tips3 = evolved_alignment(pid3[0], bl3[0], P3, rng)
    w3 = np.ones(P3)
    d_pid3, d_bl3 = dev_arrays(pid3, bl3)
    N3 = 2 * n3 - 1
    ll3 = torch.empty(T, dtype=torch.float64, device=dev)
    g3 = torch.empty((T, N3), dtype=torch.float64, device=dev)
    eng = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips3, w3,
                   device=dev.index)
    eng.reserve(T, True)
    tm = Timed(torch, eng, stream)
    ms, k_ms = tm.run(lambda: eng.gradients_device(
        stream, T, d_pid3.data_ptr(), d_bl3.data_ptr(), d_jc.data_ptr(), ll3.data_ptr(),
        g3.data_ptr(), site.data_ptr(), None), steps, 1)
    kname = eng.last_call_info()[0]
    path3 = eng.last_call_path()
    b3_ll, b3_g = algorithmic_bytes(n3, P3, K)
    f3_ll, f3_g = algorithmic_flops(n3, P3, K)
    S = 4
    og = O.unrooted_gradients(O.make_spec(n3, P3, "JC69", "weibull+4"), tips3, w3, pid3[:S],
                              bl3[:S], jc[:S], False, min(S, cores))
    err = parity([("logL", host(ll3, S), og["log_likelihood"]),
                  ("branch gradient", host(g3, S), og["branch_lengths"]),
                  ("site gradient", host(site, S), og["site_model"])])
    out.append({"workload": f"S-DS3 shape: {n3} taxa x {P3} patterns (synthetic, evolved under JC), {T} "
                            f"trees of 50 random topologies, JC69+weibull+4, phylo_gradients [{path3}]",
                "short": "S-DS3 36x1812 x1000 gradients", "trees_per_s": T / (ms * 1e-3),
                "ms_per_step": ms, "kernel": kname,
                "kernel_ms": k_ms,
                "roofline": roofline(kname, k_ms, T, f3_g, b3_g, pipe=False,
                                     exec_flops_per_unit=executed_flops_per_tree(
                                         kname, n3, P3, K, R=4 if "tile=wide" in path3 else 3)),
                "parity_checked": S, "parity_max_rel_err": err})
    eng.close()

    # --- fluA rooted, strict clock (configs[3])
    for Tf in (1, 1000):
        tips, w, pids, bls, rates, counts, hs, bs, rs = flua_workload(Tf)
        n, P = tips.shape
        N = 2 * n - 1
        eng = L.Engine(L.PhyloModelSpecification("JC69", "constant", "strict"), tips, w,
                       device=dev.index)
        pr1 = np.ones((Tf, 1))
        d = dev_arrays(pids, bls, pr1, rates, counts, hs, bs, rs)
        ll = torch.empty(Tf, dtype=torch.float64, device=dev)
        gr = torch.empty((Tf, n - 1), dtype=torch.float64, device=dev)
        gc = torch.empty((Tf, N - 1), dtype=torch.float64, device=dev)
        lib, h = eng._lib, eng._h
        eng.reserve(Tf, True)

        def call():
            rc = lib.mi_engine_gradients_rooted_device(
                h, stream, Tf, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(), d[6].data_ptr(),
                d[7].data_ptr(), 0, ll.data_ptr(), gr.data_ptr(), gc.data_ptr(), None, None)
            if rc:
                raise RuntimeError(eng._check(rc))
        tm = Timed(torch, eng, stream)
        ms, k_ms = tm.run(call, steps if Tf > 1 else 20, 2)
        kname = eng.last_call_info()[0]
        phase_ms, _ = tm.phases(call, 4)
        bf_ll, bf_g = algorithmic_bytes(n, P, 1)
        ff_ll, ff_g = algorithmic_flops(n, P, 1)
        assert bool(torch.isfinite(gr).all())
        S = min(Tf, 8)
        og = O.rooted_gradients(O.make_spec(n, P, "JC69", "constant"), tips, w, pids[:S], bls[:S],
                                pr1[:S], rates[:S], counts[:S], hs[:S], bs[:S], rs[:S], False,
                                min(S, cores))
        err = parity([("logL", host(ll, S), og["log_likelihood"]),
                      ("ratio / root-height gradient", host(gr, S), og["ratios_root_height"]),
                      ("clock gradient", host(gc, S)[:, 0], og["clock_model"][:, 0])])
        out.append({"workload": f"fluA rooted {n} taxa x {P} patterns x {Tf} tree(s), JC69, strict "
                                "clock, node-height-ratio + clock gradient",
                    "short": f"fluA rooted x{Tf} ratio+clock gradient",
                    "trees_per_s": Tf / (ms * 1e-3), "ms_per_step": ms, "kernel": kname,
                    "kernel_ms": k_ms, "logL0": float(ll[0]),
                    "phase_ms": {"setup": phase_ms[0], "walk": phase_ms[2], "rest": phase_ms[3]},
                    "roofline": roofline(kname, k_ms, Tf, ff_g, bf_g, shape=(n, P, 1), pipe=False),
                    "parity_checked": S, "parity_max_rel_err": err})
        eng.close()

    # --- 20 states, 512 taxa x 50 000 patterns x 4 categories (configs[4], S-WAG)
    n, P, K = 512, 50000, 4
    rng = np.random.default_rng(47)
    tips = rng.integers(0, 20, size=(n, P)).astype(np.int32)
    w = np.ones(P)
    eng = L.Engine(L.PhyloModelSpecification("WAG", f"weibull+{K}", "strict"), tips, w,
                   device=dev.index)
    N = 2 * n - 1
    plv = K * P * 20 * 8
    pids8 = np.stack([random_unrooted_topology(n, rng) for _ in range(8)])
    bls8 = rng.exponential(0.1, size=(8, 2 * n - 2))
    bls8[:, -1] = 0
    # tree 0 of both batch sizes against the oracle: the alignment cut into pattern blocks,
    # one block per host thread, the blocks added (every output is a sum over patterns)
    import libsbn_amd.engine as E
    ex, fr = E.wag_model()
    O.set_reversible_model(ex, fr)
    O.set_transition_mode(1)
    try:
        t0 = time.perf_counter()
        og = O.unrooted_by_pattern_blocks(O.make_spec(n, P, "reversible", f"weibull+{K}", s=20),
                                          tips, w, pids8[:1], bls8[:1], np.ones((1, 2)),
                                          rescaling=True, gradient=cores >= 8, threads=cores)
        oracle_s = time.perf_counter() - t0
    finally:
        O.set_transition_mode(0)
    def swag_legs(eng, Pe, Tw, og, oracle_note, scale, what):
        """log_likelihoods and phylo_gradients of Tw trees on an engine of Pe site patterns (scale =
        Pe / P: its share of the full alignment), tree 0 against the oracle results `og`."""
        legs = []
        plv = K * Pe * 20 * 8
        pids, bls = pids8[:Tw], bls8[:Tw]
        d_pid, d_bl, d_par = dev_arrays(pids, bls, np.ones((Tw, 2)))
        ll = torch.empty(Tw, dtype=torch.float64, device=dev)
        g = torch.empty((Tw, N), dtype=torch.float64, device=dev)
        site = torch.empty(Tw, dtype=torch.float64, device=dev)
        tm = Timed(torch, eng, stream)
        for grad in (False, True):
            eng.reserve(Tw, grad)
            if grad:
                fn = lambda: eng.gradients_device(  # noqa: E731
                    stream, Tw, d_pid.data_ptr(), d_bl.data_ptr(), d_par.data_ptr(),
                    ll.data_ptr(), g.data_ptr(), site.data_ptr(), None)
            else:
                fn = lambda: eng.log_likelihoods_device(  # noqa: E731
                    stream, Tw, d_pid.data_ptr(), d_bl.data_ptr(), d_par.data_ptr(), ll.data_ptr())
            ms, k_ms = tm.run(fn, (3 if Tw == 1 else 2) if scale == 1.0 else 10, 1)
            phase_ms, first = tm.phases(fn, 2)
            launches = eng.last_call_launches()[0]
            assert bool(torch.isfinite(ll).all())
            pairs = [("logL of tree 0", host(ll, 1), og["log_likelihood"])]
            if grad and "branch_lengths" in og:
                pairs += [("branch gradient of tree 0", host(g, 1), og["branch_lengths"]),
                          ("site gradient of tree 0", host(site, 1), og["site_model"])]
            err = parity(pairs)
            b_model = ((10 * n - 14) if grad else 2 * (n - 1)) * plv + (12 if grad else 4) * n * Pe
            f_ll_aa, f_g_aa = algorithmic_flops(n, Pe, K, s=20)
            # what the kernels really have to do: a tip child's product is a table look-up,
            # so only the n-2 internal edges cost 20x20 products: one each in the post-order
            # pass; in the pre-order pass two each, plus P L once per node with two internal
            # children (the child the post-order pass did NOT take from registers)
            prod = K * Pe * 800.0
            post = streamed_roofline(
                "aa_post_wg_kernel<2," + ("true" if grad else "false") + "> (+ aa_root_kernel)",
                phase_ms[1], first,
                "aa_post_wg_kernel<2, true>" if grad else "aa_post_wg_kernel<2, false>",
                (n - 2) * prod, (2 * (n - 1)) * plv + 4 * n * Pe, f_ll_aa, traffic_scale=scale)
            entry = {"workload": f"S-WAG 20 states, {n} taxa x {Pe} patterns x {K} categories "
                                 f"x {Tw} tree(s){what}, " + ("phylo_gradients" if grad else
                                                              "log_likelihoods"),
                     "short": f"S-WAG 512x{Pe}x4 T={Tw} " + ("gradients" if grad else "logL"),
                     "trees_per_s": Tw / (ms * 1e-3), "ms_per_step": ms,
                     "kernel": eng.last_call_info()[0], "kernel_ms": k_ms,
                     "path": eng.last_call_path(),
                     "walk_launches": launches, "evaluations_in_first_launch": first,
                     "logL0": float(ll[0]),
                     "phase_ms": {"setup": phase_ms[0], "post_order": phase_ms[1],
                                  "pre_order": phase_ms[2], "rest": phase_ms[3]},
                     "parity_checked": 1, "parity_max_rel_err": err,
                     "parity_note": "tree 0 vs the CPU oracle evaluated on pattern blocks "
                                    f"({oracle_note}): "
                                    + ("logL, branch and site gradients" if grad and
                                       "branch_lengths" in og else "logL")}
            if grad:
                pre = streamed_roofline(
                    "aa_pre_wg_kernel<2>", phase_ms[2], first, "aa_pre_wg_kernel<2>",
                    (2 * (n - 2) + two_internal_children(pids[0], n)) * prod,
                    (8 * n - 12) * plv + 8 * n * Pe, f_g_aa - f_ll_aa, traffic_scale=scale)
                entry["roofline"] = pre  # the dominant kernel of a gradient call
                entry["roofline_post_order"] = post
            else:
                entry["roofline"] = post
            legs.append(entry)
        return legs

    full = {}
    for Tw in (1, 8):
        legs = swag_legs(eng, P, Tw, og, f"{oracle_s:.0f} s on {cores} threads", 1.0, "")
        full[Tw] = legs
        out += legs
    eng.close()
    # --- configs[4] is worded "8 x MI355X": a rank's share of it on ONE GPU (VERDICT r5 item 2;
    # what `small_batch` is for DS1).  Two ways to deal it to 8 ranks (SURVEY 8(e);
    # /root/reference/src/fat_beagle.hpp:134-147): (i) 8 trees, tree-sharded -- a rank runs ONE tree on
    # the whole alignment (the T = 1 legs above); (ii) few trees, pattern-sharded -- a rank runs
    # every tree on 50 000 / 8 = 6 250 patterns (engine built from a rank's block, as
    # libsbn_amd/sharding.py deals them) and ONE all-reduce adds the per-tree results.
    # full_over_shard = time of the whole job on one GPU / time of a rank's share: the speed-up 8
    # GPUs would give before the collective.
    Ps = P // 8
    tips_s, w_s = np.ascontiguousarray(tips[:, :Ps]), np.ascontiguousarray(w[:Ps])
    O.set_reversible_model(ex, fr)
    O.set_transition_mode(1)
    try:
        t0 = time.perf_counter()
        og_s = O.unrooted_by_pattern_blocks(O.make_spec(n, Ps, "reversible", f"weibull+{K}", s=20),
                                            tips_s, w_s, pids8[:1], bls8[:1], np.ones((1, 2)),
                                            rescaling=True, gradient=cores >= 8, threads=cores)
        oracle_shard_s = time.perf_counter() - t0
    finally:
        O.set_transition_mode(0)
    eng = L.Engine(L.PhyloModelSpecification("WAG", f"weibull+{K}", "strict"), tips_s, w_s,
                   device=dev.index)
    shard = {}
    for Tw in (1, 8):
        legs = swag_legs(eng, Ps, Tw, og_s, f"{oracle_shard_s:.1f} s on {cores} threads", Ps / P,
                         " -- one rank's pattern block of eight")
        shard[Tw] = legs
        out += legs
    eng.close()
    for Tw in (1, 8):
        for i, what in ((0, "logL"), (1, "gradients")):
            shard[Tw][i]["full_over_shard"] = full[Tw][i]["ms_per_step"] / shard[Tw][i]["ms_per_step"]
            shard[Tw][i]["full_over_shard_note"] = (
                f"T = {Tw} on 50 000 patterns / T = {Tw} on a rank's 6 250: 8-way PATTERN sharding "
                "before the all-reduce")
    for i in (0, 1):
        full[1][i]["full_over_shard"] = full[8][i]["ms_per_step"] / full[1][i]["ms_per_step"]
        full[1][i]["full_over_shard_note"] = ("T = 8 / T = 1 on the whole alignment: 8-way TREE "
                                              "sharding of an 8-tree batch before the all-gather")
    return out


def swag_pattern_sharded(args, torch, dist, L, sharding, dev, rank, world, distributed):
    """BASELINE.json configs[4] over several GPUs: few trees x a very long alignment.  Every
    rank builds its engine from a contiguous block of site patterns
    (sharding.pattern_shard), evaluates ALL trees on it, and ONE all-reduce sums the packed
    per-tree results [logL | site gradient | branch gradient] -- every output of the
    unrooted path is a sum over site patterns.  Strong scaling: the alignment is fixed."""
    n, P, K, T = 512, 50000, 4, args.swag_trees
    N = 2 * n - 1
    rng = np.random.default_rng(47)
    tips = rng.integers(0, 20, size=(n, P)).astype(np.int32)
    w = np.ones(P)
    pids = np.stack([random_unrooted_topology(n, rng) for _ in range(T)])
    bls = rng.exponential(0.1, size=(T, 2 * n - 2))
    bls[:, -1] = 0
    lo, hi = sharding.pattern_shard(P, rank, world)
    eng = L.Engine(L.PhyloModelSpecification("WAG", f"weibull+{K}", "strict"),
                   np.ascontiguousarray(tips[:, lo:hi]), w[lo:hi], device=dev.index)
    grad = args.mode == "gradient"
    d_pid = torch.from_numpy(pids).to(dev)
    d_bl = torch.from_numpy(bls).to(dev)
    d_par = torch.ones((T, 2), dtype=torch.float64, device=dev)
    blk = sharding.ResultBlocks(T, N, extra=1, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    eng.reserve(T, grad)

    def step():
        if grad:
            eng.gradients_device(stream, T, d_pid.data_ptr(), d_bl.data_ptr(), d_par.data_ptr(),
                                 blk.log_likelihoods.data_ptr(), blk.branch_gradients.data_ptr(),
                                 blk.extras[0].data_ptr(), None)
        else:
            eng.log_likelihoods_device(stream, T, d_pid.data_ptr(), d_bl.data_ptr(),
                                       d_par.data_ptr(), blk.log_likelihoods.data_ptr())
        if distributed:
            sharding.all_reduce_pattern_shards(blk.buffer)

    steps, warmup = max(1, min(args.steps, 5)), max(1, min(args.warmup, 2))
    for _ in range(warmup):
        step()
    eng.check_status(stream)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    eng.profile_begin(steps)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    k_ms = float(np.mean(eng.profile_collect(steps)))
    eng.check_status(stream)
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    assert bool(torch.isfinite(blk.buffer).all())
    # a separate pass with the call cut into phases (never inside the timed region)
    eng.profile_begin_phases(2)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    _, ph, first = eng.profile_collect_phases(2)
    phase_ms = [float(np.median([p[k] for p in ph])) for k in range(4)]
    if distributed:
        dist.barrier()
    if rank != 0:
        return None
    plv = K * P * 20 * 8
    share = (hi - lo) / P  # this rank's part of the alignment
    b_model = ((10 * n - 14) if grad else 2 * (n - 1)) * plv + (12 if grad else 4) * n * P
    ms = 1e3 * elapsed / steps
    prod = K * P * 800.0 * share
    f_ll_aa, f_g_aa = algorithmic_flops(n, P, K, s=20)
    post = streamed_roofline(
        "aa_post_wg_kernel<2," + ("true" if grad else "false") + "> (+ aa_root_kernel)",
        phase_ms[1], first,
        "aa_post_wg_kernel<2, true>" if grad else "aa_post_wg_kernel<2, false>",
        (n - 2) * prod, ((2 * (n - 1)) * plv + 4 * n * P) * share, f_ll_aa * share,
        traffic_scale=share)
    rl = post
    if grad:
        rl = streamed_roofline("aa_pre_wg_kernel<2>", phase_ms[2], first, "aa_pre_wg_kernel<2>",
                               (2 * (n - 2) + two_internal_children(pids[0], n)) * prod,
                               ((8 * n - 12) * plv + 8 * n * P) * share,
                               (f_g_aa - f_ll_aa) * share, traffic_scale=share)
    return {
        "metric": "tree log-likelihoods+gradients/sec (batched)" if grad
                  else "tree log-likelihoods/sec (batched)",
        "value": T / (elapsed / steps), "unit": "trees/s", "n_gpus": world,
        "rccl_ranks": dist.get_world_size() if distributed else 1,
        "patterns_per_rank": [sharding.pattern_shard(P, r, world)[1]
                              - sharding.pattern_shard(P, r, world)[0] for r in range(world)],
        "steps": steps,
        "warmup": warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64",
        "data": "synthetic: uniform 20-state tips, random-join topologies (seed 47), branch "
                "lengths Exp(mean 0.1), built-in WAG table",
        "config": {"workload": f"S-WAG 20 states, {n} taxa x {P} patterns x {K} categories x {T} "
                               f"trees, site patterns dealt to {world} GPU(s) "
                               f"({hi - lo} on rank 0), one all_reduce per step, "
                               + ("phylo_gradients" if grad else "log_likelihoods"),
                   "taxa": n, "patterns": P, "categories": K, "trees": T,
                   "parallelism": f"pattern-sharded x{world}, one all_reduce per step"
                                  if distributed else "single GPU"},
        "roofline": rl, "roofline_post_order": post if grad else None,
        "walk_kernels_ms": k_ms,
        "phase_ms": {"setup": phase_ms[0], "post_order": phase_ms[1], "pre_order": phase_ms[2],
                     "rest": phase_ms[3]},
        "hbm_model_bytes_per_tree": b_model,
        "logL0": float(blk.log_likelihoods[0]),
    }


def host_pointer_leg(eng, pids, bls, params, head_ll, head_g, head_site, steps, warmup):
    """mi_engine_gradients_unrooted: host pointers in, host pointers out (what the adapter of
    INTEGRATION.md calls behind Engine::Gradients, /root/reference/src/engine.cpp:78-84), on
    the headline batch; then the Python mirror's Engine.gradients (one PhyloGradient per
    tree).  Results must equal the device-resident call's bit for bit."""
    import ctypes as C
    T, N = len(pids), head_g.shape[1]
    pid = np.ascontiguousarray(pids, dtype=np.int32)
    bl = np.ascontiguousarray(bls, dtype=np.float64)
    pr = np.ascontiguousarray(params, dtype=np.float64)
    ll, g, site = np.empty(T), np.empty((T, N)), np.empty(T)
    ptr = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731

    def call():
        rc = eng._lib.mi_engine_gradients_unrooted(eng._h, T, ptr(pid), ptr(bl), ptr(pr), 0,
                                                   ptr(ll), ptr(g), ptr(site), None)
        if rc:
            raise RuntimeError(eng._check(rc))

    for _ in range(warmup):
        call()
    t0 = time.perf_counter()
    for _ in range(steps):
        call()
    sec = (time.perf_counter() - t0) / steps
    same = bool(np.array_equal(ll, head_ll) and np.array_equal(g, head_g)
                and np.array_equal(site, head_site))
    assert same, "host-pointer call differs from the device-resident call"
    for _ in range(min(warmup, 2)):
        eng.gradients(pid, bl, pr)
    t0 = time.perf_counter()
    for _ in range(steps):
        res = eng.gradients(pid, bl, pr)
    py_sec = (time.perf_counter() - t0) / steps
    assert res[0].log_likelihood == head_ll[0]
    return {"trees_per_s": T / sec, "ms_per_step": 1e3 * sec, "trees": T,
            "python_mirror_trees_per_s": T / py_sec, "python_mirror_ms_per_step": 1e3 * py_sec,
            "bit_identical_to_device_call": same,
            "note": "mi_engine_gradients_unrooted with numpy buffers: upload (parent ids, "
                    "branch lengths, parameters), kernels, download (logL, branch and site "
                    "gradients) through the engine's pinned staging, one synchronisation per "
                    "call, wall clock; python mirror = Engine.gradients -> one PhyloGradient "
                    "per tree"}


def adapter_leg(bls, head_ll, head_g, head_site, steps, warmup):
    """Engine::Gradients of the C++ adapter (libsbn_amd/csrc/host/engine.hpp), built here with
    g++ and run as a child process on the headline batch: std::vector<FlatTree> in,
    std::vector<PhyloGradient> out (tools/engine_bench.cpp)."""
    import subprocess
    import tempfile
    lib = os.path.join(REPO, "libsbn_amd")
    try:
        with tempfile.TemporaryDirectory() as tmp:
            exe, blf, outf = (os.path.join(tmp, x) for x in ("engine_bench", "bl.f64", "out.f64"))
            subprocess.run(["g++", "-std=c++17", "-O2", os.path.join(REPO, "tools", "engine_bench.cpp"),
                            "-L" + lib, "-lmi_phylo", "-lmi_phylo_host", "-Wl,-rpath," + lib,
                            "-o", exe], check=True, capture_output=True, text=True, timeout=300)
            np.ascontiguousarray(bls, dtype=np.float64).tofile(blf)
            T = len(bls)
            r = subprocess.run([exe, os.path.join(DATA, "DS1.fasta"),
                                os.path.join(DATA, "DS1.100_topologies.nwk"), blf, str(T),
                                str(steps), str(warmup), outf],
                               capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                return {"error": (r.stdout + r.stderr)[-300:]}
            res = json.loads(r.stdout.strip().splitlines()[-1])
            got = np.fromfile(outf).reshape(T, -1)
            same = bool(np.array_equal(got[:, 0], head_ll) and np.array_equal(got[:, 1:-1], head_g)
                        and np.array_equal(got[:, -1], head_site))
            assert same, "the adapter's results differ from the device-resident call"
            res["bit_identical_to_device_call"] = same
            res["note"] = ("tools/engine_bench.cpp: mihost::Engine::Gradients(std::vector<FlatTree>, "
                           "ParamMatrix, false) -> std::vector<PhyloGradient> (flattening the "
                           "trees, the C ABI's host-pointer call, one std::map of vectors per "
                           "tree), wall clock in a child process")
            return res
    except (OSError, subprocess.SubprocessError, ValueError) as exc:
        return {"error": repr(exc)[:300]}


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def spawn_ranks(gpus):
    """Runs `python -m torch.distributed.run --nproc-per-node <gpus> bench.py <same args>` as a
    child and returns its exit code; the child's stdout (rank 0's JSON line last) is ours."""
    import subprocess
    port = free_port()  # for the rendezvous
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, rank, world):
    """The launcher and the dealing of the work without a GPU: every rank joins the process
    group, the ranks agree on their number and on the trees each one takes, rank 0 prints the
    line.  Nothing is measured (`value` null, `dry_run` true)."""
    import torch
    import torch.distributed as dist
    from libsbn_amd import sharding
    if world > 1 or os.environ.get("MI_BENCH_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group(args.backend, rank=rank, world_size=world)
        ranks = dist.get_world_size()
        lo, hi = sharding.tree_shard(args.trees, rank, world)
        mine = torch.zeros(world, dtype=torch.int64)
        mine[rank] = hi - lo
        dist.all_reduce(mine)
        dist.barrier()
        dist.destroy_process_group()
        per_rank = [int(x) for x in mine]
    else:
        ranks, per_rank = 1, [args.trees]
    assert sum(per_rank) == args.trees
    if rank == 0:
        print(final_line({"metric": "tree log-likelihoods+gradients/sec (batched)", "value": None,
                          "unit": "trees/s", "n_gpus": world, "rccl_ranks": ranks,
                          "trees_per_rank": per_rank, "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True, "backend": args.backend,
                          "scaling": "strong"}), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--trees", type=int, default=1000,
                    help="trees per step over ALL GPUs (strong scaling)")
    ap.add_argument("--mode", choices=["gradient", "loglik"], default="gradient")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the other configurations (one-GPU runs only carry them)")
    ap.add_argument("--no-small-batch", action="store_true",
                    help="skip the 125-tree leg (profiling: its launches of the same kernel "
                         "would be averaged into the rocprofv3 --stats line of the timed one)")
    ap.add_argument("--workload", choices=["ds1", "swag"], default="ds1",
                    help="ds1: the headline (trees dealt to the GPUs); swag: BASELINE.json "
                         "configs[4], 20 states x 512 taxa x 50 000 patterns x 4 categories, "
                         "site patterns dealt to the GPUs, one all-reduce per step")
    ap.add_argument("--swag-trees", type=int, default=8)
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed headline steps (+ parity): no CPU baseline, other "
                         "configurations, small batch, host legs or reduced leg -- for "
                         "rocprofv3 runs, whose --stats line of the dominant kernel would "
                         "otherwise average those legs' launches in")
    ap.add_argument("--no-reduced", action="store_true",
                    help="skip the leg that ends in the fused reductions + one all-reduce")
    ap.add_argument("--no-host-legs", action="store_true",
                    help="skip the host-pointer / C++ adapter / Python mirror legs")
    ap.add_argument("--no-parity", action="store_true",
                    help="kernel experiments that skip work on purpose: no oracle check, the "
                         "line says \"parity\": \"skipped\" and the exit code is 3")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default=None,
                    help="process-group backend (default nccl; gloo with --dry-run, the CPU test "
                         "of the launcher, and only there)")
    ap.add_argument("--dry-run", action="store_true",
                    help="start the ranks, form the process group, deal the trees, print the "
                         "line -- no GPU work, value null (tests/test_bench_launcher.py)")
    args = ap.parse_args()
    if args.backend is None:
        args.backend = "gloo" if args.dry_run else "nccl"
    if args.headline_only:
        args.no_cpu_baseline = args.no_also = args.no_small_batch = True
        args.no_host_legs = args.no_reduced = True

    # ---- one rank per GPU.  Without a launcher around us (no WORLD_SIZE) and --gpus N > 1,
    # start the N ranks ourselves -- as a CHILD process, before anything here touches the GPU
    # (an initialised process must not exec or be replaced on this pool) -- and relay rank 0's
    # JSON line: `python bench.py --gpus 8` cannot silently measure one GPU.
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} "
                 "ranks; they must agree")
    if args.dry_run:
        return dry_run(args, rank, world)
    if args.backend != "nccl":
        sys.exit("--backend gloo is for --dry-run only: the engine has no CPU path")

    import torch
    import torch.distributed as dist
    import libsbn_amd as L
    from libsbn_amd import sharding

    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # everything runs on ONE explicit stream: torch's default stream has the null handle, which
    # the C ABI reads as "the engine's own stream" -- events recorded by torch would then sit
    # on another stream than the kernels they are meant to time
    compute = torch.cuda.Stream(dev)
    torch.cuda.set_stream(compute)
    # MI_BENCH_FORCE_DIST=1 runs the collective path even with one rank (smoke test of
    # the RCCL branch on a 1-GPU box)
    distributed = world > 1 or os.environ.get("MI_BENCH_FORCE_DIST") == "1"
    if distributed:
        if world > 1 and ("MASTER_ADDR" not in os.environ or "MASTER_PORT" not in os.environ):
            # (a free port picked here would be a DIFFERENT port in every rank --
            # the ranks of a launcher that sets WORLD_SIZE / RANK only would never meet;
            # spawn_ranks and torch.distributed.run both set the pair)
            sys.exit("bench.py: WORLD_SIZE > 1 needs MASTER_ADDR and MASTER_PORT from the "
                     "launcher (torch.distributed.run sets them; `python bench.py --gpus N` "
                     "without a launcher starts the ranks itself)")
        if "MASTER_ADDR" not in os.environ:  # one rank, forced collective path
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    rccl_ranks = dist.get_world_size() if distributed else 1
    assert rccl_ranks == world
    # Which physical GPU each rank drives: ordinal, PCI address and UUID
    # as torch reports them, gathered on rank 0 -- N ranks on fewer than N devices would make
    # every scaling figure meaningless, so that is an error, not a note.
    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(),
          "name": props.name,
          "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", -1) & 0xff,
                                     getattr(props, "pci_device_id", -1) & 0xff),
          "uuid": str(getattr(props, "uuid", ""))}
    rank_devices = [me]
    if distributed and world > 1:
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, me)
        ids = {(d["pci"], d["uuid"]) for d in rank_devices}
        assert len(ids) == world, f"{world} ranks on {len(ids)} distinct devices: {rank_devices}"

    if args.workload == "swag":
        out = swag_pattern_sharded(args, torch, dist, L, sharding, dev, rank, world, distributed)
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            emit(out)
        return

    grad = args.mode == "gradient"
    T_total = args.trees
    tips, w, pids_all, bls_all = ds1_workload(max(T_total, 1000) * (world if world > 1 else 1))
    params_all = np.ones((len(pids_all), 2))  # Weibull shape 1.0 | clock rate (never read)
    n, P = tips.shape
    N, K = 2 * n - 1, 4
    eng = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w,
                   device=local_rank)
    stream = torch.cuda.current_stream().cuda_stream

    def timed_region(step, steps, warmup, finish=None):
        """The contract's timing: `warmup` untimed steps, then EXACTLY `steps` steps bracketed
        by a barrier + synchronize on both sides; returns the MAX over ranks of the elapsed
        seconds.  `finish` (optional) completes whatever the last step left pending."""
        for _ in range(warmup):
            step()
        if finish:
            finish()
        eng.check_status(stream)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        eng.profile_begin(steps)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        if finish:
            finish()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kernel_ms = eng.profile_collect(steps)
        eng.check_status(stream)
        if distributed:
            tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        return elapsed, kernel_ms

    def run_config(T_local_max, lo, hi, steps, warmup, spread=False):
        """Times `steps` steps of this rank's trees [lo, hi); all ranks gather blocks of
        T_local_max trees (shorter blocks are padded).  spread: afterwards, and OUTSIDE the
        timed region, the same steps once more with a HIP event after every step (the
        per-step min / median / max of the line)."""
        T = hi - lo
        d_pid = torch.from_numpy(pids_all[lo:hi]).to(dev)
        d_bl = torch.from_numpy(bls_all[lo:hi]).to(dev)
        d_par = torch.from_numpy(params_all[lo:hi]).to(dev)
        # per-tree results [logL | site gradient | branch gradient (N)]: ONE buffer per rank
        # that the engine writes into directly (no packing copy before the collective); two
        # sets, so that the all-gather of step k overlaps the kernels of step k + 1
        sets = [sharding.ResultBlocks(T_local_max, N, extra=1, device=dev) for _ in range(2)]
        for s in sets:
            s.buffer.zero_()
        outs = [torch.empty((world, sets[0].buffer.numel()), dtype=torch.float64, device=dev)
                for _ in range(2)] if distributed else [None, None]
        state = {"k": 0, "pending": None, "gathered": None}
        eng.reserve(T, grad)
        # the one collective of the call -- every rank's per-tree results, tree order.
        # "stream" (default): enqueued on the calls' own stream right behind the kernels (an
        # all-gather of <= 54 KB per rank is latency, not bandwidth: there is nothing to
        # overlap; torch still runs a synchronous collective on RCCL's stream between two event
        # waits, but it issues both itself, back to back with the launches); "overlap": asynchronously
        # on RCCL's stream under the next step's kernels, which write the other buffer set
        # (round 2-3 form: two cross-stream dependencies per step, 18.5 us on one rank).
        overlap = os.environ.get("MI_BENCH_COLLECTIVE", "stream") == "overlap"

        def drain():
            if state["pending"] is not None:
                work, out = state["pending"]
                work.wait()
                state["gathered"] = out
                state["pending"] = None

        def step():
            i = state["k"] & 1
            state["k"] += 1
            blk = sets[i]
            if grad:
                eng.gradients_device(stream, T, d_pid.data_ptr(), d_bl.data_ptr(),
                                     d_par.data_ptr(), blk.log_likelihoods.data_ptr(),
                                     blk.branch_gradients.data_ptr(), blk.extras[0].data_ptr(),
                                     None)
            else:
                eng.log_likelihoods_device(stream, T, d_pid.data_ptr(), d_bl.data_ptr(),
                                           d_par.data_ptr(), blk.log_likelihoods.data_ptr())
            if distributed and overlap:
                out, work = sharding.all_gather_result_blocks(blk, out=outs[i], async_op=True)
                drain()
                state["pending"] = (work, out)
            elif distributed:
                state["gathered"], _ = sharding.all_gather_result_blocks(blk, out=outs[i])

        elapsed, kernel_ms = timed_region(step, steps, warmup, drain)
        step_ms = None
        if spread:
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
            marks[0].record()
            for k in range(steps):
                step()
                marks[k + 1].record()
            drain()
            torch.cuda.synchronize()
            step_ms = [marks[k].elapsed_time(marks[k + 1]) for k in range(steps)]
        last = sets[(state["k"] - 1) & 1]
        d_ll, d_g = last.log_likelihoods[:T], last.branch_gradients[:T]
        assert bool(torch.isfinite(d_ll).all()), "non-finite log-likelihoods"
        if grad:
            assert bool(torch.isfinite(d_g).all()), "non-finite gradients"
        if distributed:
            ll_all, _, g_all = sharding.gathered_views(state["gathered"], last)
            assert ll_all.shape == (world, T_local_max) and g_all.shape == (world, T_local_max, N)
            assert bool(torch.equal(ll_all[rank][:T], d_ll)), "gathered slice mismatch"
            if grad:
                assert bool(torch.equal(g_all[rank][:T], d_g)), "gathered gradient slice mismatch"
        return elapsed, kernel_ms, last, T, step_ms

    # ---- strong scaling (the headline): T_total trees over all ranks
    lo, hi = sharding.tree_shard(T_total, rank, world)
    T_max = sharding.tree_shard(T_total, 0, world)[1]
    elapsed, kernel_ms, last, T_local, step_ms = run_config(T_max, lo, hi, args.steps,
                                                            args.warmup, spread=True)
    ms_per_step = 1e3 * elapsed / args.steps
    value = T_total / (elapsed / args.steps)
    kname, evals, gevals = eng.last_call_info()
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")
    head_ll = last.log_likelihoods[:T_local].cpu().numpy()
    head_g = last.branch_gradients[:T_local].cpu().numpy()
    head_site = last.extras[0][:T_local].cpu().numpy()

    # ---- parity of the timed outputs: a sample of the last step against the CPU oracle
    parity_n, parity_err = 0, None
    if rank == 0 and not args.no_parity:
        O = oracle()
        parity_n = min(16, T_local)
        ospec = O.make_spec(n, P, "JC69", "weibull+4")
        sl = slice(lo, lo + parity_n)
        ll_dev = head_ll[:parity_n]
        if grad:
            og = O.unrooted_gradients(ospec, tips, w, pids_all[sl], bls_all[sl], params_all[sl],
                                      False, min(parity_n, usable_cores()))
            g_dev, s_dev = head_g[:parity_n], head_site[:parity_n]
            errs = [np.max(np.abs(ll_dev - og["log_likelihood"]) / np.abs(og["log_likelihood"])),
                    np.max(np.abs(g_dev - og["branch_lengths"])) / np.max(np.abs(og["branch_lengths"])),
                    np.max(np.abs(s_dev - og["site_model"]) / np.abs(og["site_model"]))]
        else:
            oll = O.unrooted_log_likelihoods(ospec, tips, w, pids_all[sl], bls_all[sl],
                                             params_all[sl], False, min(parity_n, usable_cores()))
            errs = [np.max(np.abs(ll_dev - oll) / np.abs(oll))]
        parity_err = float(max(errs))
        assert parity_err <= 1e-10, f"timed outputs differ from the oracle: {errs}"

    # ---- weak scaling beside it (1000 trees per GPU), N > 1 only
    weak = None
    if world > 1:
        wl, wh = rank * T_total, (rank + 1) * T_total
        w_elapsed = run_config(T_total, wl, wh, args.steps, args.warmup)[0]
        weak = {"value": world * T_total / (w_elapsed / args.steps), "unit": "trees/s",
                "ms_per_step": 1e3 * w_elapsed / args.steps, "trees_per_gpu": T_total}

    # ---- north_star's literal collective: "a single RCCL all-reduce ... for the ELBO/gradient
    # sum".  The same trees, but the step ends in the variational-inference reductions
    # (mi_engine_gradients_unrooted_reduced_device: sum of logL, of the site gradients, and
    # the branch gradients scatter-added by split index -- vip/burrito.py:143-153,
    # vip/branch_model.py:125-132) and ONE all-reduce of 2 + index_count doubles instead of
    # the all-gather of every per-tree vector.
    reduced = None
    if grad and not args.no_reduced:
        index_count = 4096
        node = np.arange(N, dtype=np.int64)
        bi_all = ((np.arange(len(pids_all), dtype=np.int64)[:, None] % 100) * 53
                  + node[None, :] * 7) % index_count  # a stand-in for the split indices
        bi_all[:, -2:] = -1  # the fixed node and the root are not parameters
        bi_all = bi_all.astype(np.int32)
        d_pid = torch.from_numpy(pids_all[lo:hi]).to(dev)
        d_bl = torch.from_numpy(bls_all[lo:hi]).to(dev)
        d_par = torch.from_numpy(params_all[lo:hi]).to(dev)
        d_bi = torch.from_numpy(bi_all[lo:hi]).to(dev)
        packed = torch.zeros(2 + index_count, dtype=torch.float64, device=dev)
        r_ll = torch.empty(T_local, dtype=torch.float64, device=dev)
        eng.reserve_reduced(T_local, index_count)

        def reduced_step():
            rc = eng._lib.mi_engine_gradients_unrooted_reduced_device(
                eng._h, stream, T_local, d_pid.data_ptr(), d_bl.data_ptr(), d_par.data_ptr(), 0,
                d_bi.data_ptr(), None, index_count, packed.data_ptr(),
                packed.data_ptr() + 16, r_ll.data_ptr())
            if rc:
                raise RuntimeError(L._capi.last_error())
            if distributed:
                dist.all_reduce(packed)  # on the calls' stream, in place

        r_elapsed, _ = timed_region(reduced_step, args.steps, args.warmup)
        # the reductions against the per-tree results of the headline step (themselves checked
        # against the oracle above), scatter-added on the host the way vip does
        mine = np.zeros(2 + index_count)
        mine[0], mine[1] = head_ll.sum(), head_site.sum()
        keep = bi_all[lo:hi] >= 0
        np.add.at(mine[2:], bi_all[lo:hi][keep], head_g[keep])
        ref = torch.from_numpy(mine).to(dev)
        if distributed:
            dist.all_reduce(ref)
        r_err = float((packed - ref).abs().max() / ref.abs().max())
        assert r_err <= 1e-12, f"reduced step differs from the scatter-added per-tree results: {r_err}"
        reduced = {"trees_per_s": T_total / (r_elapsed / args.steps),
                   "ms_per_step": 1e3 * r_elapsed / args.steps,
                   "all_reduce_doubles": 2 + index_count,
                   "collective": f"one all_reduce over {rccl_ranks} rank(s)" if distributed
                                 else "none (one GPU)",
                   "max_rel_err": r_err,
                   "note": "mi_engine_gradients_unrooted_reduced_device (stable sort of the "
                           "(tree, node) entries by index + one ordered sum per run, after "
                           "the gradient kernels) + ONE all-reduce of [sum logL | sum site "
                           "gradient | branch gradients scatter-added by a 4096-entry index]; "
                           "checked against the host scatter-add (np.add.at) of the headline "
                           "step's per-tree results"}

    # ---- small batch: what one of 8 GPUs sees under strong scaling (125 trees)
    small = None
    if rank == 0 and world == 1 and not args.no_small_batch:
        Ts = max(1, T_total // 8)
        s_elapsed, s_kernel, _, _, s_steps = run_config(Ts, 0, Ts, 50, 5, spread=True)
        small = {"trees": Ts, "ms_per_step": 1e3 * s_elapsed / 50,
                 "step_ms_device": {"min": float(np.min(s_steps)),
                                    "median": float(np.median(s_steps)),
                                    "max": float(np.max(s_steps))},
                 "kernel_ms": float(np.mean(s_kernel)),
                 "speedup_vs_full_batch": ms_per_step / (1e3 * s_elapsed / 50),
                 "note": "the step one GPU of 8 runs under strong scaling (eager launches, "
                         + ("with" if distributed else "without") + " the all-gather)"}
        try:  # the same step replayed from a hipGraph (no host launch cost in the loop)
            gstream = torch.cuda.Stream()
            d_pid = torch.from_numpy(pids_all[:Ts]).to(dev)
            d_bl = torch.from_numpy(bls_all[:Ts]).to(dev)
            d_par = torch.from_numpy(params_all[:Ts]).to(dev)
            blk = sharding.ResultBlocks(Ts, N, extra=1, device=dev)
            # every workspace the graphs below point at is reserved BEFORE the first capture: a
            # later reserve that grew a buffer would leave a captured graph replaying freed
            # device pointers (buffers only grow)
            eng.reserve(max(Ts, T_total) if grad else Ts, grad)
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=gstream):
                cs = torch.cuda.current_stream().cuda_stream
                if grad:
                    eng.gradients_device(cs, Ts, d_pid.data_ptr(), d_bl.data_ptr(),
                                         d_par.data_ptr(), blk.log_likelihoods.data_ptr(),
                                         blk.branch_gradients.data_ptr(),
                                         blk.extras[0].data_ptr(), None)
                else:
                    eng.log_likelihoods_device(cs, Ts, d_pid.data_ptr(), d_bl.data_ptr(),
                                               d_par.data_ptr(), blk.log_likelihoods.data_ptr())
            for _ in range(5):
                graph.replay()
            torch.cuda.synchronize()
            # 100 replays back to back between TWO events (what a rank does step after step);
            # three such passes, the median.  Until round 5 an event was recorded after every
            # replay inside this pass: each record is a marker packet the next graph launch
            # queues behind -- it added ~10 us to every step (0.1424 against 0.1326 ms on one
            # box, same binary).  The per-replay spread is a second pass, marked as such.
            passes = []
            for _ in range(3):
                g0, g1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
                g0.record()
                for k in range(100):
                    graph.replay()
                g1.record()
                torch.cuda.synchronize()
                passes.append(g0.elapsed_time(g1) / 100)
            small["graph_ms_per_step"] = float(np.median(passes))
            small["graph_passes_ms_per_step"] = [float(x) for x in passes]
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(101)]
            evs[0].record()
            for k in range(100):
                graph.replay()
                evs[k + 1].record()
            torch.cuda.synchronize()
            reps = [evs[k].elapsed_time(evs[k + 1]) for k in range(100)]
            small["graph_with_event_per_replay_ms_per_step"] = evs[0].elapsed_time(evs[100]) / 100
            small["graph_step_ms"] = {"min": float(np.min(reps)), "median": float(np.median(reps)),
                                      "max": float(np.max(reps))}
            small["graph_speedup_vs_full_batch"] = ms_per_step / small["graph_ms_per_step"]
            assert bool(torch.isfinite(blk.log_likelihoods).all())
            # like against like: the FULL batch replayed from a hipGraph the same way -- the
            # one-GPU step of a strong-scaling curve whose 8-GPU step is the figure above
            if grad:
                f_pid = torch.from_numpy(pids_all[:T_total]).to(dev)
                f_bl = torch.from_numpy(bls_all[:T_total]).to(dev)
                f_par = torch.from_numpy(params_all[:T_total]).to(dev)
                fblk = sharding.ResultBlocks(T_total, N, extra=1, device=dev)
                fgraph = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(fgraph, stream=gstream):
                    eng.gradients_device(torch.cuda.current_stream().cuda_stream, T_total,
                                         f_pid.data_ptr(), f_bl.data_ptr(), f_par.data_ptr(),
                                         fblk.log_likelihoods.data_ptr(),
                                         fblk.branch_gradients.data_ptr(),
                                         fblk.extras[0].data_ptr(), None)
                for _ in range(3):
                    fgraph.replay()
                torch.cuda.synchronize()
                fpasses = []
                for _ in range(3):
                    g0, g1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
                    g0.record()
                    for k in range(20):
                        fgraph.replay()
                    g1.record()
                    torch.cuda.synchronize()
                    fpasses.append(g0.elapsed_time(g1) / 20)
                small["graph_full_batch_ms_per_step"] = float(np.median(fpasses))
                small["graph_full_over_small"] = (small["graph_full_batch_ms_per_step"]
                                                  / small["graph_ms_per_step"])
        except Exception as exc:  # capture support varies; the eager figure stands
            small["graph_error"] = repr(exc)[:200]
        if grad and "graph_error" not in small:
            # ... and with the rank's trees dealt to TWO engines of this GPU, each on its own
            # stream, forked and joined inside ONE hipGraph (what mi_engine_create_sharded's
            # logical shards of one device do for the host-pointer calls): one half's set-up,
            # matrices and reduction run under the other half's walk, and each walk's thinning
            # last round is filled by the other's waves
            try:
                Ta = (Ts + 1) // 2
                halves = [(0, Ta), (Ta, Ts)]
                eng_b = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w,
                                 device=local_rank)
                engines = [eng, eng_b]
                blk2 = sharding.ResultBlocks(Ts, N, extra=1, device=dev)
                for e2, (a0, a1) in zip(engines, halves):
                    e2.reserve(a1 - a0, True)
                side = torch.cuda.Stream()
                graph3 = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(graph3, stream=gstream):
                    side.wait_stream(gstream)
                    for e2, (a0, a1), st in zip(engines, halves, (gstream, side)):
                        with torch.cuda.stream(st):
                            e2.gradients_device(
                                st.cuda_stream, a1 - a0, d_pid[a0:a1].data_ptr(),
                                d_bl[a0:a1].data_ptr(), d_par[a0:a1].data_ptr(),
                                blk2.log_likelihoods[a0:a1].data_ptr(),
                                blk2.branch_gradients[a0:a1].data_ptr(),
                                blk2.extras[0][a0:a1].data_ptr(), None)
                    gstream.wait_stream(side)
                for _ in range(5):
                    graph3.replay()
                torch.cuda.synchronize()
                e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
                e0.record()
                for _ in range(100):
                    graph3.replay()
                e1.record()
                torch.cuda.synchronize()
                small["graph_two_engines_ms_per_step"] = e0.elapsed_time(e1) / 100
                assert bool(torch.equal(blk2.buffer, blk.buffer)), "two engines: results differ"
                eng_b.close()
            except Exception as exc:
                small["graph_two_engines_error"] = repr(exc)[:200]
        if distributed and "graph_error" not in small:
            # ... and with the step's ONE collective inside the graph (RCCL kernels capture like
            # any other when they are enqueued on the capturing stream): what a rank of a
            # strong-scaled run replays per step -- kernels + all-gather, no host launch cost
            try:
                gout = torch.empty((world, blk.buffer.numel()), dtype=torch.float64, device=dev)
                graph2 = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(graph2, stream=gstream):
                    cs = torch.cuda.current_stream().cuda_stream
                    eng.gradients_device(cs, Ts, d_pid.data_ptr(), d_bl.data_ptr(),
                                         d_par.data_ptr(), blk.log_likelihoods.data_ptr(),
                                         blk.branch_gradients.data_ptr(),
                                         blk.extras[0].data_ptr(), None)
                    sharding.all_gather_result_blocks(blk, out=gout)
                for _ in range(5):
                    graph2.replay()
                torch.cuda.synchronize()
                e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
                e0.record()
                for _ in range(100):
                    graph2.replay()
                e1.record()
                torch.cuda.synchronize()
                small["graph_with_collective_ms_per_step"] = e0.elapsed_time(e1) / 100
                assert bool(torch.equal(gout[rank], blk.buffer))
            except Exception as exc:
                small["graph_with_collective_error"] = repr(exc)[:200]

    # ---- the Engine-shaped call (reference: Engine::Gradients, src/engine.cpp:78-92 -- host
    # tree collections in, host vectors out), one GPU: (a) the C ABI's host-pointer entry with
    # numpy buffers (pinned staging, one synchronisation per call), (b) the same through the
    # g++-built C++ adapter (libsbn_amd/csrc/host/engine.hpp: std::vector<PhyloGradient>, a
    # std::map per tree), (c) the Python mirror (one PhyloGradient object per tree, as pylibsbn
    # returns them).  PCIe and host staging included; never `value`.
    host_pointer = adapter = None
    if rank == 0 and world == 1 and grad and not args.no_host_legs:
        host_pointer = host_pointer_leg(eng, pids_all[:T_total], bls_all[:T_total],
                                        params_all[:T_total], head_ll, head_g, head_site,
                                        args.steps, args.warmup)
        adapter = adapter_leg(bls_all[:T_total], head_ll, head_g, head_site, args.steps,
                              args.warmup)

    out = None
    if rank == 0:
        b_ll, b_g = algorithmic_bytes(n, P, K)
        f_ll, f_g = algorithmic_flops(n, P, K)
        ent = traffic_entry(kname.split("<")[0])
        traffic = ent.get("hbm_bytes_per_launch") if ent else None
        out = {
            "metric": "tree log-likelihoods+gradients/sec (batched)" if grad
                      else "tree log-likelihoods/sec (batched)",
            "value": value, "unit": "trees/s", "n_gpus": world, "rccl_ranks": rccl_ranks,
            "trees_per_rank": [sharding.tree_shard(T_total, r, world)[1]
                               - sharding.tree_shard(T_total, r, world)[0] for r in range(world)],
            "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step,
            "step_ms_device": {"min": float(np.min(step_ms)), "median": float(np.median(step_ms)),
                               "max": float(np.max(step_ms)),
                               "kernel_min": float(np.min(kernel_ms)) if kernel_ms else None,
                               "kernel_median": float(np.median(kernel_ms)) if kernel_ms else None,
                               "kernel_max": float(np.max(kernel_ms)) if kernel_ms else None,
                               "note": "rank 0, a second pass of the same steps AFTER the timed "
                                       "region: HIP events on the calls' stream after every step "
                                       "(device time between consecutive steps' ends); kernel_* "
                                       "= the dominant kernel's launches of the timed region"},
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "reference DS1 alignment + DS1.100_topologies; synthetic branch lengths "
                    "Exp(mean 0.1), seed 43",
            "config": {"workload": f"DS1 27 taxa x {P} site patterns x {T_total} trees per step "
                                   f"over {world} GPU(s), JC69+weibull+4 (K=4), "
                                   + ("phylo_gradients: logL + branch + site gradients"
                                      if grad else "log_likelihoods"),
                       "trees_total": T_total, "trees_per_gpu": T_local, "taxa": n,
                       "patterns": P, "categories": K,
                       "parallelism": f"trees dealt to {world} GPUs in contiguous blocks, one "
                                      "all_gather per step" if distributed else "single GPU",
                       "detail": "1949 sites -> 934 patterns; 100 topologies x "
                                 f"{max(1, T_total // 100)} branch-length draws; read through the "
                                 "product's FASTA / Newick ingest"},
            "roofline": roofline(kname, k_ms, T_local, f_g if grad else f_ll,
                                 b_g if grad else b_ll, traffic,
                                 "profiles/traffic.json (PMC, 1000-tree launch), not this run",
                                 shape=(n, P, K)),
            "parity_checked": parity_n, "parity_max_rel_err": parity_err,
            "parity_note": "first trees of the last timed step vs the CPU oracle, tolerance 1e-10",
            "rank_devices": rank_devices,
        }
        if args.no_parity:
            out["parity"] = "skipped"
        if weak:
            out["weak"] = weak
        if reduced:
            out["reduced"] = reduced
        if distributed:
            out["gathered"] = {"trees_per_s": value, "ms_per_step": ms_per_step,
                               "collective": f"one all_gather over {rccl_ranks} rank(s), "
                                             + os.environ.get("MI_BENCH_COLLECTIVE", "stream")}
        if small:
            out["small_batch_ms"] = small.get("graph_ms_per_step", small["ms_per_step"])
            out["small_batch"] = small
        if host_pointer:
            out["host_pointer"] = host_pointer
            out["host_pointer_trees_per_s"] = host_pointer["trees_per_s"]
        if adapter:
            out["adapter"] = adapter
            out["adapter_trees_per_s"] = adapter.get("trees_per_s")
        if world == 1 and not args.no_also:
            try:
                out["also"] = also_workloads(torch, dev, L, 5)
            except Exception as exc:
                out["also_error"] = repr(exc)[:300]
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(tips, w, pids_all[:T_total], bls_all[:T_total],
                                               params_all[:T_total], args.mode)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(out)
    if args.no_parity:
        # (kernel experiments that skip work on purpose: the line says "parity": "skipped" and
        # the run does not count as a success)
        sys.exit(3)


if __name__ == "__main__":
    main()
