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
figure (1000 trees per GPU) rides along in "weak".  On one GPU the line also carries, in
"also", the other BASELINE.json configurations (GTR+4G full / branch-only, fluA rooted,
log-likelihoods only, the 1949-pattern shape, the 20-state 512 x 50 000 case), a small-batch
(125 trees) step time, and a parity check of the timed outputs against the CPU oracle.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--trees T_total] [--mode gradient|loglik]

For N > 1 launch with torch.distributed.run (one rank per GPU).
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


def cpu_baseline(tips, w, pids, bls, params, mode, budget_s=12.0):
    """The CPU oracle (a port of the reference's algorithm, NOT BEAGLE itself: BEAGLE
    is not available in this image) timed on the host cores with the reference's
    tree-level threading model (one worker per core, FatBeagleParallelize)."""
    O = oracle()
    cores = usable_cores()
    spec = O.make_spec(tips.shape[0], tips.shape[1], "JC69", "weibull+4")
    fn = O.unrooted_gradients if mode == "gradient" else O.unrooted_log_likelihoods

    def take(count):  # the batch, cycled
        idx = np.arange(count) % len(pids)
        return pids[idx], bls[idx], params[idx]

    S = 4 * cores
    a, b, c = take(S)
    fn(spec, tips, w, a, b, c, False, cores)  # warm-up (page faults, thread start)
    t0 = time.perf_counter()
    fn(spec, tips, w, a, b, c, False, cores)
    dt = time.perf_counter() - t0
    rate = S / dt
    S2 = max(cores, int(rate * budget_s) // cores * cores)
    a, b, c = take(S2)
    t0 = time.perf_counter()
    fn(spec, tips, w, a, b, c, False, cores)
    dt = time.perf_counter() - t0
    S1 = max(1, int(rate / cores * 2.0))  # also on one core (about 2 s)
    a1, b1, c1 = take(S1)
    t0 = time.perf_counter()
    fn(spec, tips, w, a1, b1, c1, False, 1)
    one_core = S1 / (time.perf_counter() - t0)
    return {"value": S2 / dt, "unit": "trees/s", "cores": cores, "kind": "port",
            "value_on_one_core": one_core,
            "sample": f"{S2} trees (the same batch, cycled), {mode} semantics, "
                      f"{cores} OpenMP threads = usable cores (affinity mask capped by the "
                      f"cgroup CPU quota; host has {os.cpu_count()} logical CPUs), one tree "
                      f"per thread, one workspace per thread, {dt:.1f} s; "
                      "CPU oracle = BEAGLE-equivalent algorithm (cache-blocked over site "
                      "patterns) in plain C -O3 -march=native, not BEAGLE"}


def roofline(kname, k_ms, units, flops_per_unit, bytes_per_unit, traffic=None,
             traffic_source=None):
    """The dominant kernel against the roofline that binds it (FP64 matrix cores for the
    on-chip kernels) with the SURVEY 8(d) HBM streaming model beside it."""
    tflops = flops_per_unit * units / (k_ms * 1e-3) / 1e12
    gbps = bytes_per_unit * units / (k_ms * 1e-3) / 1e9
    return {"bound": "mfma", "achieved": tflops, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": tflops / FP64_PEAK_TFLOPS, "fp64_frac": tflops / FP64_PEAK_TFLOPS,
            "fp64_peak": FP64_PEAK_TFLOPS, "fp64_peak_source": FP64_PEAK_SOURCE,
            "traffic": traffic, "traffic_source": traffic_source,
            "kernel": kname, "kernel_ms": k_ms,
            "algorithmic_flops_per_tree": flops_per_unit,
            "hbm_model_GBps": gbps, "hbm_model_frac": gbps / HBM_PEAK_GBPS,
            "hbm_model_bytes_per_tree": bytes_per_unit,
            "note": "achieved = SURVEY 8(d) algorithmic flops x trees per launch / kernel time "
                    "(HIP events on the call's stream); hbm_model_* = the 8(d) PLV-streaming "
                    "byte model over the same time, kept for reference: the 4-state kernels "
                    "keep partial vectors in LDS, so that model does not bound them (frac > 1)"}


class Timed:
    """Times `steps` calls of fn() after `warmup` calls; kernel time from the engine's HIP
    events around its dominant kernel(s)."""

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


def also_workloads(torch, dev, L, steps):
    """The other BASELINE.json configurations, one GPU, short runs (rank 0 only)."""
    out = []
    stream = torch.cuda.current_stream().cuda_stream

    def dev_arrays(*arrays):
        return [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in arrays]

    # --- DS1 x 1000: GTR+weibull+4 (configs[2]), log-likelihoods, the 1949-pattern shape
    T = 1000
    tips, w, pids, bls = ds1_workload(T)
    n, P = tips.shape
    N, K = 2 * n - 1, 4
    rng = np.random.default_rng(45)
    gtr = np.hstack([rng.dirichlet(10 * np.ones(6), T), rng.dirichlet(10 * np.ones(4), T),
                     np.ones((T, 2))])
    d_pid, d_bl, d_gtr, d_jc = dev_arrays(pids, bls, gtr, np.ones((T, 2)))
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
    for label, s_ptr, u_ptr, bytes_, flops in (
            ("full phylo_gradients (reference semantics: 16 finite-difference passes + "
             "perturbed-model site pass)", site.data_ptr(), sub.data_ptr(),
             2 * b_g + 16 * b_ll, 2 * f_g + 16 * f_ll),
            ("logL + branch-length gradient only (configs[2] as worded)", None, None, b_g, f_g)):
        ms, k_ms = tm.run(lambda: eng.gradients_device(
            stream, T, d_pid.data_ptr(), d_bl.data_ptr(), d_gtr.data_ptr(), ll.data_ptr(),
            g.data_ptr(), s_ptr, u_ptr), steps, 1)
        kname = eng.last_call_info()[0]
        out.append({"workload": f"DS1 27 taxa x {P} patterns x {T} trees, GTR+weibull+4, " + label,
                    "trees_per_s": T / (ms * 1e-3), "ms_per_step": ms, "kernel": kname,
                    "kernel_ms": k_ms,
                    "kernel_note": "HIP events around the main gradient pass only; the call also "
                                   "runs the finite-difference / site passes",
                    "roofline": roofline(kname, ms, T, flops, bytes_)})
    assert bool(torch.isfinite(sub).all())
    eng.close()
    eng = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w,
                   device=dev.index)
    eng.reserve(T, False)
    tm = Timed(torch, eng, stream)
    ms, k_ms = tm.run(lambda: eng.log_likelihoods_device(
        stream, T, d_pid.data_ptr(), d_bl.data_ptr(), d_jc.data_ptr(), ll.data_ptr()), steps, 1)
    kname = eng.last_call_info()[0]
    out.append({"workload": f"DS1 27 taxa x {P} patterns x {T} trees, JC69+weibull+4, log_likelihoods",
                "trees_per_s": T / (ms * 1e-3), "ms_per_step": ms, "kernel": kname,
                "kernel_ms": k_ms, "roofline": roofline(kname, k_ms, T, f_ll, b_ll)})
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
    out.append({"workload": f"S-DS1 27 taxa x 1949 patterns (synthetic, evolved under JC) x {T} "
                            "trees, JC69+weibull+4, phylo_gradients",
                "trees_per_s": T / (ms * 1e-3), "ms_per_step": ms, "kernel": kname,
                "kernel_ms": k_ms, "roofline": roofline(kname, k_ms, T, f2_g, b2_g)})
    eng.close()

    # --- fluA rooted, strict clock (configs[3])
    for Tf in (1, 1000):
        tips, w, pids, bls, rates, counts, hs, bs, rs = flua_workload(Tf)
        n, P = tips.shape
        N = 2 * n - 1
        eng = L.Engine(L.PhyloModelSpecification("JC69", "constant", "strict"), tips, w,
                       device=dev.index)
        d = dev_arrays(pids, bls, np.ones((Tf, 1)), rates, counts, hs, bs, rs)
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
        bf_ll, bf_g = algorithmic_bytes(n, P, 1)
        ff_ll, ff_g = algorithmic_flops(n, P, 1)
        assert bool(torch.isfinite(gr).all())
        out.append({"workload": f"fluA rooted {n} taxa x {P} patterns x {Tf} tree(s), JC69, strict "
                                "clock, node-height-ratio + clock gradient",
                    "trees_per_s": Tf / (ms * 1e-3), "ms_per_step": ms, "kernel": kname,
                    "kernel_ms": k_ms, "logL0": float(ll[0]),
                    "roofline": roofline(kname, k_ms, Tf, ff_g, bf_g)})
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
    for Tw in (1, 8):
        pids = np.stack([random_unrooted_topology(n, rng) for _ in range(Tw)])
        bls = rng.exponential(0.1, size=(Tw, 2 * n - 2))
        bls[:, -1] = 0
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
            ms, k_ms = tm.run(fn, 3 if Tw == 1 else 2, 1)
            assert bool(torch.isfinite(ll).all())
            b_model = ((10 * n - 14) if grad else 2 * (n - 1)) * plv + (12 if grad else 4) * n * P
            f_model = algorithmic_flops(n, P, K, 20)[1 if grad else 0]
            # what the kernels really have to do: a tip child's product is a table look-up,
            # so only the n-2 internal edges cost 20x20 products (1 in the post-order, 2-3 more
            # in the pre-order); the arena moves n-2 vectors out and back (+ stacked ones)
            # the pre-order pass forms P L itself only for the child the post-order pass did
            # NOT take from registers, i.e. once per node with two internal children
            mfma_flops = ((3 * (n - 2) + two_internal_children(pids[0], n)) if grad
                          else (n - 2)) * K * P * 800.0
            r = {"bound": "hbm", "achieved": b_model * Tw / (k_ms * 1e-3) / 1e9,
                 "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                 "frac": b_model * Tw / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                 "traffic": (49.7e9 if grad else 7.9e9),
                 "traffic_source": "profiles/r02_aa_pmc_{fetch,write}.csv, profiles/traffic.json "
                                   "((2 FETCH_SIZE + WRITE_SIZE) x 1024, separate passes), per "
                                   "tree; not this run",
                 "kernel": "aa_post_wg_kernel + aa_root_kernel" + (" + aa_pre_wg_kernel" if grad else ""),
                 "kernel_ms": k_ms,
                 "hbm_model_bytes_per_tree": b_model,
                 "survey_flops_TFLOPs": f_model * Tw / (k_ms * 1e-3) / 1e12,
                 "mfma_TFLOPs": mfma_flops * Tw / (k_ms * 1e-3) / 1e12,
                 "mfma_frac_of_measured_peak": mfma_flops * Tw / (k_ms * 1e-3) / 1e12
                                               / FP64_PEAK_TFLOPS,
                 "note": "achieved = SURVEY 8(d) PLV-streaming bytes (B_LL / B_G) over the time "
                         "of the walk kernels; the kernels keep the chained child in registers "
                         "and tips compact, so the real traffic is `traffic`; mfma_TFLOPs counts "
                         "only the 20x20 products the kernels execute (tip products are table "
                         "look-ups)"}
            out.append({"workload": f"S-WAG 20 states, {n} taxa x {P} patterns x {K} categories "
                                    f"x {Tw} tree(s), " + ("phylo_gradients" if grad else
                                                           "log_likelihoods"),
                        "trees_per_s": Tw / (ms * 1e-3), "ms_per_step": ms,
                        "kernel": eng.last_call_info()[0], "kernel_ms": k_ms, "logL0": float(ll[0]),
                        "roofline": r})
    eng.close()
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
    if rank != 0:
        return None
    plv = K * P * 20 * 8
    b_model = ((10 * n - 14) if grad else 2 * (n - 1)) * plv + (12 if grad else 4) * n * P
    ms = 1e3 * elapsed / steps
    return {
        "metric": "tree log-likelihoods+gradients/sec (batched)" if grad
                  else "tree log-likelihoods/sec (batched)",
        "value": T / (elapsed / steps), "unit": "trees/s", "n_gpus": world, "steps": steps,
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
        "roofline": {"bound": "hbm", "achieved": b_model * T / world / (k_ms * 1e-3) / 1e9,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": b_model * T / world / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "traffic": None, "kernel": eng.last_call_info()[0], "kernel_ms": k_ms,
                     "note": "SURVEY 8(d) PLV-streaming bytes of this rank's pattern block over "
                             "the time of its walk kernels"},
        "logL0": float(blk.log_likelihoods[0]),
    }


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
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import libsbn_amd as L
    from libsbn_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # MI_BENCH_FORCE_DIST=1 runs the collective path even with one rank (smoke test of
    # the RCCL branch on a 1-GPU box)
    distributed = world > 1 or os.environ.get("MI_BENCH_FORCE_DIST") == "1"
    if distributed:
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if args.workload == "swag":
        out = swag_pattern_sharded(args, torch, dist, L, sharding, dev, rank, world, distributed)
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            import ctypes
            ctypes.CDLL(None).fflush(None)
            sys.stdout.flush()
            print(json.dumps(out), flush=True)
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

    def run_config(T_local_max, lo, hi, steps, warmup):
        """Times `steps` steps of this rank's trees [lo, hi); all ranks gather blocks of
        T_local_max trees (shorter blocks are padded)."""
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
            if distributed:
                # the one collective of the call: every rank's per-tree results, tree order,
                # enqueued asynchronously (RCCL's stream) so that it overlaps the next step's
                # kernels, which write the other buffer set; it is waited for one step later
                # (and, for the last step, before the timed region ends).
                out, work = sharding.all_gather_result_blocks(blk, out=outs[i], async_op=True)
                drain()
                state["pending"] = (work, out)

        for _ in range(warmup):
            step()
        drain()
        eng.check_status(stream)
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        eng.profile_begin(steps)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        drain()
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
        return elapsed, kernel_ms, last, T

    # ---- strong scaling (the headline): T_total trees over all ranks
    lo, hi = sharding.tree_shard(T_total, rank, world)
    T_max = sharding.tree_shard(T_total, 0, world)[1]
    elapsed, kernel_ms, last, T_local = run_config(T_max, lo, hi, args.steps, args.warmup)
    ms_per_step = 1e3 * elapsed / args.steps
    value = T_total / (elapsed / args.steps)
    kname, evals, gevals = eng.last_call_info()
    k_ms = float(np.mean(kernel_ms)) if kernel_ms else float("nan")

    # ---- parity of the timed outputs: a sample of the last step against the CPU oracle
    parity_n, parity_err = 0, None
    if rank == 0:
        O = oracle()
        parity_n = min(16, T_local)
        ospec = O.make_spec(n, P, "JC69", "weibull+4")
        sl = slice(lo, lo + parity_n)
        ll_dev = last.log_likelihoods[:parity_n].cpu().numpy()
        if grad:
            og = O.unrooted_gradients(ospec, tips, w, pids_all[sl], bls_all[sl], params_all[sl],
                                      False, min(parity_n, usable_cores()))
            g_dev = last.branch_gradients[:parity_n].cpu().numpy()
            s_dev = last.extras[0][:parity_n].cpu().numpy()
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
        w_elapsed, _, _, _ = run_config(T_total, wl, wh, args.steps, args.warmup)
        weak = {"value": world * T_total / (w_elapsed / args.steps), "unit": "trees/s",
                "ms_per_step": 1e3 * w_elapsed / args.steps, "trees_per_gpu": T_total}

    # ---- small batch: what one of 8 GPUs sees under strong scaling (125 trees)
    small = None
    if rank == 0 and world == 1 and not args.no_small_batch:
        Ts = max(1, T_total // 8)
        s_elapsed, s_kernel, _, _ = run_config(Ts, 0, Ts, 50, 5)
        small = {"trees": Ts, "ms_per_step": 1e3 * s_elapsed / 50,
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
            eng.reserve(Ts, grad)
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
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            small["graph_ms_per_step"] = e0.elapsed_time(e1) / 100
            small["graph_speedup_vs_full_batch"] = ms_per_step / small["graph_ms_per_step"]
            assert bool(torch.isfinite(blk.log_likelihoods).all())
        except Exception as exc:  # capture support varies; the eager figure stands
            small["graph_error"] = repr(exc)[:200]

    out = None
    if rank == 0:
        b_ll, b_g = algorithmic_bytes(n, P, K)
        f_ll, f_g = algorithmic_flops(n, P, K)
        traffic = None
        tpath = os.path.join(REPO, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(kname, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "tree log-likelihoods+gradients/sec (batched)" if grad
                      else "tree log-likelihoods/sec (batched)",
            "value": value, "unit": "trees/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "reference DS1 alignment + DS1.100_topologies (read through the product's "
                    "FASTA / Newick ingest); synthetic branch lengths Exp(mean 0.1), seed 43",
            "config": {"workload": f"DS1 27 taxa x {P} site patterns (1949 sites) x {T_total} "
                                   f"trees per step over {world} GPU(s) (100 topologies x "
                                   f"{max(1, T_total // 100)} draws), JC69+weibull+4 (K=4), "
                                   + ("phylo_gradients: logL + branch + site gradients"
                                      if grad else "log_likelihoods"),
                       "trees_total": T_total, "trees_per_gpu": T_local, "taxa": n,
                       "patterns": P, "categories": K,
                       "parallelism": f"trees dealt to {world} GPUs in contiguous blocks, one "
                                      "all_gather per step" if distributed else "single GPU"},
            "roofline": roofline(kname, k_ms, T_local, f_g if grad else f_ll,
                                 b_g if grad else b_ll, traffic,
                                 "profiles/traffic.json (PMC, 1000-tree launch), not this run"),
            "parity_checked": parity_n, "parity_max_rel_err": parity_err,
            "parity_note": "first trees of the last timed step vs the CPU oracle, tolerance 1e-10",
        }
        if weak:
            out["weak"] = weak
        if small:
            out["small_batch_ms"] = small.get("graph_ms_per_step", small["ms_per_step"])
            out["small_batch"] = small
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
        # RCCL prints a version banner through C stdio, which is flushed at exit when
        # stdout is a file: flush it now so that the JSON line is the last line.
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)  # the ONE JSON line, last thing on stdout


if __name__ == "__main__":
    main()
