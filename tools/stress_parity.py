"""Randomised parity sweep of the GPU engine against the CPU oracle (not part of the test
suite: run on an MI355X box, optionally with MI_PHYLO_SUBST_GRADIENT=analytic).
Random taxa / pattern / category counts, models, rescaling, branch-length scales, gaps."""
import sys, os, itertools, collections
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import oracle_lib as O, libsbn_amd as L, tree_utils as TU
import test_gpu_parity as TG
RTOL = 1e-10
rng = np.random.default_rng(int(os.environ.get('STRESS_SEED', '2024')))
bad = 0; total = 0
seen = collections.Counter()  # which kernels / stores the trials exercised
for trial in range(int(os.environ.get('STRESS_TRIALS', '60'))):
    n = int(rng.choice([3, 4, 5, 6, 7, 9, 12, 17, 26, 27, 28, 29, 30, 31, 32, 33, 36, 45, 50, 64, 69, 80, 100, 130, 257]))
    P = int(rng.choice([1, 2, 3, 11, 12, 13, 16, 47, 48, 49, 64, 100, 257]))
    # (one and two categories -- the first-generation walk's engines -- and three / four -- the
    # third's -- weighted up: every generation x store appears a hundred times per 1 000 trials)
    K = int(rng.choice([1, 1, 2, 2, 3, 3, 4, 4, 4, 5, 6, 8, 11, 16, 23, 64]))
    subst = str(rng.choice(["JC69", "GTR"]))
    resc = bool(rng.integers(0, 2))
    # (mostly a few trees; now and then a batch large enough that 32-100-taxon trees take the
    # arena variant of the walk and small ones fill the machine)
    T = int(rng.choice([1, 2, 9, 9, 150])) if n <= 100 and P <= 100 and K <= 8 else int(rng.choice([1, 2, 9]))
    # STRESS_FOCUS=arena: batches of 150 trees of 32-100 taxa (the arena variants of the walk
    # kernels); =fused: small trees, three or four categories, 1-300 trees (the one-launch call,
    # half of the GTR trials asking for the branch-length gradient only)
    focus = os.environ.get("STRESS_FOCUS", "")
    if focus == "arena":
        n = int(rng.choice([32, 33, 36, 41, 45, 50, 59, 64, 69, 80, 100]))
        P = int(rng.choice([11, 13, 47, 49, 64, 100]))
        K = int(rng.choice([1, 2, 3, 4, 4, 8]))
        T = 150
    elif focus == "fused":
        # (round 6: one and two categories take the look-up walk and its one-launch call as well)
        n = int(rng.choice([3, 4, 5, 7, 9, 12, 17, 26, 27, 28, 30, 31, 32, 33]))
        K = int(rng.choice([1, 2, 3, 4, 4]))
        T = int(rng.choice([1, 2, 9, 64, 300]))
    site = "constant" if K == 1 else f"weibull+{K}"
    tips, w = TU.random_alignment(n, P, rng, gap_fraction=float(rng.choice([0.0, 0.05, 0.5])))
    pids, bls = TU.random_trees(n, T, rng, mean_bl=float(rng.choice([0.001, 0.1, 1.0])))
    if rng.integers(0, 3) == 0: pids[0] = TU.ladder_topology(n)
    as_partials = rng.integers(0, 5) == 0  # tips handed over as 0/1 partial vectors
    eng = L.Engine(L.PhyloModelSpecification(subst, site, "none"), tips, w, device=0,
                   use_tip_states=not as_partials)
    spec = O.make_spec(n, P, subst, site, "none", use_tip_states=0 if as_partials else 1)
    blocks = {}
    if subst == "GTR":
        r, f = TU.random_gtr_params(T, rng); blocks["GTR rates"] = r; blocks["frequencies"] = f
    if K > 1: blocks["Weibull shape"] = rng.uniform(0.2, 3.0, size=(T, 1))
    pr = TG._params(spec, T, **blocks)
    # (a GTR call that asks for the branch-length gradient only: one evaluation per tree)
    only_branch = subst == "GTR" and bool(rng.integers(0, 2)) and focus == "fused"
    g = eng.gradients(pids, bls, pr, resc, gradient_blocks=("branch_lengths",) if only_branch else None)
    kern = eng.last_call_info()[0]
    seen[eng.last_call_path()] += 1
    O.set_transition_mode(1)
    og = O.unrooted_gradients(spec, tips, w, pids, bls, pr, resc, 4)
    O.set_transition_mode(0)
    ok = True
    if not np.all(np.isfinite(og["log_likelihood"])):
        # unscaled evaluation underflowed in the oracle too (large trees): nothing to
        # compare beyond "the engine does not pretend to have a number"
        total += 1
        if np.all(np.isfinite([x.log_likelihood for x in g])):
            bad += 1
            print("MISMATCH (finite where the oracle underflows)", dict(n=n, P=P, K=K, resc=resc))
        continue
    for t in range(T):
        okl = abs(g[t].log_likelihood - og["log_likelihood"][t]) <= RTOL * abs(og["log_likelihood"][t]) + 1e-13
        if not okl: print("  ll", g[t].log_likelihood, og["log_likelihood"][t])
        # (a gradient that is zero up to rounding -- all-gap columns -- has no relative scale)
        scale = max(np.max(np.abs(og["branch_lengths"][t])), 1e-4)  # (cancellation noise of O(1) terms is ~1e-15)
        okb = np.max(np.abs(g[t].gradient["branch_lengths"] - og["branch_lengths"][t])) <= 1e-9 * scale
        if not okb: print("  bl", g[t].gradient["branch_lengths"], og["branch_lengths"][t])
        ok &= okl and okb
        if K > 1 and not only_branch:
            tol_site = 1e-4 if (subst == "GTR" and os.environ.get("MI_PHYLO_SUBST_GRADIENT")) else 1e-8
            okk = abs(g[t].gradient["site_model"][0] - og["site_model"][t]) <= tol_site * max(1.0, abs(og["site_model"][t]))
            if not okk: print("  site", g[t].gradient["site_model"][0], og["site_model"][t])
            ok &= okk
        if subst == "GTR" and not only_branch:
            a = g[t].gradient["substitution_model"]; f_ = og["substitution_model"][t]
            # (finite differences of logL with a 1e-6 step: rounding noise ~ 1e-16 |logL| / 2e-6
            # per ulp of difference in the summation order)
            okk = np.max(np.abs(a - f_) / np.maximum(np.abs(f_), 1.0)) <= \
                max(1e-4, 2e-9 * abs(og["log_likelihood"][t]))
            if not okk: print("  subst", a, f_)
            ok &= okk
    # the log-likelihood-only call (its own kernel and traversal order)
    ll = np.asarray(eng.log_likelihoods(pids, bls, pr, resc))
    seen[eng.last_call_path()] += 1
    okl = np.all(np.abs(ll - og["log_likelihood"]) <= RTOL * np.abs(og["log_likelihood"]) + 1e-13)
    if not okl: print("  logL call", ll, og["log_likelihood"])
    ok &= bool(okl)
    total += 1
    if not ok:
        bad += 1
        print("MISMATCH", dict(n=n, P=P, K=K, subst=subst, resc=resc, T=T, kern=kern))
print("trials", total, "bad", bad, "analytic" if os.environ.get("MI_PHYLO_SUBST_GRADIENT") else "fd")
print("kernels seen:")
for path, count in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("  %5d  %s" % (count, path))
