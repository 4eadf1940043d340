"""Randomised parity sweep of the ROOTED path (rate-scaled branch lengths, log-det Jacobian,
node-height-ratio / clock gradients) against the CPU oracle; tree sizes on both sides of
every tree-setup variant (N <= 64, 128, 192, 256, LDS).  Run on an MI355X box."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import oracle_lib as O, libsbn_amd as L, tree_utils as TU
import test_gpu_parity as TG

rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "99")))
import collections
bad = 0
seen = collections.Counter()  # which kernels / stores the trials exercised
trials = int(os.environ.get("STRESS_TRIALS", "30"))
for trial in range(trials):
    # (STRESS_N="257,258,300": sizes on both sides of the 257-taxon limit of the register
    # recurrences in finalize)
    sizes = [int(x) for x in os.environ["STRESS_N"].split(",")] if os.environ.get("STRESS_N") else \
        [3, 4, 5, 8, 13, 31, 32, 33, 40, 64, 65, 70, 96, 97, 128, 129, 140, 257, 258]
    n = int(rng.choice(sizes))
    P = int(rng.choice([5, 12, 33, 64, 100]))
    K = int(rng.choice([1, 2, 4]))
    subst = str(rng.choice(["JC69", "GTR"]))
    site = "constant" if K == 1 else f"weibull+{K}"
    T = int(rng.choice([1, 3]))
    N = 2 * n - 1
    tips, w = TU.random_alignment(n, P, rng, gap_fraction=0.02)
    pids, bls, hs, bds, ras = [], [], [], [], []
    for _ in range(T):
        pid, bl, dates = TU.clocklike_rooted_tree(n, rng)
        h, bd, ra = O.time_tree_init(n, pid, bl, dates)
        pids.append(pid); bls.append(bl); hs.append(h); bds.append(bd); ras.append(ra)
    pids, bls, hs, bds, ras = map(np.stack, (pids, bls, hs, bds, ras))
    rates = rng.uniform(0.01, 0.1, size=(T, N - 1))
    rcounts = [int(rng.choice([1, N - 1])) for _ in range(T)]
    for t in range(T):
        if rcounts[t] == 1:
            rates[t] = rates[t, 0]
    eng = L.Engine(L.PhyloModelSpecification(subst, site, "strict"), tips, w, device=0)
    spec = O.make_spec(n, P, subst, site, "strict")
    blocks = {}
    if subst == "GTR":
        r, f = TU.random_gtr_params(T, rng); blocks["GTR rates"] = r; blocks["frequencies"] = f
    if K > 1:
        blocks["Weibull shape"] = rng.uniform(0.3, 2.0, size=(T, 1))
    pr = TG._params(spec, T, **blocks)
    resc = n > 100
    ll = eng.rooted_log_likelihoods(pids, bls, pr, rates, hs, bds, resc)
    seen[eng.last_call_path()] += 1
    oll = O.rooted_log_likelihoods(spec, tips, w, pids, bls, pr, rates, hs, bds, rescaling=resc)
    g = eng.rooted_gradients(pids, bls, pr, rates, rcounts, hs, bds, ras, resc)
    seen[eng.last_call_path()] += 1
    O.set_transition_mode(1)
    og = O.rooted_gradients(spec, tips, w, pids, bls, pr, rates, rcounts, hs, bds, ras, resc)
    O.set_transition_mode(0)
    ok = bool(np.all(np.abs(ll - oll) <= 1e-10 * np.abs(oll)))
    if not ok: print("  ll", ll, oll)
    for t in range(T):
        o1 = TG._close(g[t].gradient["ratios_root_height"], og["ratios_root_height"][t], 1e-9)
        if not o1: print("  ratios", np.max(np.abs(g[t].gradient["ratios_root_height"] - og["ratios_root_height"][t])), np.max(np.abs(og["ratios_root_height"][t])))
        oc = og["clock_model"][t, :1] if rcounts[t] == 1 else og["clock_model"][t]
        o2 = TG._close(g[t].gradient["clock_model"], oc, 1e-9)
        if not o2: print("  clock", np.max(np.abs(g[t].gradient["clock_model"] - oc)), np.max(np.abs(oc)))
        ok &= o1 and o2
        if K > 1:
            # (analytic mode evaluates the site gradient at the unperturbed model: the
            # reference's finite-difference pass leaves it perturbed by 1e-6)
            tol = 1e-4 if (subst == "GTR" and os.environ.get("MI_PHYLO_SUBST_GRADIENT")) else 1e-8
            ok &= abs(g[t].gradient["site_model"][0] - og["site_model"][t]) <= tol * max(1.0, abs(og["site_model"][t]))
        if subst == "GTR":
            a, f_ = g[t].gradient["substitution_model"], og["substitution_model"][t]
            ok &= np.max(np.abs(a - f_) / np.maximum(np.abs(f_), 1.0)) <= 1e-4
    if not ok:
        bad += 1
        print("MISMATCH", dict(n=n, P=P, K=K, subst=subst, T=T, resc=resc, kern=eng.last_call_info()[0]))
print("rooted trials", trials, "bad", bad)
print("kernels seen:")
for path, count in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("  %5d  %s" % (count, path))
