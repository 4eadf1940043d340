"""Randomised parity sweep of the 20-state engine against the s-generic CPU oracle (not part
of the test suite: run on an MI355X box).  Random taxa / pattern / category counts, random
reversible models and the built-in WAG table, unrooted and rooted calls, gaps, ladder and
balanced topologies, branch-length scales, sharded handles, both tile counts per wave."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np  # noqa: E402
import aa_utils as A  # noqa: E402
import libsbn_amd as L  # noqa: E402
import libsbn_amd.engine as E  # noqa: E402
import oracle_lib as O  # noqa: E402
import tree_utils as TU  # noqa: E402

rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "2025")))
trials = int(os.environ.get("STRESS_TRIALS", "60"))
bad = total = 0
import collections  # noqa: E402
seen = collections.Counter()  # which kernels the trials exercised
wag = E.wag_model()
for trial in range(trials):
    n = int(rng.choice([3, 4, 5, 6, 9, 16, 17, 31, 32, 33, 64, 100, 129, 200, 257, 300]))
    P = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 130]))
    K = int(rng.choice([1, 2, 3, 4, 5, 8]))
    site = "constant" if K == 1 else f"weibull+{K}"
    T = int(rng.choice([1, 2, 5]))
    rooted = bool(rng.integers(0, 4) == 0) and n <= 64
    model = wag if rng.integers(0, 2) else A.random_reversible_model(rng)
    shards = int(rng.choice([0, 0, 2, 3]))
    tips, w = A.random_aa_alignment(n, P, rng, gap_fraction=float(rng.choice([0.0, 0.05, 0.5])))
    pr = A.params_for(site, T, rng)
    O.set_reversible_model(*model)
    spec = A.oracle_spec(n, P, site)
    kw = dict(reversible_model=model)
    if shards and not rooted:
        kw.update(shard_devices=[0] * shards,
                  shard_mode="patterns" if (rng.integers(0, 2) and P >= shards) else "trees")
    elif shards:
        kw.update(shard_devices=[0] * shards)
    eng = L.Engine(L.PhyloModelSpecification("reversible", site, "strict"), tips, w, **kw)
    ok = True
    O.set_transition_mode(1)
    try:
        if not rooted:
            pids, bls = TU.random_trees(n, T, rng, mean_bl=float(rng.choice([0.001, 0.1, 1.0])))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                pids[0] = TU.ladder_topology(n)
            elif kind == 1:
                pids[0] = TU.balanced_topology(n)
            og = O.unrooted_gradients(spec, tips, w, pids, bls, pr, True, 4)
            g = eng.gradients(pids, bls, pr)
            seen[eng.last_call_path()] += 1
            ll = eng.log_likelihoods(pids, bls, pr)
            seen[eng.last_call_path()] += 1
            keys = ["branch_lengths"]
        else:
            trees = [TU.clocklike_rooted_tree(n, rng) for _ in range(T)]
            pids = np.stack([t[0] for t in trees])
            bls = np.stack([t[1] for t in trees])
            st = [O.time_tree_init(n, t[0], t[1], t[2]) for t in trees]
            h, bd, ra = (np.stack([s[i] for s in st]) for i in range(3))
            rates = np.full((T, 2 * n - 2), float(rng.choice([0.01, 0.5])))
            counts = np.ones(T, np.int32)
            og = O.rooted_gradients(spec, tips, w, pids, bls, pr, rates, counts, h, bd, ra, True, 4)
            g = eng.rooted_gradients(pids, bls, pr, rates, counts, h, bd, ra)
            seen[eng.last_call_path() + " rooted"] += 1
            ll = np.array([x.log_likelihood for x in g])
            keys = ["ratios_root_height", "clock_model"]
    finally:
        O.set_transition_mode(0)
    oll = og["log_likelihood"]
    if not np.all(np.abs(ll - oll) <= 1e-10 * np.abs(oll) + 1e-12):
        ok = False
        print("  logL", ll, oll)
    for t in range(T):
        if abs(g[t].log_likelihood - oll[t]) > 1e-10 * abs(oll[t]) + 1e-12:
            ok = False
        for k in keys:
            want = og[k][t] if k != "clock_model" else og[k][t][:1]
            got = g[t].gradient[k]
            scale = max(np.max(np.abs(want)), 1e-4)
            if np.max(np.abs(got - want)) > 1e-9 * scale:
                ok = False
                print("  ", k, np.max(np.abs(got - want)) / scale)
        if K > 1 and abs(g[t].gradient["site_model"][0] - og["site_model"][t]) > \
                1e-8 * max(1.0, abs(og["site_model"][t])):
            ok = False
            print("  site", g[t].gradient["site_model"][0], og["site_model"][t])
    total += 1
    if not ok:
        bad += 1
        print("MISMATCH", dict(n=n, P=P, K=K, T=T, rooted=rooted, shards=kw.get("shard_devices"),
                               mode=kw.get("shard_mode")))
    eng.close()
print("trials", total, "bad", bad)
print("kernels seen:")
for path, count in sorted(seen.items(), key=lambda kv: -kv[1]):
    print("  %5d  %s" % (count, path))
