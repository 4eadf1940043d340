"""20-state path (BASELINE.json configs[4], kernels_aa.hip) against the s-generic CPU oracle.

The reference has no 20-state code (substitution_model.cpp:6-15, site_pattern.cpp:16-46 are
DNA-only), so parity here is GPU vs the oracle's s-generic loops -- the SAME loops
tests/test_oracle_kats.py shows to reproduce every known-answer value of the reference at
s = 4 (orc_set_generic_states) -- plus size-independent properties at larger shapes.
Tolerances: 1e-10 relative vs the FP64 oracle, 1e-12 vs its 80-bit build.
"""
import numpy as np
import pytest

import aa_utils as A
import oracle_lib as O
import tree_utils as TU

pytestmark = pytest.mark.gpu


def _engine(tips, w, site, model=None, **kw):
    import libsbn_amd as L
    return L.Engine(L.PhyloModelSpecification("WAG" if model is None else "reversible", site,
                                              "strict"), tips, w, reversible_model=model, **kw)


def _rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def _wag():
    import libsbn_amd.engine as E
    return E.wag_model()


@pytest.mark.parametrize("n,P,site", [
    (3, 1, "constant"), (4, 15, "weibull+4"), (5, 16, "constant"), (8, 33, "weibull+4"),
    (17, 100, "weibull+4"), (33, 47, "weibull+2"), (64, 40, "weibull+4"), (27, 200, "weibull+8"),
])
def test_unrooted_matches_oracle(n, P, site):
    rng = np.random.default_rng(1000 + n * 7 + P)
    tips, w = A.random_aa_alignment(n, P, rng)
    T = 3
    pids, bls = TU.random_trees(n, T, rng)
    pr = A.params_for(site, T, rng)
    ex, fr = _wag()
    O.set_reversible_model(ex, fr)
    spec = A.oracle_spec(n, P, site)
    O.set_transition_mode(1)
    try:
        oll = O.unrooted_log_likelihoods(spec, tips, w, pids, bls, pr, True, 4)
        og = O.unrooted_gradients(spec, tips, w, pids, bls, pr, True, 4)
    finally:
        O.set_transition_mode(0)
    eng = _engine(tips, w, site)
    ll = eng.log_likelihoods(pids, bls, pr)
    g = eng.gradients(pids, bls, pr)
    assert eng.last_call_info()[0].startswith("aa_pre")
    assert _rel(ll, oll) <= 1e-10
    assert _rel([x.log_likelihood for x in g], oll) <= 1e-10
    gb = np.stack([x.gradient["branch_lengths"] for x in g])
    assert gb.shape == (T, 2 * n - 1) and np.all(gb[:, -2:] == 0.0)
    assert _rel(gb, og["branch_lengths"]) <= 1e-10
    if site != "constant":
        assert _rel([x.gradient["site_model"][0] for x in g], og["site_model"]) <= 1e-9
    assert "substitution_model" not in g[0].gradient


def test_beagle_transition_form_agrees_too():
    """The oracle's default (BEAGLE's V exp(L t) V^-1) against the engine's expm1 form: equal
    to 1e-10 at ordinary branch lengths."""
    rng = np.random.default_rng(5)
    n, P, site = 12, 60, "weibull+4"
    tips, w = A.random_aa_alignment(n, P, rng)
    pids, bls = TU.random_trees(n, 2, rng)
    pr = A.params_for(site, 2, rng)
    ex, fr = _wag()
    O.set_reversible_model(ex, fr)
    og = O.unrooted_gradients(A.oracle_spec(n, P, site), tips, w, pids, bls, pr, False, 2)
    g = _engine(tips, w, site).gradients(pids, bls, pr)
    assert _rel([x.log_likelihood for x in g], og["log_likelihood"]) <= 1e-10
    assert _rel(np.stack([x.gradient["branch_lengths"] for x in g]), og["branch_lengths"]) <= 1e-9


def test_extended_precision():
    rng = np.random.default_rng(77)
    n, P, site = 20, 50, "weibull+4"
    tips, w = A.random_aa_alignment(n, P, rng)
    pids, bls = TU.random_trees(n, 2, rng)
    pr = A.params_for(site, 2, rng)
    ex, fr = A.random_reversible_model(rng)
    O.set_reversible_model(ex, fr)
    O.select("ld")
    O.set_transition_mode(1)
    try:
        og = O.unrooted_gradients(A.oracle_spec(n, P, site), tips, w, pids, bls, pr, True, 2)
    finally:
        O.set_transition_mode(0)
        O.select("f64")
    g = _engine(tips, w, site, model=(ex, fr)).gradients(pids, bls, pr)
    assert _rel([x.log_likelihood for x in g], og["log_likelihood"]) <= 1e-12
    assert _rel(np.stack([x.gradient["branch_lengths"] for x in g]), og["branch_lengths"]) <= 1e-11
    assert _rel([x.gradient["site_model"][0] for x in g], og["site_model"]) <= 1e-10


def test_rooted_strict_clock_matches_oracle():
    rng = np.random.default_rng(9)
    n, P, site = 14, 70, "weibull+4"
    tips, w = A.random_aa_alignment(n, P, rng)
    T = 2
    trees = [TU.clocklike_rooted_tree(n, rng) for _ in range(T)]
    pids = np.stack([t[0] for t in trees])
    bls = np.stack([t[1] for t in trees])
    state = [O.time_tree_init(n, t[0], t[1], t[2]) for t in trees]
    h = np.stack([s[0] for s in state])
    bd = np.stack([s[1] for s in state])
    ra = np.stack([s[2] for s in state])
    rates = np.full((T, 2 * n - 2), 0.7)
    counts = np.ones(T, np.int32)
    pr = A.params_for(site, T, rng)
    ex, fr = _wag()
    O.set_reversible_model(ex, fr)
    spec = A.oracle_spec(n, P, site)
    O.set_transition_mode(1)
    try:
        oll = O.rooted_log_likelihoods(spec, tips, w, pids, bls, pr, rates, h, bd, True, True, 2)
        og = O.rooted_gradients(spec, tips, w, pids, bls, pr, rates, counts, h, bd, ra, True, 2)
    finally:
        O.set_transition_mode(0)
    eng = _engine(tips, w, site)
    ll = eng.rooted_log_likelihoods(pids, bls, pr, rates, h, bd)
    g = eng.rooted_gradients(pids, bls, pr, rates, counts, h, bd, ra)
    assert _rel(ll, oll) <= 1e-10
    assert _rel(np.stack([x.gradient["ratios_root_height"] for x in g]),
                og["ratios_root_height"]) <= 1e-9
    assert _rel([x.gradient["clock_model"][0] for x in g], og["clock_model"][:, 0]) <= 1e-9


def test_ladder_and_balanced_topologies_and_gaps_only_column():
    rng = np.random.default_rng(21)
    n, P, site = 40, 37, "weibull+4"
    tips, w = A.random_aa_alignment(n, P, rng, gap_fraction=0.2)
    tips[:, 3] = 20  # a column of gaps only: site likelihood 1
    pids = np.stack([TU.ladder_topology(n), TU.balanced_topology(n)])
    bls = rng.exponential(0.1, size=(2, 2 * n - 2))
    pr = A.params_for(site, 2, rng)
    ex, fr = _wag()
    O.set_reversible_model(ex, fr)
    O.set_transition_mode(1)
    try:
        og = O.unrooted_gradients(A.oracle_spec(n, P, site), tips, w, pids, bls, pr, True, 2)
    finally:
        O.set_transition_mode(0)
    eng = _engine(tips, w, site)
    g = eng.gradients(pids, bls, pr)
    ll = eng.log_likelihoods(pids, bls, pr)
    assert _rel(ll, og["log_likelihood"]) <= 1e-10
    assert _rel(np.stack([x.gradient["branch_lengths"] for x in g]), og["branch_lengths"]) <= 1e-10


def test_deep_tree_needs_rescaling_and_stays_finite():
    """512 taxa: unscaled site likelihoods (~20^-512) underflow FP64; the engine rescales by
    exact powers of two at every node.  Checked against the oracle (log rescaling)."""
    rng = np.random.default_rng(3)
    n, P, site = 512, 24, "weibull+4"
    tips, w = A.random_aa_alignment(n, P, rng, gap_fraction=0.02)
    pids, bls = TU.random_trees(n, 1, rng)
    pr = A.params_for(site, 1, rng)
    ex, fr = _wag()
    O.set_reversible_model(ex, fr)
    O.set_transition_mode(1)
    try:
        og = O.unrooted_gradients(A.oracle_spec(n, P, site), tips, w, pids, bls, pr, True, 4)
    finally:
        O.set_transition_mode(0)
    eng = _engine(tips, w, site)
    g = eng.gradients(pids, bls, pr)
    ll = eng.log_likelihoods(pids, bls, pr)
    assert np.isfinite(ll).all() and ll[0] < -1e4
    assert _rel(ll, og["log_likelihood"]) <= 1e-10
    assert _rel(g[0].gradient["branch_lengths"], og["branch_lengths"][0]) <= 1e-9


def test_properties_at_size():
    """Size-independent properties at a shape the oracle is not run on (128 taxa x 5 000
    patterns x 4 categories): pattern-block additivity (what pattern sharding over GPUs
    relies on), the directional derivative of logL along a random branch-length direction,
    run-to-run bitwise reproducibility, tree-order independence."""
    rng = np.random.default_rng(123)
    n, P, site = 128, 5000, "weibull+4"
    tips, w = A.random_aa_alignment(n, P, rng)
    T = 2
    pids, bls = TU.random_trees(n, T, rng)
    pr = A.params_for(site, T, rng)
    eng = _engine(tips, w, site)
    g = eng.gradients(pids, bls, pr)
    ll = np.array([x.log_likelihood for x in g])
    gb = np.stack([x.gradient["branch_lengths"] for x in g])
    gs = np.array([x.gradient["site_model"][0] for x in g])
    # reproducible, and independent of the position in the batch
    g2 = eng.gradients(pids[::-1].copy(), bls[::-1].copy(), pr[::-1].copy())
    assert np.array_equal(np.stack([x.gradient["branch_lengths"] for x in g2])[::-1], gb)
    assert np.array_equal(np.array([x.log_likelihood for x in g2])[::-1], ll)
    assert np.array_equal(eng.log_likelihoods(pids, bls, pr), eng.log_likelihoods(pids, bls, pr))
    assert _rel(eng.log_likelihoods(pids, bls, pr), ll) <= 1e-13
    # pattern blocks add up
    cut = 1777
    parts = []
    for sl in (slice(0, cut), slice(cut, P)):
        sub = _engine(tips[:, sl], w[sl], site).gradients(pids, bls, pr)
        parts.append((np.array([x.log_likelihood for x in sub]),
                      np.stack([x.gradient["branch_lengths"] for x in sub]),
                      np.array([x.gradient["site_model"][0] for x in sub])))
    assert _rel(parts[0][0] + parts[1][0], ll) <= 1e-12
    assert _rel(parts[0][1] + parts[1][1], gb) <= 1e-11
    assert _rel(parts[0][2] + parts[1][2], gs) <= 1e-10
    # directional derivative (central difference along a direction proportional to the
    # branch lengths, so that the step is small against every branch: truncation ~h^2)
    d = bls * rng.normal(size=bls.shape)
    d[:, -1] = 0
    h = 1e-5
    lp = eng.log_likelihoods(pids, bls + h * d, pr)
    lm = eng.log_likelihoods(pids, bls - h * d, pr)
    fd = (lp - lm) / (2 * h)
    an = np.sum(gb[:, :2 * n - 2] * d, axis=1)
    assert np.max(np.abs(fd - an) / np.abs(an)) <= 1e-6
    # ... and of the Weibull shape
    dp = pr.copy()
    dp[:, 0] += 1e-6
    dm = pr.copy()
    dm[:, 0] -= 1e-6
    fds = (eng.log_likelihoods(pids, bls, dp) - eng.log_likelihoods(pids, bls, dm)) / 2e-6
    assert np.max(np.abs(fds - gs) / np.abs(gs)) <= 1e-5


def test_builtin_wag_equals_explicit_table_and_errors():
    import libsbn_amd as L
    rng = np.random.default_rng(8)
    n, P = 6, 20
    tips, w = A.random_aa_alignment(n, P, rng)
    pids, bls = TU.random_trees(n, 1, rng)
    pr = np.ones((1, 1))
    ex, fr = _wag()
    assert abs(fr.sum() - 1) < 1e-12 and ex.shape == (190,)
    a = _engine(tips, w, "constant").log_likelihoods(pids, bls, pr)
    b = _engine(tips, w, "constant", model=(ex, fr)).log_likelihoods(pids, bls, pr)
    assert np.array_equal(a, b)
    assert _engine(tips, w, "constant").block_specification() == {
        "clock rate": (0, 1), "entire": (0, 1), "entire clock": (0, 1), "entire site": (0, 0),
        "entire substitution": (0, 0)}
    with pytest.raises(RuntimeError, match="sum to 1"):
        _engine(tips, w, "constant", model=(ex, fr * 1.1))
    # tip partials in SitePattern::GetPartials form give the same result; others are refused
    tp = np.zeros((n, P, 20))
    for i in range(n):
        for p in range(P):
            if tips[i, p] >= 20:
                tp[i, p, :] = 1.0
            else:
                tp[i, p, tips[i, p]] = 1.0
    c = L.Engine(L.PhyloModelSpecification("WAG", "constant", "strict"), None, w,
                 use_tip_states=False, tip_partials=tp).log_likelihoods(pids, bls, pr)
    assert np.array_equal(a, c)
    tp[0, 0, :] = 0.5
    with pytest.raises(RuntimeError, match="one-hot"):
        L.Engine(L.PhyloModelSpecification("WAG", "constant", "strict"), None, w,
                 use_tip_states=False, tip_partials=tp)


def test_kernel_forms_agree_bitwise(tmp_path):
    """The wave-per-block walk kernels (MI_PHYLO_AA_POST=wave / MI_PHYLO_AA_PRE=wave, kept
    selectable) and the workgroup kernels with LDS-DMA staging are the same arithmetic in the
    same order: identical log-likelihoods and gradients.  (The switches are read once per
    process: each form runs in its own interpreter.)"""
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "run_forms.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {repo!r}); sys.path.insert(0, {os.path.join(repo, 'tests')!r})\n"
        "import libsbn_amd as L, aa_utils as A, tree_utils as TU\n"
        "rng = np.random.default_rng(5)\n"
        "tips, w = A.random_aa_alignment(41, 700, rng)\n"
        "pids, bls = TU.random_trees(41, 3, rng)\n"
        "pids[0] = TU.balanced_topology(41)\n"  # (the deepest stack a tree of this size can have)
        "pr = A.params_for('weibull+4', 3, rng)\n"
        "eng = L.Engine(L.PhyloModelSpecification('WAG', 'weibull+4', 'strict'), tips, w)\n"
        "g = eng.gradients(pids, bls, pr)\n"
        "ll = eng.log_likelihoods(pids, bls, pr)\n"
        "out = np.concatenate([ll, [x.log_likelihood for x in g]] + [x.gradient['branch_lengths'] for x in g]"
        " + [np.atleast_1d(x.gradient['site_model']) for x in g])\n"
        "np.save(sys.argv[1], out)\n")
    outs = []
    # (round 5: the log-likelihood walk keeps the top 0 / 1 / 2 / 4 entries of its vector stack
    # in an LDS ring -- MI_PHYLO_AA_RING; a 41-taxon tree's stack reaches three or four entries,
    # so the smaller rings spill to the arena and read it back: same loads, products and stores)
    # (likewise MI_PHYLO_AA_PRE_RING: the pre-order walk parks the vectors it comes back for in
    # a ring of 0 / 1 / 2 entries)
    # (round 6: MI_PHYLO_AA_POST_TILES -- one, two or four 16-pattern tiles per wave of the
    # post-order kernel; a launch short of work takes one by itself, this one is forced either way)
    for post, pre, ring, pre_ring, tiles in (
            ("", "", "", "", "2"), ("wave", "", "", "", ""), ("", "wave", "", "", ""),
            ("wave", "wave", "", "", ""), ("", "", "0", "", "2"), ("", "", "1", "", "2"),
            ("", "", "2", "", "2"), ("", "", "4", "", "2"), ("", "", "", "0", ""),
            ("", "", "", "1", ""), ("", "", "", "2", ""), ("", "", "", "", "1"), ("", "", "2", "", "1"),
            ("", "", "0", "", "1"), ("", "", "", "", "4"), ("", "", "", "", "")):
        env = dict(os.environ)
        env.pop("MI_PHYLO_AA_POST_TILES", None)
        if tiles:
            env["MI_PHYLO_AA_POST_TILES"] = tiles
        env.pop("MI_PHYLO_AA_POST", None)
        env.pop("MI_PHYLO_AA_PRE", None)
        env.pop("MI_PHYLO_AA_RING", None)
        env.pop("MI_PHYLO_AA_PRE_RING", None)
        if post:
            env["MI_PHYLO_AA_POST"] = post
        if pre:
            env["MI_PHYLO_AA_PRE"] = pre
        if ring:
            env["MI_PHYLO_AA_RING"] = ring
        if pre_ring:
            env["MI_PHYLO_AA_PRE_RING"] = pre_ring
        out = tmp_path / f"out_{post}_{pre}_{ring}_{pre_ring}_{tiles}.npy"
        r = subprocess.run([sys.executable, str(script), str(out)], env=env, capture_output=True,
                           text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(np.load(out))
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)


def test_wave_parallel_eigensolver_agrees_with_the_sequential_one(monkeypatch):
    """the 20-state engine's one eigendecomposition (the reference's GTR recipe,
    /root/reference/src/substitution_model.cpp:39-80, with an empirical table) ran 100 sweeps of
    190 one-lane rotations -- 10 ms at every engine creation.  The wave-parallel cyclic Jacobi
    (ten disjoint rotations per step, rounding-level stopping bound) is another rotation order:
    eigenvectors may differ in sign and last bits, P(t) = V exp(L t) V^-1 does not -- results of
    engines made either way agree to 1e-13 (WAG and a random reversible model, very short and
    long branches), and both agree with the oracle."""
    import time
    rng = np.random.default_rng(606)
    n, P, T = 12, 40, 4
    tips, w = A.random_aa_alignment(n, P, rng)
    pids, bls = TU.random_trees(n, T, rng)
    bls[0] *= 1e-4
    bls[1] *= 30.0
    pr = A.params_for("weibull+4", T, rng)
    ex = rng.uniform(0.05, 5.0, size=190)
    fr = rng.dirichlet(5 * np.ones(20))
    for model in (None, (ex, fr)):
        got = {}
        for form in ("seq", "wave"):
            monkeypatch.setenv("MI_PHYLO_AA_JACOBI", form)
            t0 = time.perf_counter()
            eng = _engine(tips, w, "weibull+4", model=model)
            made = time.perf_counter() - t0
            g = eng.gradients(pids, bls, pr)
            got[form] = (np.array([x.log_likelihood for x in g]),
                         np.stack([x.gradient["branch_lengths"] for x in g]),
                         np.array([x.gradient["site_model"][0] for x in g]), made)
            eng.close()
        for a, b in zip(got["seq"][:3], got["wave"][:3]):
            assert np.isfinite(b).all() and _rel(b, a) <= 1e-13, _rel(b, a)
        mex, mfr = model if model is not None else _wag()
        O.set_reversible_model(mex, mfr)
        O.set_transition_mode(1)
        try:
            og = O.unrooted_gradients(A.oracle_spec(n, P, "weibull+4"), tips, w, pids, bls, pr, True, 4)
        finally:
            O.set_transition_mode(0)
        assert _rel(got["wave"][0], og["log_likelihood"]) <= 1e-10
        assert _rel(got["wave"][1], og["branch_lengths"]) <= 1e-10
