"""Oracle, floating-point half: every known-answer value the reference's tests
hold for the Engine/FatBeagle path (tests/golden/reference_kats.json), at the
tolerance the reference's test states."""
import numpy as np
import pytest

import oracle_lib as O

K = O.load_kats()


@pytest.fixture(autouse=True, params=["unrolled_s4", "generic_s"])
def _state_loops(request):
    """Every known-answer test runs twice: with the compiler-unrolled s = 4 copy of the
    oracle's sweeps and with the s-generic loops (orc_set_generic_states) -- the code path
    the 20-state parity tests (tests/test_aa_gpu.py) rely on, which has no reference values
    of its own (the reference is DNA-only)."""
    O.set_generic_states(request.param == "generic_s")
    yield
    O.set_generic_states(False)


def _params(spec, T, **blocks):
    pc = O.param_count(spec)
    lay = O.param_layout(spec)
    pr = np.zeros((T, max(pc, 1)))
    for key, val in blocks.items():
        off = lay[key]
        val = np.atleast_1d(val)
        pr[:, off:off + len(val)] = val
    if lay["clock rate"] >= 0 and "clock rate" not in blocks:
        pr[:, lay["clock rate"]] = 1.0
    return pr[:, :pc] if pc else pr[:, :0]


def test_weibull_rates():
    k = K["weibull_rates"]
    r, w, _ = O.weibull_rates(4, 1.0)
    assert np.allclose(r, k["shape_1.0"], atol=k["tol"], rtol=0)
    r2, w2, _ = O.weibull_rates(4, 0.1)
    assert np.allclose(r2, k["shape_0.1"], atol=k["tol"], rtol=0)
    assert np.allclose(w, 0.25) and abs(r @ w - 1) < 1e-4 and abs(r2 @ w2 - 1) < 1e-4


def test_weibull_rate_derivative_matches_finite_difference():
    for shape in (0.1, 0.5, 1.0, 2.3):
        _, _, d = O.weibull_rates(4, shape)
        e = 1e-6 * shape
        rp, _, _ = O.weibull_rates(4, shape + e)
        rm, _, _ = O.weibull_rates(4, shape - e)
        assert np.allclose(d, (rp - rm) / (2 * e), rtol=1e-5, atol=1e-9)


def test_gtr_eigenvalues():
    k = K["gtr_eigenvalues"]
    spec = O.make_spec(4, 1, "GTR", "constant", "strict")
    pr = _params(spec, 1, **{"GTR rates": k["rates"], "frequencies": k["frequencies"]})
    m = O.model_set(spec, pr[0])
    ev = sorted(list(m.lam)[:4])
    assert np.allclose(ev, sorted(k["eigenvalues"]), atol=k["tol"], rtol=0)
    # V * Vinv = I ; V diag(l) Vinv = Q
    V = np.array(list(m.V)[:16]).reshape(4, 4)
    Vi = np.array(list(m.Vinv)[:16]).reshape(4, 4)
    Q = np.array(list(m.Q)[:16]).reshape(4, 4)
    assert np.allclose(V @ Vi, np.eye(4), atol=1e-13)
    assert np.allclose(V @ np.diag(list(m.lam)[:4]) @ Vi, Q, atol=1e-13)
    assert np.allclose(Q.sum(1), 0, atol=1e-15)


def test_gtr_default_equals_jc69():
    gtr = O.make_spec(4, 1, "GTR", "constant", "strict")
    pr = _params(gtr, 1, **{"GTR rates": [1 / 6] * 6, "frequencies": [0.25] * 4})
    m = O.model_set(gtr, pr[0])
    jc = O.model_set(O.make_spec(4, 1, "JC69", "constant", "strict"), np.array([1.0]))
    assert np.allclose(sorted(list(m.lam)[:4]), sorted(list(jc.lam)[:4]), atol=1e-4)
    assert np.allclose(list(m.Q)[:16], list(jc.Q)[:16], atol=1e-15)


def test_gtr_rejects_bad_sums():
    gtr = O.make_spec(4, 1, "GTR", "constant", "strict")
    with pytest.raises(RuntimeError, match="frequencies do not sum"):
        O.model_set(gtr, _params(gtr, 1, **{"GTR rates": [1 / 6] * 6,
                                            "frequencies": [0.3] * 4})[0])
    with pytest.raises(RuntimeError, match="rates do not sum"):
        O.model_set(gtr, _params(gtr, 1, **{"GTR rates": [0.2] * 6,
                                            "frequencies": [0.25] * 4})[0])


def test_stick_breaking():
    k = K["stick_breaking"]
    x = O.stick_breaking(k["y"])
    assert np.allclose(x, k["x"], atol=k["tol"], rtol=0)
    assert np.allclose(O.stick_breaking_inverse(x), k["y"], atol=k["tol"], rtol=0)
    assert abs(x.sum() - 1) < 1e-15


def test_block_layout():
    # GTR+weibull+strict row = rates[0:6] freqs[6:10] shape[10] clock[11]
    spec = O.make_spec(4, 1, "GTR", "weibull+4", "strict")
    assert O.param_count(spec) == 12
    assert O.param_layout(spec) == {"GTR rates": 0, "frequencies": 6, "Weibull shape": 10,
                                    "clock rate": 11}
    spec = O.make_spec(4, 1, "JC69", "constant", "strict")
    assert O.param_count(spec) == 1
    assert O.param_count(O.make_spec(4, 1, "JC69", "constant", "none")) == 0


@pytest.mark.parametrize("name,key", [("hello", "hello_jc69"), ("hello_out", "hello_out_jc69")])
def test_hello(name, key):
    st = O.load_struct(name)
    tips, w, pids, bls = O.struct_arrays(st)
    spec = O.make_spec(st["taxon_count"], st["pattern_count"])
    for resc in (False, True):
        ll = O.unrooted_log_likelihoods(spec, tips, w, pids, bls, _params(spec, 1), resc)
        k = K[key]
        tol = k.get("tol", abs(k["log_likelihood"]) * k.get("rel_tol", 0))
        assert abs(ll[0] - k["log_likelihood"]) < tol


@pytest.mark.parametrize("rescaling", [False, True])
def test_ds1_jc69(rescaling):
    st = O.load_struct("ds1_sub10")
    k = K["ds1_sub10_jc69"]
    tips, w, pids, bls = O.struct_arrays(st)
    spec = O.make_spec(st["taxon_count"], st["pattern_count"])
    pr = _params(spec, 10)
    ll = O.unrooted_log_likelihoods(spec, tips, w, pids, bls, pr, rescaling, nthreads=4)
    assert np.all(np.abs(ll - k["log_likelihoods"]) < k["ll_tol"])
    g = O.unrooted_gradients(spec, tips, w, pids, bls, pr, rescaling, nthreads=4)
    assert np.all(np.abs(g["log_likelihood"] - k["log_likelihoods"]) < k["ll_tol"])
    last = np.sort(g["branch_lengths"][-1])
    assert last.shape == (53,)
    assert np.all(np.abs(last - k["last_tree_sorted_branch_gradient"]) < k["grad_tol"])
    # two exact zeros: root and the fixed child of the root (fat_beagle.cpp:499)
    assert np.all(g["branch_lengths"][:, -2:] == 0.0)
    assert "site_model" not in g and "substitution_model" not in g


@pytest.mark.parametrize("rescaling", [False, True])
def test_ds1_weibull(rescaling):
    st = O.load_struct("ds1_sub10")
    k = K["ds1_sub10_jc69_weibull4_shape0.1"]
    tips, w, pids, bls = O.struct_arrays(st)
    spec = O.make_spec(st["taxon_count"], st["pattern_count"], "JC69", "weibull+4")
    pr = _params(spec, 10, **{"Weibull shape": k["shape"]})
    ll = O.unrooted_log_likelihoods(spec, tips, w, pids, bls, pr, rescaling, nthreads=4)
    assert np.all(np.abs(ll - k["log_likelihoods"]) < k["tol"])
    g = O.unrooted_gradients(spec, tips, w, pids, bls, pr, rescaling, nthreads=4)
    assert np.all(np.abs(g["branch_lengths"][:, 0] - k["branch_gradient_0"]) < k["tol"])
    assert g["site_model"].shape == (10,)


def _flua(subst="JC69", site="constant"):
    st = O.load_struct("flua")
    n = st["taxon_count"]
    tips, w, pids, bls = O.struct_arrays(st)
    dates = O.parse_dates_from_names(st["taxon_names"])
    h, bd, ra = O.time_tree_init(n, pids[0], bls[0], dates)
    spec = O.make_spec(n, st["pattern_count"], subst, site, "strict")
    rates = np.full((1, 2 * n - 2), 0.001)
    return st, spec, tips, w, pids, bls, rates, h[None], bd[None], ra[None]


def test_rooted_tree_example():
    k = K["rooted_tree_example"]
    h, bd, ra = O.time_tree_init(4, k["parent_ids"], k["branch_lengths"], k["tip_dates"])
    assert h.tolist() == k["node_heights"]
    assert bd.tolist() == k["node_bounds"]
    assert ra.tolist() == [1. / 3.5, 1.5 / 4., 7.]


def test_time_tree_rejects_non_clocklike():
    k = K["rooted_tree_example"]
    bl = list(k["branch_lengths"])
    bl[0] += 0.01
    with pytest.raises(RuntimeError, match="time-calibrated"):
        O.time_tree_init(4, k["parent_ids"], bl, k["tip_dates"])


def test_flua_rooted_jc69():
    k = K["flua_jc69_strict"]
    st, spec, tips, w, pids, bls, rates, h, bd, ra = _flua()
    pr = _params(spec, 1)
    ll = O.rooted_log_likelihoods(spec, tips, w, pids, bls, pr, rates, h, bd, True)
    assert abs(ll[0] - (k["log_likelihood_no_jacobian"] + k["log_det_jacobian"])) < k["tol"]
    ll0 = O.rooted_log_likelihoods(spec, tips, w, pids, bls * 0.001, pr, rates, h, bd, False)
    assert abs(ll0[0] - k["log_likelihood_no_jacobian"]) < k["tol"]
    g = O.rooted_gradients(spec, tips, w, pids, bls, pr, rates, [1], h, bd, ra)
    assert abs(g["log_likelihood"][0] - k["log_likelihood_no_jacobian"]) < k["tol"]
    assert np.all(np.abs(g["ratios_root_height"][0] - k["ratios_root_height_gradient"])
                  < k["tol"])


def test_flua_clock_gradient_vs_finite_difference():
    # rooted_sbn_instance.hpp:288-324 (DerivativeStrictClock / DerivativeRelaxedClock)
    k = K["flua_clock"]
    st, spec, tips, w, pids, bls, rates, h, bd, ra = _flua()
    pr = _params(spec, 1)
    n = st["taxon_count"]
    g = O.rooted_gradients(spec, tips, w, pids, bls, pr, rates, [1], h, bd, ra)
    eps = k["fd_eps"]
    f = lambda r: O.rooted_log_likelihoods(spec, tips, w, pids, bls, pr, r, h, bd, True)[0]
    fd = (f(rates + eps) - f(rates - eps)) / (2 * eps)
    assert abs(g["clock_model"][0, 0] - fd) < k["tol"]
    # relaxed clock: rate_i *= i%3+1, rate_count = #branches
    rel = rates * (np.arange(2 * n - 2) % 3 + 1.0)
    g2 = O.rooted_gradients(spec, tips, w, pids, bls, pr, rel, [2 * n - 2], h, bd, ra)
    for j in (0, 1, 7, 50, 2 * n - 3):
        rp, rm = rel.copy(), rel.copy()
        rp[0, j] += eps
        rm[0, j] -= eps
        assert abs(g2["clock_model"][0, j] - (f(rp) - f(rm)) / (2 * eps)) < k["tol"]
    with pytest.raises(RuntimeError):
        O.rooted_gradients(spec, tips, w, pids, bls, pr, rel, [3], h, bd, ra)


def test_flua_gtr():
    k = K["flua_gtr"]
    st, spec, tips, w, pids, bls, rates, h, bd, ra = _flua("GTR")
    pr = _params(spec, 1, **{"GTR rates": k["rates"], "frequencies": k["frequencies"]})
    jac = K["flua_jc69_strict"]["log_det_jacobian"]
    ll = O.rooted_log_likelihoods(spec, tips, w, pids, bls, pr, rates, h, bd, True)
    assert abs(ll[0] - (k["log_likelihood_no_jacobian"] + jac)) < k["tol"]
    g = O.rooted_gradients(spec, tips, w, pids, bls, pr, rates, [1], h, bd, ra)
    assert abs(g["log_likelihood"][0] - k["log_likelihood_no_jacobian"]) < k["tol"]
    assert np.all(np.abs(g["substitution_model"][0] - k["substitution_model_gradient"])
                  < k["tol"])


def test_flua_weibull():
    k = K["flua_jc69_weibull4_shape0.1"]
    st, spec, tips, w, pids, bls, rates, h, bd, ra = _flua("JC69", "weibull+4")
    pr = _params(spec, 1, **{"Weibull shape": k["shape"]})
    jac = K["flua_jc69_strict"]["log_det_jacobian"]
    ll = O.rooted_log_likelihoods(spec, tips, w, pids, bls, pr, rates, h, bd, True)
    assert abs(ll[0] - (k["log_likelihood_no_jacobian"] + jac)) < k["ll_tol"]
    g = O.rooted_gradients(spec, tips, w, pids, bls, pr, rates, [1], h, bd, ra)
    assert abs(g["site_model"][0] - k["site_model_gradient"]) < k["grad_tol"]
    assert abs(g["log_likelihood"][0] - k["log_likelihood_no_jacobian"]) < 0.001


def test_jc69_equals_gtr_at_jc_params():
    # test/test_libsbn.py:95-118: DS1 tree 0, all bl = 0.1
    st = O.load_struct("ds1_top100")
    tips, w, pids, bls = O.struct_arrays(st)
    pids, bls = pids[:1], np.full_like(bls[:1], 0.1)
    jc = O.make_spec(27, 934)
    gtr = O.make_spec(27, 934, "GTR")
    a = O.unrooted_log_likelihoods(jc, tips, w, pids, bls, _params(jc, 1))
    b = O.unrooted_log_likelihoods(gtr, tips, w, pids, bls,
                                   _params(gtr, 1, **{"GTR rates": [1 / 6] * 6,
                                                      "frequencies": [0.25] * 4}))
    assert abs(a[0] - b[0]) < 1e-6 * abs(a[0])


def test_branch_gradient_matches_finite_difference_gtr_weibull():
    """Independent check of the analytic pre-order gradient (no golden exists for
    GTR+weibull on unrooted trees): central differences of the oracle's own logL."""
    st = O.load_struct("five_taxon")
    tips, w, pids, bls = O.struct_arrays(st)
    pids, bls = pids[:1], bls[:1] + 0.05
    n = 5
    spec = O.make_spec(n, st["pattern_count"], "GTR", "weibull+4")
    pr = _params(spec, 1, **{"GTR rates": [0.05, 0.1, 0.15, 0.2, 0.25, 0.25],
                             "frequencies": [0.1, 0.2, 0.3, 0.4], "Weibull shape": 0.7})
    g = O.unrooted_gradients(spec, tips, w, pids, bls, pr)
    f = lambda b, p=pr: O.unrooted_log_likelihoods(spec, tips, w, pids, b, p)[0]
    eps = 1e-6
    for j in range(2 * n - 3):  # root entry of the trifurcating tree is not an edge
        bp, bm = bls.copy(), bls.copy()
        bp[0, j] += eps
        bm[0, j] -= eps
        assert abs(g["branch_lengths"][0, j] - (f(bp) - f(bm)) / (2 * eps)) < 1e-5
    lay = O.param_layout(spec)
    pp, pm = pr.copy(), pr.copy()
    pp[0, lay["Weibull shape"]] += eps
    pm[0, lay["Weibull shape"]] -= eps
    # site gradient is evaluated at the FD-perturbed substitution model (see
    # subst_gradient_fd); 1e-6 perturbation -> compare loosely
    assert abs(g["site_model"][0] - (f(bls, pp) - f(bls, pm)) / (2 * eps)) < 1e-4


def test_reversible_model_at_four_states_is_gtr():
    """The 'reversible' model (a table, no free parameters: what the 20-state engine uses)
    goes through the reference's GTR recipe: at s = 4 with GTR's own rates and frequencies it
    must give GTR's results exactly."""
    rng = np.random.default_rng(12)
    st = O.load_struct("ds1_sub10")
    tips, w, pids, bls = O.struct_arrays(st)
    pids, bls = pids[:3], bls[:3]
    rates = rng.dirichlet(10 * np.ones(6))
    freqs = rng.dirichlet(10 * np.ones(4))
    gtr = O.make_spec(27, 934, "GTR", "weibull+4", "strict")
    pr = _params(gtr, 3, **{"GTR rates": rates, "frequencies": freqs, "Weibull shape": 0.9})
    a = O.unrooted_gradients(gtr, tips, w, pids, bls, pr, False, 2)
    O.set_reversible_model(rates, freqs)
    rev = O.make_spec(27, 934, "reversible", "weibull+4", "strict")
    pr2 = np.ones((3, 2))
    pr2[:, 0] = 0.9
    b = O.unrooted_gradients(rev, tips, w, pids, bls, pr2, False, 2)
    assert np.array_equal(a["log_likelihood"], b["log_likelihood"])
    assert np.array_equal(a["branch_lengths"], b["branch_lengths"])
    # (GTR's site-model pass runs under the model its finite-difference substitution gradient
    # leaves perturbed by 1e-6 -- fat_beagle.cpp:433-436; a model
    # without free parameters has no such pass)
    assert np.max(np.abs(a["site_model"] - b["site_model"]) / np.abs(a["site_model"])) < 1e-6
