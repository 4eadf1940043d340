"""Opt-in analytic substitution-model gradient (MI_PHYLO_SUBST_GRADIENT=analytic;
SURVEY.md 8f rank 3).  The default path reproduces the reference's 16-evaluation central
finite differences (fat_beagle.cpp:400-465); this one computes the same derivatives --
w.r.t. the stick-breaking coordinates of the GTR rates and frequencies -- in the gradient
pass itself.  Its parity target is therefore the finite-difference value *up to the
finite-difference error*: checked against the oracle's 80-bit build (whose FD noise is
~1e-9 relative) and against the reference's own known-answer values for fluA."""
import os

import numpy as np
import pytest

import oracle_lib as O
import tree_utils as TU

pytestmark = pytest.mark.gpu

# The analytic gradient runs on the second-generation matrix-core walk (kernels_walk.hip); the
# name means "the matrix-core path ran, not the HBM-streamed fallback".
WALK_KERNEL = ("gradient_walk_kernel",)


@pytest.fixture
def analytic(monkeypatch):
    monkeypatch.setenv("MI_PHYLO_SUBST_GRADIENT", "analytic")  # read at engine creation


def _params(spec, T, **blocks):
    import test_gpu_parity as TG
    return TG._params(spec, T, **blocks)


@pytest.mark.parametrize("site", ["constant", "weibull+4", "weibull+2", "weibull+3"])
def test_matches_extended_precision_finite_differences(analytic, site):
    import libsbn_amd as L
    st = O.load_struct("ds1_sub10")
    tips, w, pids, bls = O.struct_arrays(st)
    T = 4
    pids, bls = pids[:T], bls[:T]
    rng = np.random.default_rng(3)
    gr, gf = TU.random_gtr_params(T, rng)
    eng = L.Engine(L.PhyloModelSpecification("GTR", site, "strict"), tips, w, device=0)
    spec = O.make_spec(27, 934, "GTR", site)
    blocks = {"GTR rates": gr, "frequencies": gf}
    if site != "constant":
        blocks["Weibull shape"] = np.full((T, 1), 0.7)
    pr = _params(spec, T, **blocks)
    g = eng.gradients(pids, bls, pr)
    assert eng.last_call_info()[0] in WALK_KERNEL and eng.last_call_info()[1:] == (T, T)  # no finite-difference passes
    got = np.array([x.gradient["substitution_model"] for x in g])
    O.select("ld")
    O.set_transition_mode(1)
    try:
        og = O.unrooted_gradients(spec, tips, w, pids, bls, pr, False, 4)
    finally:
        O.set_transition_mode(0)
        O.select("f64")
    want = og["substitution_model"]
    assert np.max(np.abs(got - want) / np.maximum(np.abs(want), 1.0)) <= 1e-8
    # everything else is the ordinary gradient pass
    for t in range(T):
        assert abs(g[t].log_likelihood - og["log_likelihood"][t]) <= 1e-10 * abs(og["log_likelihood"][t])
        scale = np.max(np.abs(og["branch_lengths"][t]))
        assert np.max(np.abs(g[t].gradient["branch_lengths"] - og["branch_lengths"][t])) <= 1e-10 * scale
    # exact power-of-two rescaling leaves it unchanged
    gs = eng.gradients(pids, bls, pr, True)
    gots = np.array([x.gradient["substitution_model"] for x in gs])
    assert np.max(np.abs(gots - got) / np.maximum(np.abs(got), 1.0)) <= 1e-12


def test_flua_rooted_known_answers(analytic):
    """The reference's own expected values for this gradient (rooted_sbn_instance.hpp test,
    tolerance 1e-3) were produced analytically by physher."""
    import test_gpu_parity as TG
    k = TG.K["flua_gtr"]
    eng, spec, tips, w, pids, bls, rates, h, bd, ra = TG._flua("GTR")
    pr = _params(spec, 1, **{"GTR rates": k["rates"], "frequencies": k["frequencies"]})
    g = eng.rooted_gradients(pids, bls, pr, rates, [1], h, bd, ra)
    assert eng.last_call_info()[0] in WALK_KERNEL
    assert abs(g[0].log_likelihood - k["log_likelihood_no_jacobian"]) < k["tol"]
    assert np.all(np.abs(g[0].gradient["substitution_model"]
                         - np.array(k["substitution_model_gradient"])) < 1e-4)
