"""The reference's own hot-path tests, re-played through the drop-in instance API
(method names and argument meaning of src/pylibsbn.cpp), at the reference's
tolerances:
  src/unrooted_sbn_instance.hpp:206-335, src/rooted_sbn_instance.hpp:246-378,
  vip/test/test_burrito.py:7-52 (hello case), test/test_libsbn.py:95-118.
"""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
D = O.DATA
K = O.load_kats()


def _spec(*a):
    import libsbn_amd as L
    return L.PhyloModelSpecification(*a)


def test_unrooted_likelihood_and_gradient():
    import libsbn_amd as L
    inst = L.unrooted_instance("charlie")
    inst.read_newick_file(os.path.join(D, "hello.nwk"))
    inst.read_fasta_file(os.path.join(D, "hello.fasta"))
    simple = _spec("JC69", "constant", "strict")
    inst.prepare_for_phylo_likelihood(simple, 2)
    for ll in inst.log_likelihoods():
        assert abs(ll - -84.852358) < 0.000001
    inst.read_nexus_file(os.path.join(D, "DS1.subsampled_10.t"))
    inst.read_fasta_file(os.path.join(D, "DS1.fasta"))
    k = K["ds1_sub10_jc69"]
    for vector_flag in ("VECTOR_NONE", "VECTOR_SSE"):
        for tip_state_option in (False, True):
            inst.prepare_for_phylo_likelihood(simple, 2, [vector_flag], tip_state_option)
            likelihoods = inst.log_likelihoods()
            assert np.all(np.abs(likelihoods - k["log_likelihoods"]) < 0.00011)
            gradients = inst.phylo_gradients()
            for g, ref in zip(gradients, k["log_likelihoods"]):
                assert abs(g.log_likelihood - ref) < 0.00011
            last = np.sort(gradients[-1].gradient["branch_lengths"])
            assert np.all(np.abs(last - k["last_tree_sorted_branch_gradient"]) < 0.0001)
            inst.set_rescaling(True)
            assert np.all(np.abs(inst.log_likelihoods() - k["log_likelihoods"]) < 0.00011)
            inst.prepare_for_phylo_likelihood(simple, 1, [], tip_state_option)
            gr = inst.phylo_gradients()
            for g, ref in zip(gr, k["log_likelihoods"]):
                assert abs(g.log_likelihood - ref) < 0.00011
            last = np.sort(gr[-1].gradient["branch_lengths"])
            assert np.all(np.abs(last - k["last_tree_sorted_branch_gradient"]) < 0.0001)
            inst.set_rescaling(False)


def test_unrooted_likelihood_and_gradient_with_weibull():
    import libsbn_amd as L
    inst = L.unrooted_instance("charlie")
    inst.read_nexus_file(os.path.join(D, "DS1.subsampled_10.t"))
    inst.read_fasta_file(os.path.join(D, "DS1.fasta"))
    k = K["ds1_sub10_jc69_weibull4_shape0.1"]
    for tip_state_option in (False, True):
        inst.set_rescaling(False)
        inst.prepare_for_phylo_likelihood(_spec("JC69", "weibull+4", "strict"), 2, [],
                                          tip_state_option)
        inst.get_phylo_model_param_block_map()["Weibull shape"][:] = 0.1
        assert np.all(np.abs(inst.log_likelihoods() - k["log_likelihoods"]) < 0.00011)
        g = inst.phylo_gradients()
        bl0 = np.array([x.gradient["branch_lengths"][0] for x in g])
        assert np.all(np.abs(bl0 - k["branch_gradient_0"]) < 0.00011)
        inst.set_rescaling(True)
        assert np.all(np.abs(inst.log_likelihoods() - k["log_likelihoods"]) < 0.00011)
        g = inst.phylo_gradients()
        bl0 = np.array([x.gradient["branch_lengths"][0] for x in g])
        assert np.all(np.abs(bl0 - k["branch_gradient_0"]) < 0.00011)


def _flu_instance(initialize_time_trees, spec=("JC69", "constant", "strict")):
    import libsbn_amd as L
    inst = L.rooted_instance("charlie")
    inst.read_newick_file(os.path.join(D, "fluA.tree"))
    inst.parse_dates_from_taxon_names(initialize_time_trees)
    inst.read_fasta_file(os.path.join(D, "fluA.fa"))
    inst.prepare_for_phylo_likelihood(_spec(*spec), 1)
    return inst


def test_rooted_gradients():
    k = K["flua_jc69_strict"]
    inst = _flu_instance(True)
    inst.rates[:] = 0.001
    ll = inst.log_likelihoods()
    assert abs(ll[0] - (k["log_likelihood_no_jacobian"] + k["log_det_jacobian"])) < 0.0001
    g = inst.phylo_gradients()
    assert np.all(np.abs(g[0].gradient["ratios_root_height"]
                         - k["ratios_root_height_gradient"]) < 0.0001)
    assert abs(g[0].log_likelihood - k["log_likelihood_no_jacobian"]) < 0.0001


def test_rooted_clock_gradients():
    inst = _flu_instance(True)
    inst.rates[:] = 0.001
    eps = 1e-8

    def fd(j=None):
        saved = inst.rates.copy()
        out = []
        for sign in (1, -1):
            inst.rates[:] = saved
            if j is None:
                inst.rates += sign * eps
            else:
                inst.rates[0, j] += sign * eps
            out.append(inst.log_likelihoods()[0])
        inst.rates[:] = saved
        return (out[0] - out[1]) / (2 * eps)

    g = inst.phylo_gradients()
    assert abs(g[0].gradient["clock_model"][0] - fd()) < 0.001
    inst.rates[0] *= np.arange(inst.rates.shape[1]) % 3 + 1.0
    inst.rate_counts[0] = inst.rates.shape[1]
    g = inst.phylo_gradients()
    for j in (0, 5, 17, 100, inst.rates.shape[1] - 1):
        assert abs(g[0].gradient["clock_model"][j] - fd(j)) < 0.001


def test_rooted_gtr_gradients():
    k = K["flua_gtr"]
    inst = _flu_instance(True, ("GTR", "constant", "strict"))
    inst.rates[:] = 0.001
    blocks = inst.get_phylo_model_param_block_map()
    blocks["frequencies"][:] = k["frequencies"]
    blocks["GTR rates"][:] = k["rates"]
    jac = K["flua_jc69_strict"]["log_det_jacobian"]
    assert abs(inst.log_likelihoods()[0] - (k["log_likelihood_no_jacobian"] + jac)) < 0.001
    g = inst.phylo_gradients()
    assert np.all(np.abs(g[0].gradient["substitution_model"]
                         - k["substitution_model_gradient"]) < 0.001)
    assert abs(g[0].log_likelihood - k["log_likelihood_no_jacobian"]) < 0.001


def test_rooted_weibull_gradients():
    k = K["flua_jc69_weibull4_shape0.1"]
    inst = _flu_instance(True, ("JC69", "weibull+4", "strict"))
    inst.rates[:] = 0.001
    inst.get_phylo_model_param_block_map()["Weibull shape"][:] = 0.1
    jac = K["flua_jc69_strict"]["log_det_jacobian"]
    assert abs(inst.log_likelihoods()[0] - (k["log_likelihood_no_jacobian"] + jac)) < 0.0001
    g = inst.phylo_gradients()
    assert abs(g[0].gradient["site_model"][0] - k["site_model_gradient"]) < 0.001
    assert abs(g[0].log_likelihood - k["log_likelihood_no_jacobian"]) < 0.001


def test_uninitialized_time_trees_raise():
    # rooted_sbn_instance.hpp:399-403 (issue #281): an error, not UB
    inst = _flu_instance(False)
    with pytest.raises(RuntimeError, match="time tree"):
        inst.phylo_gradients()
    assert np.isfinite(inst.unrooted_log_likelihoods()[0])


def test_burrito_hello_and_jc_equals_gtr():
    import libsbn_amd as L
    # vip/test/test_burrito.py:48: hello_out.t branch lengths -> -81.446550
    inst = L.unrooted_instance("burrito")
    inst.read_nexus_file(os.path.join(D, "hello_out.t"))
    inst.read_fasta_file(os.path.join(D, "hello.fasta"))
    inst.prepare_for_phylo_likelihood(_spec("JC69", "constant", "strict"), 2)
    assert inst.log_likelihoods()[0] == pytest.approx(-81.446550, rel=1e-6)
    # test/test_libsbn.py:95-118: JC69 == GTR at JC parameters, DS1 tree 0, bl = 0.1
    inst = L.unrooted_instance("ds1")
    inst.read_newick_file(os.path.join(D, "DS1.100_topologies.nwk"))
    inst.read_fasta_file(os.path.join(D, "DS1.fasta"))
    for bl in inst.tree_collection.branch_lengths:
        bl[:] = 0.1
    inst.prepare_for_phylo_likelihood(_spec("JC69", "constant", "strict"), 2)
    jc = inst.log_likelihoods()
    inst.prepare_for_phylo_likelihood(_spec("GTR", "constant", "strict"), 2)
    blocks = inst.get_phylo_model_param_block_map()
    blocks["GTR rates"][:] = 1 / 6
    blocks["frequencies"][:] = 0.25
    gtr = inst.log_likelihoods()
    assert jc[0] == pytest.approx(gtr[0], rel=1e-6)
    # resized parameter matrix with a different tree count is an error at call time
    inst.resize_phylo_model_params(3)
    with pytest.raises(RuntimeError, match="as many rows"):
        inst.log_likelihoods()
