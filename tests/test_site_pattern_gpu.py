"""Site-pattern compression on the device (mi_site_pattern_compress):
bit-exact against the fixtures dumped with the reference's own SitePattern code
(tests/golden/*.struct.json) and against the CPU oracle on large random alignments --
pattern matrix, weights AND the order (the iteration order of the reference's
unordered_map)."""
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

CODE = {c: i for i, c in enumerate("ACGT")}


def _codes(rows):
    return np.array([[CODE.get(ch.upper(), 4) for ch in row] for row in rows], dtype=np.int8)


@pytest.mark.parametrize("name", ["hello", "ds1_sub10", "flua"])
def test_reference_alignments_bit_exact(name):
    import libsbn_amd as L
    st = O.load_struct(name)
    seqs = O.read_fasta(os.path.join(O.DATA, st["source"]["fasta"]))
    rows = [seqs[t] for t in st["taxon_names"]]
    pats, w, _ = L.site_pattern_compress_device(_codes(rows))
    want_p = np.array(st["patterns"], dtype=np.int32)
    want_w = np.array(st["weights"], dtype=np.float64)
    assert pats.shape == want_p.shape
    assert np.array_equal(pats, want_p)  # same distinct columns in the same order
    assert np.array_equal(w, want_w)


@pytest.mark.parametrize("n,L,alphabet", [(1, 1, 4), (3, 7, 2), (40, 20000, 2), (64, 50000, 5),
                                          (5, 300000, 3)])
def test_random_alignments_match_the_oracle(n, L, alphabet):
    """Heavy duplication (small alphabets) and many distinct columns; the oracle's
    restatement of libstdc++'s unordered_map is itself pinned by the fixtures above."""
    import libsbn_amd as L_
    rng = np.random.default_rng(n * 1000 + L)
    codes = rng.integers(0, alphabet, size=(n, L)).astype(np.int8)
    if L > 100:  # blocks of repeated columns, so that weights > 1 occur at all sizes
        codes[:, L // 2:] = codes[:, :L - L // 2]
    rows = ["".join("ACGT-"[c] for c in row) for row in codes]
    want_p, want_w = O.site_pattern_compress(rows)
    pats, w, ms = L_.site_pattern_compress_device(codes)
    assert pats.shape == want_p.shape
    assert np.array_equal(pats, want_p)
    assert np.array_equal(w, want_w)
    assert w.sum() == L and ms >= 0.0


def test_device_resident_patterns_feed_an_engine_without_the_round_trip():
    """VERDICT r3 (housekeeping): mi_site_pattern_compress_device leaves the pattern matrix
    and the weights in device memory, mi_engine_create_device_tips makes the engine from them
    (states, masks, table offsets and -- use_tip_states=False -- the 0/1 partial vectors are
    derived by a kernel).  The engine's results equal, bit for bit, those of an engine made the
    usual way from the host arrays of the plain compression."""
    import torch
    import libsbn_amd as L
    import tree_utils as TU
    st = O.load_struct("ds1_sub10")
    seqs = O.read_fasta(os.path.join(O.DATA, st["source"]["fasta"]))
    codes = _codes([seqs[t] for t in st["taxon_names"]])
    pats, w, _ = L.site_pattern_compress_device(codes)
    dp, ms = L.site_pattern_compress_device(codes, keep_on_device=True)
    n, P = pats.shape
    assert (dp.taxon_count, dp.pattern_count) == (n, P) and ms >= 0.0
    _, _, pids, bls = O.struct_arrays(st)
    rng = np.random.default_rng(3)
    for subst, site, tip_states in (("JC69", "weibull+4", True), ("GTR", "constant", False)):
        spec = L.PhyloModelSpecification(subst, site, "strict")
        a = L.Engine(spec, pats, w, use_tip_states=tip_states)
        b = L.Engine(spec, None, None, use_tip_states=tip_states, device_tips=dp.as_device_tips())
        T = len(pids)
        pr = np.ones((T, a.param_count))
        if subst == "GTR":
            r, f = TU.random_gtr_params(T, rng)
            pr[:, :6], pr[:, 6:10] = r, f
        assert np.array_equal(a.log_likelihoods(pids, bls, pr), b.log_likelihoods(pids, bls, pr))
        ga, gb = a.gradients(pids, bls, pr), b.gradients(pids, bls, pr)
        assert a.last_call_info() == b.last_call_info()
        for x, y in zip(ga, gb):
            for k in x.gradient:
                assert np.array_equal(x.gradient[k], y.gradient[k]), k
    # 20 states: a random protein alignment through the same two doors
    import aa_utils as A
    tips, aw = A.random_aa_alignment(12, 333, rng)
    d_t = torch.from_numpy(tips).to("cuda:0")
    d_w = torch.from_numpy(aw).to("cuda:0")
    s20 = L.PhyloModelSpecification("WAG", "weibull+4", "strict")
    a = L.Engine(s20, tips, aw)
    b = L.Engine(s20, None, None, device_tips=(d_t.data_ptr(), d_w.data_ptr(), 12, 333))
    p2, b2 = TU.random_trees(12, 2, rng)
    assert np.array_equal(a.log_likelihoods(p2, b2, np.ones((2, 2))),
                          b.log_likelihoods(p2, b2, np.ones((2, 2))))
    dp.release()


def test_device_tips_arguments_are_validated():
    """the device-resident door validates what it is given -- host arrays beside
    device_tips, a 20-state model table, a sharded handle and a HOST pointer passed as a device
    pointer are errors with a message, not silently ignored (or a fault in a kernel)."""
    torch = pytest.importorskip("torch")
    import libsbn_amd as L
    rng = np.random.default_rng(4)
    n, P = 6, 50
    tips = rng.integers(0, 5, size=(n, P)).astype(np.int32)
    w = np.ones(P)
    dev = torch.device("cuda", 0)
    d_t, d_w = torch.from_numpy(tips).to(dev), torch.from_numpy(w).to(dev)
    spec = L.PhyloModelSpecification("JC69", "weibull+4", "strict")
    ok = L.Engine(spec, None, None, device_tips=(d_t.data_ptr(), d_w.data_ptr(), n, P))
    ok.close()
    for kw in ({"tip_partials": np.zeros((n, P, 4))}, {"shard_devices": [0, 0]},
               {"reversible_model": (np.ones(190), np.full(20, 0.05))}):
        with pytest.raises(RuntimeError, match="does not combine"):
            L.Engine(spec, None, None, device_tips=(d_t.data_ptr(), d_w.data_ptr(), n, P), **kw)
    with pytest.raises(RuntimeError, match="does not combine"):
        L.Engine(spec, tips, w, device_tips=(d_t.data_ptr(), d_w.data_ptr(), n, P))
    host = np.ascontiguousarray(tips)
    with pytest.raises(RuntimeError, match="device memory"):
        L.Engine(spec, None, None, device_tips=(host.ctypes.data, d_w.data_ptr(), n, P))
