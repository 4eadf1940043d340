"""One handle driving several shards (mi_engine_create_sharded; the counterpart of Engine's
thread_count FatBeagles, src/engine.cpp:23-27), and the variational-inference reductions fused
behind the gradient call (mi_engine_gradients_unrooted_reduced; vip/burrito.py:143-166,
vip/branch_model.py:125-132).  The GPU box has one device, so the shards are logical shards
of device 0: the dealing, staging and gathering are the same code."""
import numpy as np
import pytest

import oracle_lib as O
import tree_utils as TU

pytestmark = pytest.mark.gpu


def _ds1(T, seed=5):
    st = O.load_struct("ds1_top100")
    tips, w, pids, _ = O.struct_arrays(st)
    rng = np.random.default_rng(seed)
    pids = pids[:T]
    bls = rng.exponential(0.1, size=(T, pids.shape[1] + 1))
    bls[:, -1] = 0
    return tips, w, pids, bls, rng


@pytest.mark.parametrize("shards,T", [(2, 11), (3, 7), (4, 3), (1, 5)])
def test_tree_shards_are_bit_identical_to_one_engine(shards, T):
    import libsbn_amd as L
    tips, w, pids, bls, rng = _ds1(T)
    spec = L.PhyloModelSpecification("GTR", "weibull+4", "strict")
    r, f = TU.random_gtr_params(T, rng)
    pr = np.hstack([r, f, np.full((T, 1), 0.7), np.ones((T, 1))])
    one = L.Engine(spec, tips, w)
    many = L.Engine(spec, tips, w, shard_devices=[0] * shards)
    assert many.block_specification() == one.block_specification()
    assert np.array_equal(many.log_likelihoods(pids, bls, pr), one.log_likelihoods(pids, bls, pr))
    ga, gb = one.gradients(pids, bls, pr), many.gradients(pids, bls, pr)
    for a, b in zip(ga, gb):
        assert a.log_likelihood == b.log_likelihood and sorted(a.gradient) == sorted(b.gradient)
        for k in a.gradient:
            assert np.array_equal(a.gradient[k], b.gradient[k]), k
    # per-tree errors still surface (second shard's tree)
    bad = pr.copy()
    bad[-1, 6:10] = 0.1
    with pytest.raises(RuntimeError, match="frequencies"):
        many.gradients(pids, bls, bad)
    # the device-pointer entry points need a single-device engine
    with pytest.raises(RuntimeError, match="single-device"):
        many.log_likelihoods_device(None, T, 0, 0, 0, 0)


def test_relative_device_ordinals_wrap_around():
    """devices = {-1, -2, -3} means "the current device, the next one, the one after", wrapping
    around the visible devices (what a one-process-per-GPU launch relies on to stay on the
    device it selected; engine.cpp:23-27 is the reference's thread_count counterpart).  On a
    box with one GPU every shard lands on device 0 -- and on any box the placement is
    (current + k) mod count and the results are a single engine's, bit for bit."""
    import libsbn_amd as L
    from libsbn_amd import _capi
    T = 9
    tips, w, pids, bls, rng = _ds1(T)
    spec = L.PhyloModelSpecification("JC69", "weibull+4", "strict")
    pr = np.ones((T, 2)); pr[:, 0] = rng.uniform(0.4, 1.6, T)
    count = _capi.load().mi_device_count()
    assert count >= 1
    one = L.Engine(spec, tips, w)
    current = one.shard_devices()[0]
    for devs in ([-1, -2, -3], None):
        many = (L.Engine(spec, tips, w, shard_devices=devs) if devs is not None
                else L.Engine(spec, tips, w, shard_devices=[-1 - k for k in range(2)]))
        k = len(many.shard_devices())
        assert many.shard_devices() == [(current + i) % count for i in range(k)]
        assert np.array_equal(many.log_likelihoods(pids, bls, pr), one.log_likelihoods(pids, bls, pr))
        for a, b in zip(one.gradients(pids, bls, pr), many.gradients(pids, bls, pr)):
            assert a.log_likelihood == b.log_likelihood
            for key in a.gradient:
                assert np.array_equal(a.gradient[key], b.gradient[key]), key
        many.close()
    with pytest.raises(RuntimeError, match="out of range"):
        L.Engine(spec, tips, w, shard_devices=[count])
    one.close()


def test_rooted_tree_shards():
    import libsbn_amd as L
    rng = np.random.default_rng(2)
    n, P, T = 9, 40, 5
    tips, w = TU.random_alignment(n, P, rng)
    trees = [TU.clocklike_rooted_tree(n, rng) for _ in range(T)]
    pids = np.stack([t[0] for t in trees])
    bls = np.stack([t[1] for t in trees])
    st = [O.time_tree_init(n, t[0], t[1], t[2]) for t in trees]
    h, bd, ra = (np.stack([s[i] for s in st]) for i in range(3))
    rates = np.full((T, 2 * n - 2), 0.5)
    counts = np.ones(T, np.int32)
    spec = L.PhyloModelSpecification("JC69", "weibull+4", "strict")
    pr = np.ones((T, 2))
    one, many = L.Engine(spec, tips, w), L.Engine(spec, tips, w, shard_devices=[0, 0, 0])
    assert np.array_equal(one.rooted_log_likelihoods(pids, bls, pr, rates, h, bd),
                          many.rooted_log_likelihoods(pids, bls, pr, rates, h, bd))
    ga = one.rooted_gradients(pids, bls, pr, rates, counts, h, bd, ra)
    gb = many.rooted_gradients(pids, bls, pr, rates, counts, h, bd, ra)
    for a, b in zip(ga, gb):
        for k in a.gradient:
            assert np.array_equal(a.gradient[k], b.gradient[k]), k


def test_pattern_shards_add_up():
    import libsbn_amd as L
    tips, w, pids, bls, rng = _ds1(4)
    spec = L.PhyloModelSpecification("JC69", "weibull+4", "strict")
    pr = np.ones((4, 2))
    one = L.Engine(spec, tips, w)
    many = L.Engine(spec, tips, w, shard_devices=[0, 0, 0], shard_mode="patterns")
    a, b = one.log_likelihoods(pids, bls, pr), many.log_likelihoods(pids, bls, pr)
    assert np.max(np.abs(a - b) / np.abs(a)) <= 1e-13
    ga, gb = one.gradients(pids, bls, pr), many.gradients(pids, bls, pr)
    for x, y in zip(ga, gb):
        for k in x.gradient:
            assert np.max(np.abs(x.gradient[k] - y.gradient[k])) <= 1e-11 * np.max(
                np.abs(x.gradient[k])), k
    with pytest.raises(RuntimeError, match="unrooted calls only"):
        many.rooted_log_likelihoods(np.zeros((1, 52), np.int32), np.zeros((1, 53)), pr[:1],
                                    with_jacobian=False)
    # the 20-state engine, pattern-sharded (the 8-GPU form of BASELINE.json configs[4])
    import aa_utils as A
    rng = np.random.default_rng(4)
    tips, w = A.random_aa_alignment(12, 300, rng)
    p2, b2 = TU.random_trees(12, 2, rng)
    pr2 = np.ones((2, 2))
    s20 = L.PhyloModelSpecification("WAG", "weibull+4", "strict")
    x = L.Engine(s20, tips, w).gradients(p2, b2, pr2)
    y = L.Engine(s20, tips, w, shard_devices=[0, 0], shard_mode="patterns").gradients(p2, b2, pr2)
    for u, v in zip(x, y):
        assert abs(u.log_likelihood - v.log_likelihood) <= 1e-12 * abs(u.log_likelihood)
        assert np.max(np.abs(u.gradient["branch_lengths"] - v.gradient["branch_lengths"])) <= \
            1e-11 * np.max(np.abs(u.gradient["branch_lengths"]))


@pytest.mark.parametrize("shards", [None, 3])
def test_fused_vi_reductions_match_host_scatter_add_of_oracle_results(shards):
    """HIP output vs the oracle's per-tree results scatter-added on the host the way
    vip/branch_model.py:125-132 does (np.add.at by split index), incl. per-tree weights."""
    import libsbn_amd as L
    T = 24
    tips, w, pids, bls, rng = _ds1(T, seed=9)
    n, N = 27, 53
    spec = L.PhyloModelSpecification("JC69", "weibull+4", "strict")
    pr = np.ones((T, 2))
    pr[:, 0] = rng.uniform(0.5, 1.5, T)
    index_count = 300
    bi = rng.integers(0, index_count, size=(T, N)).astype(np.int32)
    bi[:, -2:] = -1          # the fixed node and the root are not parameters
    bi[rng.random((T, N)) < 0.05] = -1
    tw = rng.uniform(0.1, 2.0, T)
    ospec = O.make_spec(n, 934, "JC69", "weibull+4")
    og = O.unrooted_gradients(ospec, tips, w, pids, bls, pr, False, 4)
    eng = L.Engine(spec, tips, w, shard_devices=None if shards is None else [0] * shards)
    for weights in (None, tw):
        wt = np.ones(T) if weights is None else weights
        want = np.zeros(index_count)
        for t in range(T):
            ok = bi[t] >= 0
            np.add.at(want, bi[t][ok], wt[t] * og["branch_lengths"][t][ok])
        s_ll, s_site, ig, ll = eng.gradients_reduced(pids, bls, pr, bi, index_count, weights)
        assert abs(s_ll - np.sum(wt * og["log_likelihood"])) <= 1e-10 * abs(s_ll)
        assert abs(s_site - np.sum(wt * og["site_model"])) <= 1e-9 * abs(s_site)
        assert np.max(np.abs(ig - want)) <= 1e-10 * np.max(np.abs(want))
        assert np.max(np.abs(ll - og["log_likelihood"]) / np.abs(ll)) <= 1e-10
    # reproducible bit for bit, and equal to the engine's own per-tree results reduced on the host
    a = eng.gradients_reduced(pids, bls, pr, bi, index_count, tw)
    b = eng.gradients_reduced(pids, bls, pr, bi, index_count, tw)
    assert a[0] == b[0] and np.array_equal(a[2], b[2])


def test_rooted_gradient_does_not_read_stale_lds():
    """the LDS form of the rooted chain rule read the slot of a LEAF child before
    anything had written it; residue of an earlier workgroup (NaN bit patterns) then
    poisoned 0 * x.  Fill LDS with NaNs by an unrelated call pattern first: run the gradient
    kernels of a rescaled call (int16 exponents / -1 words in LDS), then a rooted gradient."""
    import libsbn_amd as L
    rng = np.random.default_rng(31)
    n, P = 12, 64
    tips, w = TU.random_alignment(n, P, rng)
    tree = TU.clocklike_rooted_tree(n, rng)
    h, bd, ra = O.time_tree_init(n, tree[0], tree[1], tree[2])
    spec = L.PhyloModelSpecification("JC69", "constant", "strict")
    eng = L.Engine(spec, tips, w)
    args = (tree[0][None], tree[1][None], np.ones((1, 1)), np.full((1, 2 * n - 2), 0.3),
            np.ones(1, np.int32), h[None], bd[None], ra[None])
    first = eng.rooted_gradients(*args)[0].gradient["ratios_root_height"]
    # poison: NaN tip partials make every kernel that stages them leave NaNs in LDS
    upids, ubls = TU.random_trees(n, 64, rng)
    for _ in range(3):
        eng.gradients(upids, ubls, np.ones((64, 1)), rescaling=True)
        again = eng.rooted_gradients(*args)[0].gradient["ratios_root_height"]
        assert np.all(np.isfinite(again)) and np.array_equal(again, first)


def test_device_call_replays_from_a_hip_graph():
    """include/mi_phylo.h: after mi_engine_reserve a *_device call allocates nothing and can
    be captured in a hipGraph.  Capture one gradient call (4 states) and one 20-state call,
    change the inputs in place, replay: results equal the eager call's bit for bit."""
    import torch
    import libsbn_amd as L
    import aa_utils as A
    dev = torch.device("cuda", 0)
    cases = []
    tips, w, pids, bls, rng = _ds1(20)
    cases.append((L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w),
                  pids, bls, np.ones((20, 2))))
    rng = np.random.default_rng(6)
    tips, w = A.random_aa_alignment(10, 70, rng)
    p2, b2 = TU.random_trees(10, 3, rng)
    cases.append((L.Engine(L.PhyloModelSpecification("WAG", "weibull+4", "strict"), tips, w),
                  p2, b2, np.ones((3, 2))))
    # two more, captured WITHOUT a warm-up call: the arena variant of the 4-state gradient
    # kernel (workgroup-per-tree set-up and macro-slot kernels) and a 20-state engine whose
    # tree set-up kernel needs more than 64 KiB of LDS (opt-in attribute set at first launch)
    tips, w = TU.random_alignment(40, 200, rng)
    p3, b3 = TU.random_trees(40, 5, rng)
    cases.append((L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w),
                  p3, b3, np.ones((5, 2))))
    tips, w = A.random_aa_alignment(600, 64, rng)
    p4, b4 = TU.random_trees(600, 2, rng)
    cases.append((L.Engine(L.PhyloModelSpecification("WAG", "weibull+4", "strict"), tips, w),
                  p4, b4, np.ones((2, 2))))
    for case, (eng, pids, bls, pr) in enumerate(cases):
        T, N = len(pids), eng.node_count
        d_pid = torch.from_numpy(np.ascontiguousarray(pids)).to(dev)
        d_bl = torch.from_numpy(np.ascontiguousarray(bls)).to(dev)
        d_pr = torch.from_numpy(pr).to(dev)
        ll = torch.zeros(T, dtype=torch.float64, device=dev)
        g = torch.zeros((T, N), dtype=torch.float64, device=dev)
        site = torch.zeros(T, dtype=torch.float64, device=dev)
        eng.reserve(T, True)

        def call(stream):
            eng.gradients_device(stream, T, d_pid.data_ptr(), d_bl.data_ptr(), d_pr.data_ptr(),
                                 ll.data_ptr(), g.data_ptr(), site.data_ptr(), None)
        if case < 2:
            call(torch.cuda.current_stream().cuda_stream)  # warm-up outside the capture
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=torch.cuda.Stream()):
            call(torch.cuda.current_stream().cuda_stream)
        # new branch lengths in the captured buffers
        bl2 = bls * rng.uniform(0.5, 1.5, size=bls.shape)
        d_bl.copy_(torch.from_numpy(bl2))
        ll.zero_(), g.zero_()
        graph.replay()
        torch.cuda.synchronize()
        eng.check_status(None)
        got_ll, got_g = ll.cpu().numpy().copy(), g.cpu().numpy().copy()
        want = eng.gradients(pids, bl2, pr)
        assert np.array_equal(got_ll, np.array([x.log_likelihood for x in want]))
        assert np.array_equal(got_g, np.stack([x.gradient["branch_lengths"] for x in want]))


def test_reduced_device_call_allocates_nothing_after_reserve_reduced():
    """mi_engine_reserve_reduced: the per-tree buffers and the index sort's
    workspace of the fused-reduction call are reserved up front, so the *_device call that
    follows changes the free device memory by nothing and replays from a hipGraph; its sums
    equal the host scatter-add of the per-tree results bit for bit."""
    torch = pytest.importorskip("torch")
    import libsbn_amd as L
    from libsbn_amd import _capi
    T, index_count = 40, 300
    tips, w, pids, bls, rng = _ds1(T)
    N = 53
    spec = L.PhyloModelSpecification("JC69", "weibull+4", "strict")
    pr = np.ones((T, 2)); pr[:, 0] = rng.uniform(0.4, 1.6, T)
    bi = rng.integers(-1, index_count, size=(T, N)).astype(np.int32)
    dev = torch.device("cuda", 0)
    d_pid, d_bl, d_par, d_bi = (torch.from_numpy(x).to(dev) for x in (pids, bls, pr, bi))
    packed = torch.zeros(2 + index_count, dtype=torch.float64, device=dev)
    d_ll = torch.zeros(T, dtype=torch.float64, device=dev)
    gs = torch.cuda.Stream()
    # (another engine runs the same call first: the kernels' code objects are loaded -- into
    # device memory -- at their first launch, which is not the engine's allocation)
    warm = L.Engine(spec, tips, w)
    rc = warm._lib.mi_engine_gradients_unrooted_reduced_device(
        warm._h, gs.cuda_stream, T, d_pid.data_ptr(), d_bl.data_ptr(), d_par.data_ptr(), 0,
        d_bi.data_ptr(), None, index_count, packed.data_ptr(), packed.data_ptr() + 16, d_ll.data_ptr())
    assert rc == 0, _capi.last_error()
    torch.cuda.synchronize()
    eng = L.Engine(spec, tips, w)
    eng.reserve_reduced(T, index_count)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]

    def call(stream):
        rc = eng._lib.mi_engine_gradients_unrooted_reduced_device(
            eng._h, stream, T, d_pid.data_ptr(), d_bl.data_ptr(), d_par.data_ptr(), 0,
            d_bi.data_ptr(), None, index_count, packed.data_ptr(), packed.data_ptr() + 16,
            d_ll.data_ptr())
        assert rc == 0, _capi.last_error()

    with torch.cuda.stream(gs):
        call(gs.cuda_stream)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=gs):
        call(torch.cuda.current_stream().cuda_stream)
    packed.zero_()
    graph.replay()
    torch.cuda.synchronize()
    eng.check_status()
    g = eng.gradients(pids, bls, pr)
    want = np.zeros(2 + index_count)
    want[0] = sum(x.log_likelihood for x in g)   # (tree order)
    want[1] = sum(np.atleast_1d(x.gradient["site_model"])[0] for x in g)
    gb = np.stack([x.gradient["branch_lengths"] for x in g])
    keep = bi >= 0
    np.add.at(want[2:], bi[keep], gb[keep])
    got = packed.cpu().numpy()
    assert np.array_equal(got[2:], want[2:])
    assert np.allclose(got[:2], want[:2], rtol=1e-13, atol=0)
    with pytest.raises(RuntimeError, match="index_count"):
        eng.reserve_reduced(T, -1)
    eng.close(); warm.close()
