"""The one-launch small call (round 5): tree set-up, model instances and
operand records run as the first workgroups of the gradient walk's own launch
(`gradient_walk_lut_fused_kernel`, libsbn_amd/csrc/kernels_walk3.hip) and hand each tree to
its walk waves through a per-tree word INSIDE the kernel.  The arithmetic is the four-launch
sequence's, instruction for instruction (same device functions): every output must be
BIT-IDENTICAL to `MI_PHYLO_FUSED_SETUP=0` -- at 1 / 125 / 1000 trees, rescaled or not, rooted
or not, for the GTR call that asks for the branch-length gradient only, after an error, and
when calls of different sizes re-use the hand-off words.  Replaces the per-tree dispatch of
/root/reference/src/fat_beagle.hpp:119-149 (FatBeagleParallelize) for small batches.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
import tree_utils as TU
from test_gpu_parity import _params

pytestmark = pytest.mark.gpu

FUSED = "gradient_walk_lut_fused_kernel"
PLAIN = "gradient_walk_lut_kernel"


def _engines(subst, site, tips, w, **kw):
    """(one-launch engine, four-launch engine): the switch is read at engine creation."""
    import libsbn_amd as L
    old = os.environ.get("MI_PHYLO_FUSED_SETUP")
    try:
        os.environ["MI_PHYLO_FUSED_SETUP"] = "1"
        a = L.Engine(L.PhyloModelSpecification(subst, site, "strict"), tips, w, **kw)
        os.environ["MI_PHYLO_FUSED_SETUP"] = "0"
        b = L.Engine(L.PhyloModelSpecification(subst, site, "strict"), tips, w, **kw)
    finally:
        if old is None:
            os.environ.pop("MI_PHYLO_FUSED_SETUP", None)
        else:
            os.environ["MI_PHYLO_FUSED_SETUP"] = old
    return a, b


def _flat(grads):
    out = [np.array([g.log_likelihood for g in grads])]
    for k in sorted(grads[0].gradient):
        out.append(np.stack([np.atleast_1d(g.gradient[k]) for g in grads]).ravel())
    return np.concatenate(out)


def _ds1(T, seed=5):
    st = O.load_struct("ds1_top100")
    tips, w, pids, _ = O.struct_arrays(st)
    reps = (T + len(pids) - 1) // len(pids)
    pids = np.ascontiguousarray(np.tile(pids, (reps, 1))[:T])
    rng = np.random.default_rng(seed)
    bls = rng.exponential(0.1, size=(T, pids.shape[1] + 1))
    bls[:, -1] = 0
    return tips, w, pids, bls, rng


@pytest.mark.parametrize("T", [1, 125, 500, 1000])
@pytest.mark.parametrize("rescaling", [False, True])
def test_one_launch_is_bit_identical_to_four_launches(T, rescaling):
    tips, w, pids, bls, rng = _ds1(T)
    spec = O.make_spec(27, tips.shape[1], "JC69", "weibull+4", "strict")
    pr = _params(spec, T, **{"Weibull shape": rng.uniform(0.3, 2.0, size=(T, 1))})
    fused, plain = _engines("JC69", "weibull+4", tips, w)
    a = fused.gradients(pids, bls, pr, rescaling)
    # (above 512 trees the engine keeps four launches: the cross-over of DESIGN.md 4.7)
    assert fused.last_call_info() == (FUSED if T <= 512 else PLAIN, T, T)
    b = plain.gradients(pids, bls, pr, rescaling)
    assert plain.last_call_info() == (PLAIN, T, T)
    assert np.array_equal(_flat(a), _flat(b))
    # ... and both agree with the oracle (first and last trees)
    sel = np.unique(np.r_[0:min(T, 3), max(T - 3, 0):T])
    og = O.unrooted_gradients(spec, tips, w, pids[sel], bls[sel], pr[sel], rescaling, 4)
    ll = np.array([a[i].log_likelihood for i in sel])
    assert np.allclose(ll, og["log_likelihood"], rtol=1e-10, atol=0)
    gb = np.stack([a[i].gradient["branch_lengths"] for i in sel])
    assert np.allclose(gb, og["branch_lengths"], rtol=1e-10, atol=1e-10 * np.max(np.abs(gb)))
    fused.close(); plain.close()


def test_gtr_branch_gradient_only_takes_one_launch_and_equals_the_full_call():
    """BASELINE configs[2] as worded -- GTR + 4 categories, log-likelihood and branch-length
    gradient: without the substitution / site outputs the call is one evaluation per tree with
    the tree's own model (no perturbed instances), and takes the one-launch path; what it
    returns is bit-identical to the same blocks of the full finite-difference call."""
    T = 125
    tips, w, pids, bls, rng = _ds1(T, seed=11)
    spec = O.make_spec(27, tips.shape[1], "GTR", "weibull+4", "strict")
    r, f = TU.random_gtr_params(T, rng)
    pr = _params(spec, T, **{"GTR rates": r, "frequencies": f,
                             "Weibull shape": rng.uniform(0.4, 1.5, size=(T, 1))})
    fused, plain = _engines("GTR", "weibull+4", tips, w)
    a = fused.gradients(pids, bls, pr, gradient_blocks=("branch_lengths",))
    assert fused.last_call_info() == (FUSED, T, T)
    b = plain.gradients(pids, bls, pr, gradient_blocks=("branch_lengths",))
    assert plain.last_call_info() == (PLAIN, T, T)
    full = plain.gradients(pids, bls, pr)
    assert plain.last_call_info()[1:] == (18 * T, 2 * T)
    for x, y, z in zip(a, b, full):
        assert x.log_likelihood == y.log_likelihood == z.log_likelihood
        assert np.array_equal(x.gradient["branch_lengths"], y.gradient["branch_lengths"])
        assert np.array_equal(x.gradient["branch_lengths"], z.gradient["branch_lengths"])
    og = O.unrooted_gradients(spec, tips, w, pids[:4], bls[:4], pr[:4], False, 4)
    assert np.allclose([x.log_likelihood for x in a[:4]], og["log_likelihood"], rtol=1e-10, atol=0)
    fused.close(); plain.close()


def _site_params(spec, T, site, rng):
    return _params(spec, T) if site == "constant" else \
        _params(spec, T, **{"Weibull shape": rng.uniform(0.4, 1.5, size=(T, 1))})


@pytest.mark.parametrize("site", ["constant", "weibull+2", "weibull+3", "weibull+4"])
def test_small_and_rooted_trees(site):
    """(round 6: two rate categories take the look-up walk, and with it the one-launch call, as
    well; one category with the vectors in LDS stays with the second generation, whose waves take
    several tiles in a row -- MI_PHYLO_WALK3_K1=1 sends it through the look-up walk too)"""
    if site == "constant":
        os.environ["MI_PHYLO_WALK3_K1"] = "1"
    try:
        _small_and_rooted(site)
    finally:
        os.environ.pop("MI_PHYLO_WALK3_K1", None)


def _small_and_rooted(site):
    rng = np.random.default_rng(77)
    # 5 taxa x 7 patterns (one partial tile), 300 trees: more set-up waves than some CUs hold
    n, P, T = 5, 7, 300
    tips, w = TU.random_alignment(n, P, rng)
    pids, bls = TU.random_trees(n, T, rng, mean_bl=0.07)
    spec = O.make_spec(n, P, "JC69", site, "strict")
    pr = _site_params(spec, T, site, rng)
    fused, plain = _engines("JC69", site, tips, w)
    a, b = fused.gradients(pids, bls, pr), plain.gradients(pids, bls, pr)
    assert fused.last_call_info()[0] == FUSED and plain.last_call_info()[0] == PLAIN
    assert np.array_equal(_flat(a), _flat(b))
    fused.close(); plain.close()
    # rooted, 14 taxa: the height-ratio chain rule follows in reduce_finalize
    n, P, T = 14, 70, 40
    tips, w = TU.random_alignment(n, P, rng)
    trees = [TU.clocklike_rooted_tree(n, rng) for _ in range(T)]
    pids = np.stack([t[0] for t in trees]); bls = np.stack([t[1] for t in trees])
    state = [O.time_tree_init(n, t[0], t[1], t[2]) for t in trees]
    h = np.stack([s[0] for s in state]); bd = np.stack([s[1] for s in state]); ra = np.stack([s[2] for s in state])
    rates = np.full((T, 2 * n - 2), 0.7)
    spec = O.make_spec(n, P, "JC69", site, "strict")
    pr = _site_params(spec, T, site, rng)
    fused, plain = _engines("JC69", site, tips, w)
    a = fused.rooted_gradients(pids, bls, pr, rates, np.ones(T, np.int32), h, bd, ra)
    assert fused.last_call_info()[0] == FUSED
    b = plain.rooted_gradients(pids, bls, pr, rates, np.ones(T, np.int32), h, bd, ra)
    assert plain.last_call_info()[0] == PLAIN
    assert np.array_equal(_flat(a), _flat(b))
    fused.close(); plain.close()


def test_hand_off_words_survive_errors_and_changing_batch_sizes():
    """The per-tree words are zero between calls (reduce_finalize clears them).  A malformed
    tree is an error of the call, as in the four-launch sequence (same message), and leaves
    the engine usable; calls of 125 / 1000 / 1 / 125 trees on one engine reproduce the
    four-launch results bit for bit each time."""
    tips, w, pids, bls, rng = _ds1(1000, seed=3)
    spec = O.make_spec(27, tips.shape[1], "JC69", "weibull+4", "strict")
    pr = _params(spec, 1000, **{"Weibull shape": rng.uniform(0.3, 2.0, size=(1000, 1))})
    fused, plain = _engines("JC69", "weibull+4", tips, w)
    bad = pids[:125].copy()
    bad[60, 5] = 2  # a tip as a parent: not the reference's id form
    msgs = []
    for eng in (fused, plain):
        with pytest.raises(RuntimeError) as err:
            eng.gradients(bad, bls[:125], pr[:125])
        msgs.append(str(err.value))
    assert msgs[0] == msgs[1] and "(tree 60)" in msgs[0]
    for T in (125, 1000, 1, 500, 125):
        a = fused.gradients(pids[:T], bls[:T], pr[:T])
        assert fused.last_call_info() == (FUSED if T <= 512 else PLAIN, T, T)
        b = plain.gradients(pids[:T], bls[:T], pr[:T])
        assert np.array_equal(_flat(a), _flat(b))
    fused.close(); plain.close()


def test_replayed_from_a_graph_under_load():
    """The hand-off inside the kernel, replayed 300 times from a hipGraph with a second engine's
    large batch running beside it on another stream (uneven load, the words re-used every
    replay): every replay's outputs equal the four-launch results."""
    torch = pytest.importorskip("torch")
    T = 125
    tips, w, pids, bls, rng = _ds1(1000, seed=9)
    N = 2 * 27 - 1
    params = np.ones((1000, 2))
    fused, plain = _engines("JC69", "weibull+4", tips, w)
    ref = plain.gradients(pids[:T], bls[:T], params[:T])
    ref_ll = np.array([g.log_likelihood for g in ref])
    ref_g = np.stack([g.gradient["branch_lengths"] for g in ref])
    dev = torch.device("cuda", 0)

    def buffers(Tn):
        return (torch.from_numpy(pids[:Tn]).to(dev), torch.from_numpy(bls[:Tn]).to(dev),
                torch.from_numpy(params[:Tn]).to(dev), torch.zeros(Tn, dtype=torch.float64, device=dev),
                torch.zeros((Tn, N), dtype=torch.float64, device=dev), torch.zeros(Tn, dtype=torch.float64, device=dev))

    d_pid, d_bl, d_par, d_ll, d_g, d_site = buffers(T)
    b_pid, b_bl, b_par, b_ll, b_g, b_site = buffers(1000)
    fused.reserve(T, True)
    plain.reserve(1000, True)
    gs, side = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(gs):
        for _ in range(2):
            fused.gradients_device(gs.cuda_stream, T, d_pid.data_ptr(), d_bl.data_ptr(), d_par.data_ptr(),
                                   d_ll.data_ptr(), d_g.data_ptr(), d_site.data_ptr(), None)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=gs):
        fused.gradients_device(torch.cuda.current_stream().cuda_stream, T, d_pid.data_ptr(), d_bl.data_ptr(),
                               d_par.data_ptr(), d_ll.data_ptr(), d_g.data_ptr(), d_site.data_ptr(), None)
    assert fused.last_call_info()[0] == FUSED
    for rep in range(300):
        if rep % 3 == 0:  # a 1000-tree batch of the other engine beside some of the replays
            plain.gradients_device(side.cuda_stream, 1000, b_pid.data_ptr(), b_bl.data_ptr(), b_par.data_ptr(),
                                   b_ll.data_ptr(), b_g.data_ptr(), b_site.data_ptr(), None)
        d_ll.zero_(); d_g.zero_()
        graph.replay()
        if rep % 10 == 0 or rep > 290:
            torch.cuda.synchronize()
            assert np.array_equal(d_ll.cpu().numpy(), ref_ll), rep
            assert np.array_equal(d_g.cpu().numpy(), ref_g), rep
    torch.cuda.synchronize()
    fused.check_status(); plain.check_status()
    assert np.array_equal(d_ll.cpu().numpy(), ref_ll) and np.array_equal(d_g.cpu().numpy(), ref_g)
    fused.close(); plain.close()


def test_a_time_out_does_not_reach_a_host_pointer_caller(monkeypatch):
    """VERDICT r5 item 6 / ADVICE r5: the walk waves of the one-launch call poll their tree's
    hand-off word for a bounded (wall-clock) time.  Should they ever wait in vain, a caller of
    a host-pointer entry point must not see it -- the reference never fails spuriously
    (/root/reference/src/engine.cpp:54-92): the engine runs the call again through the four-launch
    sequence and returns ITS results.  Forced here with a debug switch that makes one set-up
    wave of tree 2 never report (and a 20 ms budget instead of one second): results equal the
    four-launch engine's bit for bit, the path string says `setup=own-launch`, later calls of
    the engine keep four launches; an input error in the same batch is still reported as such.
    A *_device caller finds the sticky status with its message."""
    torch = pytest.importorskip("torch")
    import libsbn_amd as L
    T = 64
    tips, w, pids, bls, rng = _ds1(T, seed=21)
    spec = O.make_spec(27, tips.shape[1], "JC69", "weibull+4", "strict")
    pr = _params(spec, T, **{"Weibull shape": rng.uniform(0.3, 2.0, size=(T, 1))})
    _, plain = _engines("JC69", "weibull+4", tips, w)
    ref = _flat(plain.gradients(pids, bls, pr))
    monkeypatch.setenv("MI_PHYLO_FUSED_SPIN_MS", "20")  # (read at engine creation)
    monkeypatch.setenv("MI_PHYLO_DEBUG_FUSED_SKIP", "3")
    monkeypatch.setenv("MI_PHYLO_FUSED_SETUP", "1")
    broken = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w)
    other = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w)
    third = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w)
    monkeypatch.delenv("MI_PHYLO_DEBUG_FUSED_SKIP")
    a = broken.gradients(pids, bls, pr)
    assert np.array_equal(_flat(a), ref)
    assert "setup=own-launch" in broken.last_call_path() and broken.last_call_info()[0] == PLAIN
    a = broken.gradients(pids, bls, pr)  # ... and stays with four launches
    assert np.array_equal(_flat(a), ref) and broken.last_call_info()[0] == PLAIN
    # an input error beside the time-out is reported as the input error it is
    bad = pids.copy()
    bad[40, 5] = 2
    with pytest.raises(RuntimeError) as err:
        other.gradients(bad, bls, pr)
    assert "(tree 40)" in str(err.value) and "waited in vain" not in str(err.value)
    a = other.gradients(pids, bls, pr)
    assert np.array_equal(_flat(a), ref) and other.last_call_info()[0] == PLAIN
    # device-pointer caller: sticky status, reported once, with the message
    dev = torch.device("cuda", 0)
    N = 2 * 27 - 1
    d = [torch.from_numpy(x).to(dev) for x in (pids, bls, pr)]
    o_ll = torch.zeros(T, dtype=torch.float64, device=dev)
    o_g = torch.zeros((T, N), dtype=torch.float64, device=dev)
    o_s = torch.zeros(T, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    third.gradients_device(st, T, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), o_ll.data_ptr(),
                           o_g.data_ptr(), o_s.data_ptr(), None)
    assert third.last_call_info()[0] == FUSED
    with pytest.raises(RuntimeError) as err:
        third.check_status(st)
    assert "waited in vain" in str(err.value) and "(tree 2)" in str(err.value)
    third.check_status(st)  # reported once
    third.gradients_device(st, T, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), o_ll.data_ptr(),
                           o_g.data_ptr(), o_s.data_ptr(), None)
    third.check_status(st)
    assert third.last_call_info()[0] == PLAIN
    assert np.array_equal(o_ll.cpu().numpy(), ref[:T])
    for e in (plain, broken, other, third):
        e.close()


def test_hand_off_modes_and_arena_access_forms_are_bit_identical(monkeypatch):
    """The one-launch call's three hand-off modes (MI_PHYLO_FUSED_FENCE=none|l1|agent: DESIGN.md
    4.7), its set-up waves placed on their tree's XCD or in id order, and the arena variant's
    plain / non-temporal accesses (MI_PHYLO_ARENA_NT) change how data travels, never what is
    computed: every form gives the bits of the four-launch sequence / of the other form."""
    import libsbn_amd as L
    T = 125
    tips, w, pids, bls, rng = _ds1(T, seed=31)
    spec = O.make_spec(27, tips.shape[1], "JC69", "weibull+4", "strict")
    pr = _params(spec, T, **{"Weibull shape": rng.uniform(0.3, 2.0, size=(T, 1))})
    _, plain = _engines("JC69", "weibull+4", tips, w)
    ref = _flat(plain.gradients(pids, bls, pr))
    plain.close()
    monkeypatch.setenv("MI_PHYLO_FUSED_SETUP", "1")
    for fence in ("none", "l1", "agent"):
        for colocate in ("1", "0"):
            monkeypatch.setenv("MI_PHYLO_FUSED_FENCE", fence)
            monkeypatch.setenv("MI_PHYLO_FUSED_COLOCATE", colocate)
            eng = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w)
            for _ in range(3):
                got = _flat(eng.gradients(pids, bls, pr))
                assert eng.last_call_info()[0] == FUSED
                assert np.array_equal(got, ref), (fence, colocate)
            eng.close()
    monkeypatch.delenv("MI_PHYLO_FUSED_FENCE")
    monkeypatch.delenv("MI_PHYLO_FUSED_COLOCATE")
    # the arena variant, plain and non-temporal accesses (the launcher chooses by tiles per tree)
    rng = np.random.default_rng(64)
    n, P, T = 50, 60, 300
    tips, w = TU.random_alignment(n, P, rng, gap_fraction=0.05)
    pids, bls = TU.random_trees(n, T, rng, mean_bl=0.06)
    spec = O.make_spec(n, P, "JC69", "weibull+4", "strict")
    pr = _params(spec, T, **{"Weibull shape": rng.uniform(0.4, 1.5, size=(T, 1))})
    # (default tile width: an engine of this shape gets wide tiles -- and then takes the arena for
    # every call -- since round 6; both widths run both access forms)
    got = {}
    for regs in ("3", "4"):
        monkeypatch.setenv("MI_PHYLO_WALK_TILE_REGS", regs)
        monkeypatch.setenv("MI_PHYLO_GRADIENT_STORE", "arena")
        eng = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w)
        for nt in ("0", "1"):
            monkeypatch.setenv("MI_PHYLO_ARENA_NT", nt)  # (read by the launcher at every launch)
            got[regs, nt] = _flat(eng.gradients(pids, bls, pr))
            assert "store=arena" in eng.last_call_path()
            assert ("tile=wide" in eng.last_call_path()) == (regs == "4")
        monkeypatch.delenv("MI_PHYLO_ARENA_NT")
        eng.close()
    monkeypatch.setenv("MI_PHYLO_WALK_TILE_REGS", "3")
    monkeypatch.setenv("MI_PHYLO_GRADIENT_STORE", "lds")
    eng = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w)
    lds = _flat(eng.gradients(pids, bls, pr))
    assert "store=lds" in eng.last_call_path()
    eng.close()
    assert np.array_equal(got["3", "0"], lds) and np.array_equal(got["3", "1"], lds)
    assert np.array_equal(got["4", "0"], got["4", "1"])
    assert np.allclose(got["4", "0"], lds, rtol=1e-11, atol=1e-12 * np.max(np.abs(lds)))


def test_setup_and_records_in_one_launch_for_large_batches(monkeypatch):
    """Beyond the one-launch call's 512 trees the same set-up waves can run as ONE launch in front
    of the walk's (round 6: launch_setup_records, MI_PHYLO_SETUP_RECORDS=1, read per call; measured
    level with the two launches it replaces and not the default) instead of the tree set-up and
    the record launches.  Same trees, model instances and operand records: results bit-identical
    to the four-launch sequence -- two to four categories,
    rescaling, GTR with the branch-length gradient only, an input error reported with its tree."""
    import libsbn_amd as L
    T = 700
    tips, w, pids, bls, rng = _ds1(T, seed=31)
    for subst, site in (("JC69", "weibull+4"), ("JC69", "weibull+3"), ("JC69", "weibull+2"), ("GTR", "weibull+4")):
        spec = O.make_spec(27, tips.shape[1], subst, site, "strict")
        blocks = {"Weibull shape": rng.uniform(0.3, 2.0, size=(T, 1))}
        if subst == "GTR":
            r, f = TU.random_gtr_params(T, rng)
            blocks["GTR rates"] = r
            blocks["frequencies"] = f
        pr = _params(spec, T, **blocks)
        eng = L.Engine(L.PhyloModelSpecification(subst, site, "strict"), tips, w)
        only = ("branch_lengths",) if subst == "GTR" else None
        got = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("MI_PHYLO_SETUP_RECORDS", mode)
            res = []
            for resc in (False, True):
                res.append(_flat(eng.gradients(pids, bls, pr, resc, gradient_blocks=only)))
                assert eng.last_call_info()[0] == PLAIN
                assert ("setup=with-records" in eng.last_call_path()) == (mode == "1"), eng.last_call_path()
            got[mode] = np.concatenate(res)
        monkeypatch.delenv("MI_PHYLO_SETUP_RECORDS")
        assert np.isfinite(got["1"]).all() and np.array_equal(got["1"], got["0"]), (subst, site)
        if subst == "JC69" and site == "weibull+4":
            monkeypatch.setenv("MI_PHYLO_SETUP_RECORDS", "1")
            bad = pids.copy()
            bad[640, 5] = 2
            with pytest.raises(RuntimeError) as err:
                eng.gradients(bad, bls, pr)
            assert "(tree 640)" in str(err.value)
            assert np.array_equal(_flat(eng.gradients(pids, bls, pr)), got["1"][:len(got["1"]) // 2])
            monkeypatch.delenv("MI_PHYLO_SETUP_RECORDS")
        eng.close()
