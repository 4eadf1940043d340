"""N > 1 path on CPU: two processes, gloo backend.  Tree-sharded evaluation +
the single all_gather must reproduce the unsharded per-tree results in tree order
(even and uneven shard sizes).  The per-rank compute stand-in is the CPU oracle --
the test exercises the sharding/collective logic, which is the same code bench.py
and a multi-GPU caller use (libsbn_amd/sharding.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _split_index(T):
    """Deterministic stand-in for the split indexer: branch v of tree t -> parameter
    (7 t + 3 v) mod 40; the root and the two trailing entries are skipped."""
    idx = (7 * np.arange(T)[:, None] + 3 * np.arange(53)[None, :]) % 40
    idx[:, 51:] = -1
    return idx.astype(np.int64)


def _worker(rank, world, port, T, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libsbn_amd import sharding as S
    st = O.load_struct("ds1_sub10")
    tips, w, pids, bls = O.struct_arrays(st)
    pids, bls = pids[:T], bls[:T]
    spec = O.make_spec(27, 934, "JC69", "weibull+4")
    pr = np.zeros((T, 2))
    pr[:, 0] = 0.7
    pr[:, 1] = 1.0

    def compute(lo, hi):
        if hi == lo:
            return torch.zeros((0, 2 + 53), dtype=torch.float64)
        g = O.unrooted_gradients(spec, tips, w, pids[lo:hi], bls[lo:hi], pr[lo:hi])
        return S.pack_results(torch.from_numpy(g["log_likelihood"]),
                              [torch.from_numpy(g["site_model"]),
                               torch.from_numpy(g["branch_lengths"])])

    batch = S.ShardedBatch(T)
    assert (batch.lo, batch.hi) == S.tree_shard(T, rank, world)
    gathered = batch.run(compute)
    assert gathered.shape == (T, 55)
    # the asynchronous form (what bench.py overlaps with the next step) gives the same rows
    handle = S.all_gather_trees(compute(batch.lo, batch.hi), T, async_op=True)
    assert torch.equal(handle.result(), gathered)
    # the zero-copy form bench.py uses: the "engine" writes straight into one buffer per rank
    # (equal shards only)
    if T % world == 0:
        Tl = T // world
        blocks = S.ResultBlocks(Tl, 53, extra=1)
        mine = compute(batch.lo, batch.hi)
        blocks.log_likelihoods.copy_(mine[:, 0])
        blocks.extras[0].copy_(mine[:, 1])
        blocks.branch_gradients.copy_(mine[:, 2:])
        out, work = S.all_gather_result_blocks(blocks, async_op=True)
        work.wait()
        ll_all, extras_all, g_all = S.gathered_views(out, blocks)
        assert torch.equal(ll_all.reshape(-1), gathered[:, 0])
        assert torch.equal(extras_all[0].reshape(-1), gathered[:, 1])
        assert torch.equal(g_all.reshape(T, 53), gathered[:, 2:])
    # the fused step reduction: sum of log-likelihoods + scatter-add of branch gradients by
    # (here: synthetic) split index, one all-reduce
    local = compute(batch.lo, batch.hi)
    idx = _split_index(T)[batch.lo:batch.hi]
    wts = torch.linspace(0.5, 1.5, T, dtype=torch.float64)[batch.lo:batch.hi]
    total, grad = S.all_reduce_step_terms(local[:, 0], local[:, 2:], torch.from_numpy(idx), 40,
                                          tree_weights=wts)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), gathered.numpy())
        np.save(os.path.join(out_dir, "reduced.npy"),
                np.concatenate([[float(total)], grad.numpy()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T", [4, 5])
def test_two_rank_tree_sharding_matches_unsharded(T, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, T, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "gathered.npy")
    st = O.load_struct("ds1_sub10")
    tips, w, pids, bls = O.struct_arrays(st)
    spec = O.make_spec(27, 934, "JC69", "weibull+4")
    pr = np.zeros((T, 2))
    pr[:, 0] = 0.7
    pr[:, 1] = 1.0
    g = O.unrooted_gradients(spec, tips, w, pids[:T], bls[:T], pr)
    assert np.array_equal(got[:, 0], g["log_likelihood"])
    assert np.array_equal(got[:, 1], g["site_model"])
    assert np.array_equal(got[:, 2:], g["branch_lengths"])
    # fused reduction == the same sums done on the unsharded results
    red = np.load(tmp_path / "reduced.npy")
    wts = np.linspace(0.5, 1.5, T)
    idx = _split_index(T)
    want = np.zeros(40)
    for t in range(T):
        for v in range(53):
            if idx[t, v] >= 0:
                want[idx[t, v]] += wts[t] * g["branch_lengths"][t, v]
    assert abs(red[0] - float(np.sum(wts * g["log_likelihood"]))) <= 1e-12 * abs(red[0])
    assert np.allclose(red[1:], want, rtol=1e-12, atol=1e-12)


def _pattern_worker(rank, world, port, out_dir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from libsbn_amd import sharding as S
    st = O.load_struct("ds1_sub10")
    tips, w, pids, bls = O.struct_arrays(st)
    T = 3
    lo, hi = S.pattern_shard(tips.shape[1], rank, world)
    spec = O.make_spec(27, hi - lo, "GTR", "weibull+4")
    pr = _gtr_params(T)
    g = O.unrooted_gradients(spec, np.ascontiguousarray(tips[:, lo:hi]),
                             np.ascontiguousarray(w[lo:hi]), pids[:T], bls[:T], pr, True)
    packed = S.pack_results(torch.from_numpy(g["log_likelihood"]),
                            [torch.from_numpy(g["site_model"]),
                             torch.from_numpy(g["substitution_model"]),
                             torch.from_numpy(g["branch_lengths"])])
    total = S.all_reduce_pattern_shards(packed)
    if rank == 0:
        np.save(os.path.join(out_dir, "summed.npy"), total.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _gtr_params(T):
    spec = O.make_spec(27, 934, "GTR", "weibull+4")
    lay = O.param_layout(spec)
    pr = np.zeros((T, O.param_count(spec)))
    pr[:, lay["GTR rates"]:lay["GTR rates"] + 6] = [0.05, 0.1, 0.15, 0.2, 0.25, 0.25]
    pr[:, lay["frequencies"]:lay["frequencies"] + 4] = [0.1, 0.2, 0.3, 0.4]
    pr[:, lay["Weibull shape"]] = 0.7
    return pr


def test_two_rank_pattern_sharding_sums_to_unsharded(tmp_path):
    """SURVEY 8e, second way: ranks own contiguous blocks of site patterns, evaluate all
    trees, one all-reduce(sum).  Log-likelihood, branch, site-model and (finite-difference)
    substitution-model gradients of the whole alignment are recovered (rescaling on)."""
    port = _free_port()
    mp.spawn(_pattern_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "summed.npy")
    st = O.load_struct("ds1_sub10")
    tips, w, pids, bls = O.struct_arrays(st)
    T = 3
    spec = O.make_spec(27, 934, "GTR", "weibull+4")
    g = O.unrooted_gradients(spec, tips, w, pids[:T], bls[:T], _gtr_params(T), True)
    want = np.concatenate([g["log_likelihood"][:, None], g["site_model"].reshape(T, -1),
                           g["substitution_model"].reshape(T, -1), g["branch_lengths"]], axis=1)
    assert got.shape == want.shape
    # sums are taken in a different order; the substitution block is a finite difference
    # of log-likelihoods (1e-6 steps), so its rounding noise is ~1e-16 * |logL| / 1e-6
    assert np.allclose(got[:, :2], want[:, :2], rtol=1e-11, atol=1e-9)
    assert np.allclose(got[:, 10:], want[:, 10:], rtol=1e-11, atol=1e-9)
    assert np.allclose(got[:, 2:10], want[:, 2:10], rtol=1e-5, atol=1e-4)


def test_shard_arithmetic():
    from libsbn_amd import sharding as S
    for T in (1, 7, 8, 1000, 1001):
        for G in (1, 2, 3, 8):
            blocks = [S.tree_shard(T, r, G) for r in range(G)]
            assert blocks[0][0] == 0 and blocks[-1][1] == T
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(G - 1))
            sizes = S.shard_sizes(T, G)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == T
