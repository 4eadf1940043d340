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


def test_shard_arithmetic():
    from libsbn_amd import sharding as S
    for T in (1, 7, 8, 1000, 1001):
        for G in (1, 2, 3, 8):
            blocks = [S.tree_shard(T, r, G) for r in range(G)]
            assert blocks[0][0] == 0 and blocks[-1][1] == T
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(G - 1))
            sizes = S.shard_sizes(T, G)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == T
