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
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), gathered.numpy())
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


def test_shard_arithmetic():
    from libsbn_amd import sharding as S
    for T in (1, 7, 8, 1000, 1001):
        for G in (1, 2, 3, 8):
            blocks = [S.tree_shard(T, r, G) for r in range(G)]
            assert blocks[0][0] == 0 and blocks[-1][1] == T
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(G - 1))
            sizes = S.shard_sizes(T, G)
            assert max(sizes) - min(sizes) <= 1 and sum(sizes) == T
