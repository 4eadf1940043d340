"""Tree-batch sharding across the GPUs of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in CPU tests).

The reference parallelises over trees with a thread pool (FatBeagleParallelize,
src/fat_beagle.hpp:119-149, TaskProcessor src/task_processor.hpp:43-112): trees are
independent units, results come back in tree order.  Here every rank evaluates a
contiguous block of the batch on its own GPU and ONE collective per engine call
(all_gather of the packed per-tree results) rebuilds the per-tree vectors the
Engine API returns.  No other data-path communication exists.
"""
import numpy as np


def tree_shard(tree_count, rank, world_size):
    """Contiguous block [lo, hi) of rank `rank`: sizes differ by <= 1."""
    base, extra = divmod(tree_count, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(tree_count, world_size):
    return [tree_shard(tree_count, r, world_size)[1] - tree_shard(tree_count, r, world_size)[0]
            for r in range(world_size)]


def pack_results(log_likelihoods, columns):
    """[T] + list of [T, c_i] -> [T, 1 + sum c_i] (one buffer -> one collective)."""
    import torch
    parts = [log_likelihoods.reshape(-1, 1)] + [c.reshape(len(log_likelihoods), -1)
                                                for c in columns]
    return torch.cat(parts, dim=1).contiguous()


class GatheredTrees:
    """Handle of one all_gather_trees call.  `result()` waits for the collective (on the
    current stream) and returns the [tree_count, C] rows in tree order."""

    def __init__(self, work, out, keep, sizes, width):
        self._work, self._out, self._keep = work, out, keep
        self._sizes, self._width = sizes, width

    def result(self):
        import torch
        if self._work is not None:
            self._work.wait()
            self._work = None
        self._keep = None
        out, sizes, width = self._out, self._sizes, self._width
        if all(s == width for s in sizes):
            return out
        rows = [out[r * width:r * width + sizes[r]] for r in range(len(sizes))]
        return torch.cat(rows, dim=0)


def all_gather_trees(local, tree_count, group=None, async_op=False):
    """All-gather per-tree rows of every rank, back into tree order.

    local: [T_local, C] tensor of this rank (rows of its tree_shard).  Uneven shards
    are padded to the largest one so that a single all_gather_into_tensor is enough.
    With async_op=True the collective is only enqueued (RCCL runs it on its own stream,
    so it overlaps whatever the caller launches next) and a GatheredTrees handle is
    returned; otherwise the gathered tensor.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = shard_sizes(tree_count, world)
    width = max(sizes)
    C = local.shape[1]
    padded = local
    if local.shape[0] < width:
        padded = torch.zeros((width, C), dtype=local.dtype, device=local.device)
        padded[:local.shape[0]] = local
    padded = padded.contiguous()
    out = torch.empty((world * width, C), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, padded, group=group, async_op=async_op)
    handle = GatheredTrees(work if async_op else None, out, padded, sizes, width)
    return handle if async_op else handle.result()


class ResultBlocks:
    """The per-tree results of one rank as ONE buffer the engine writes into directly:
    [logL (T) | `extra` scalar columns (T each) | branch gradient (T x N)], each block
    contiguous (what the C ABI's output pointers want) -- so the collective needs no packing
    copy.  `log_likelihoods`, `extras[i]`, `branch_gradients` are views of `buffer`."""

    def __init__(self, tree_count, node_count, extra=0, device=None, dtype=None):
        import torch
        dtype = dtype or torch.float64
        T, N = tree_count, node_count
        self.tree_count, self.node_count, self.extra = T, N, extra
        self.buffer = torch.zeros(T * (1 + extra + N), dtype=dtype, device=device)
        self.log_likelihoods = self.buffer[:T]
        self.extras = [self.buffer[(1 + i) * T:(2 + i) * T] for i in range(extra)]
        self.branch_gradients = self.buffer[(1 + extra) * T:].view(T, N)


def all_gather_result_blocks(blocks, out=None, group=None, async_op=False):
    """All-gather the ResultBlocks buffers of all ranks (equal tree counts per rank) into
    `out` [world, T (1 + extra + N)] -- one collective, no packing.  Returns (out, work);
    `work` is None unless async_op.  gathered_views(out, blocks) gives tree-order views."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world, blocks.buffer.numel()), dtype=blocks.buffer.dtype,
                          device=blocks.buffer.device)
    # (the flat view is the concatenated form every backend accepts)
    work = dist.all_gather_into_tensor(out.view(-1), blocks.buffer, group=group,
                                       async_op=async_op)
    return out, (work if async_op else None)


def gathered_views(out, blocks):
    """Tree-order results from all_gather_result_blocks' output: per rank r the trees
    [r T, (r + 1) T).  Returns (logL [world, T], extras [world, T] each, branch gradients
    [world, T, N]) as views of `out` (rank-major = tree order; reshape(-1, ...) to flatten)."""
    T, N, X = blocks.tree_count, blocks.node_count, blocks.extra
    ll = out[:, :T]
    extras = [out[:, (1 + i) * T:(2 + i) * T] for i in range(X)]
    g = out[:, (1 + X) * T:].view(out.shape[0], T, N)
    return ll, extras, g


def all_reduce_step_terms(log_likelihoods, branch_gradients, branch_index, parameter_count,
                          tree_weights=None, group=None):
    """The caller-side reductions of one variational-inference step folded into ONE
    all-reduce (north_star: "a single RCCL all-reduce over xGMI for
    the ELBO/gradient sum").

    vip sums the per-tree log-likelihoods (vip/burrito.py:143-153) and scatter-adds each
    tree's branch gradient into the shared branch-parameter vector by split index
    (vip/branch_model.py:125-132).  Every rank does that for its own trees on its GPU;
    the packed [1 + parameter_count] vector is then summed over ranks.

      log_likelihoods  [T_local]       this rank's trees
      branch_gradients [T_local, N]    d logL / d branch length, node-id order
      branch_index     [T_local, N]    int64 index of each branch's parameter (split
                                       index); negative entries are skipped (root, the
                                       two trailing zeros of the unrooted gradient)
      tree_weights     [T_local] or None   per-tree multipliers (e.g. VIMCO weights)
    Returns (sum of weighted log-likelihoods, [parameter_count] gradient), identical on
    every rank.
    """
    import torch
    import torch.distributed as dist
    ll = log_likelihoods.reshape(-1)
    g = branch_gradients.reshape(len(ll), -1)
    if tree_weights is not None:
        wt = tree_weights.reshape(-1).to(ll.dtype)
        ll = ll * wt
        g = g * wt[:, None]
    packed = torch.zeros(1 + parameter_count, dtype=ll.dtype, device=ll.device)
    packed[0] = ll.sum()
    idx = branch_index.reshape(-1).to(torch.int64)
    keep = idx >= 0
    packed[1:].index_add_(0, idx[keep], g.reshape(-1)[keep])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    return packed[0], packed[1:]


def pattern_shard(pattern_count, rank, world_size):
    """Contiguous block [lo, hi) of site patterns of rank `rank` (second way:
    few trees x very long alignments).  Every rank builds its Engine from
    tips[:, lo:hi] / weights[lo:hi], evaluates ALL trees with the same parameters, and the
    per-tree results are summed over ranks (all_reduce_pattern_shards)."""
    return tree_shard(pattern_count, rank, world_size)


def all_reduce_pattern_shards(packed, group=None):
    """Sum of the per-tree results of pattern shards: ONE all-reduce.

    The log-likelihood and every gradient block of the unrooted path are sums over site
    patterns (rescaling is per pattern, the finite-difference substitution gradient is
    linear in log-likelihoods), so `packed` = pack_results(logL, [gradient blocks...]) of
    this rank's pattern block summed over ranks is the result for the whole alignment.
    (Rooted trees: the log-det-Jacobian of LogLikelihood(RootedTree) is a per-tree
    constant, not a pattern sum -- shard the unrooted quantities and add it once.)
    In place; returns `packed`.
    """
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    return packed


class ShardedBatch:
    """Splits the inputs of one Engine call by tree and reassembles the outputs.

    `compute(lo, hi)` is whatever evaluates trees [lo, hi) on this rank (the HIP
    engine in production) and returns a [hi-lo, C] tensor of packed per-tree results.
    """

    def __init__(self, tree_count, group=None):
        import torch.distributed as dist
        self.tree_count = tree_count
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.lo, self.hi = tree_shard(tree_count, self.rank, self.world)

    def run(self, compute):
        local = compute(self.lo, self.hi)
        if self.world == 1:
            return local
        return all_gather_trees(local, self.tree_count, self.group)


def numpy_shard(arrays, lo, hi):
    return [np.ascontiguousarray(a[lo:hi]) for a in arrays]
