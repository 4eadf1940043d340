"""Synthetic inputs for parity tests: random topologies numbered exactly as the
reference numbers them (leaves keep ids, internal nodes in post-order with
children ordered by max leaf id -- node.cpp:32-59,341-357)."""
import numpy as np


def _polish(tree, n):
    """tree: nested tuples / ints. Returns parent-id vector (root excluded)."""
    parent = {}
    next_id = [n]

    def maxleaf(t):
        return t if isinstance(t, int) else max(maxleaf(c) for c in t)

    def visit(t):
        if isinstance(t, int):
            return t
        kids = sorted(t, key=maxleaf)
        ids = [visit(c) for c in kids]
        me = next_id[0]
        next_id[0] += 1
        for c in ids:
            parent[c] = me
        return me

    root = visit(tree)
    return np.array([parent[v] for v in range(root)], dtype=np.int32)


def random_topology(n, rng, rooted=False):
    """Uniform random-join topology on n leaves; unrooted = trifurcation at the root."""
    parts = list(range(n))
    target = 1 if rooted else 3
    while len(parts) > target:
        i, j = sorted(rng.choice(len(parts), size=2, replace=False))
        b = parts.pop(j)
        a = parts.pop(i)
        parts.append((a, b))
    tree = tuple(parts) if not rooted else parts[0]
    return _polish(tree, n)


def ladder_topology(n, rooted=False):
    t = 0
    if rooted:
        for i in range(1, n):
            t = (i, t)
        return _polish(t, n)
    for i in range(1, n - 2):
        t = (i, t)
    return _polish((t, n - 2, n - 1), n)


def balanced_topology(n, rooted=False):
    """Most balanced tree on n leaves (pairs joined level by level): the worst case for
    the number of simultaneously live partial-likelihood vectors of a post-order walk."""
    parts = list(range(n))
    target = 1 if rooted else 3
    while len(parts) > target:
        nxt = [(parts[i], parts[i + 1]) for i in range(0, len(parts) - 1, 2)]
        if len(parts) % 2:
            nxt.append(parts[-1])
        if len(nxt) < target:  # unrooted: stop at the trifurcation
            nxt = [parts[0], parts[1]] + ([tuple(parts[2:])] if len(parts) > 3 else parts[2:])
        parts = nxt
    tree = tuple(parts) if not rooted else parts[0]
    return _polish(tree, n)


def random_trees(n, T, rng, rooted=False, mean_bl=0.1):
    nodes = 2 * n - 1 if rooted else 2 * n - 2
    pids = np.stack([random_topology(n, rng, rooted) for _ in range(T)])
    bls = rng.exponential(mean_bl, size=(T, nodes))
    bls[:, -1] = 0.0
    return pids, bls


def random_alignment(n, P, rng, gap_fraction=0.05):
    """Random tip states 0..3 with some gaps (code 4); integer-valued weights."""
    tips = rng.integers(0, 4, size=(n, P)).astype(np.int32)
    tips[rng.random((n, P)) < gap_fraction] = 4
    weights = rng.integers(1, 6, size=P).astype(np.float64)
    return tips, weights


def random_gtr_params(T, rng):
    rates = rng.dirichlet(10 * np.ones(6), size=T)
    freqs = rng.dirichlet(10 * np.ones(4), size=T)
    return rates, freqs


def clocklike_rooted_tree(n, rng):
    """Rooted topology + tip dates + branch lengths consistent with node heights."""
    pid = random_topology(n, rng, rooted=True)
    N = 2 * n - 1
    dates = np.round(rng.uniform(0, 3, size=n), 3)
    dates[rng.integers(n)] = 0.0
    h = np.zeros(N)
    h[:n] = dates
    kids = {}
    for v, p in enumerate(pid):
        kids.setdefault(int(p), []).append(v)
    for v in range(n, N):
        h[v] = max(h[c] for c in kids[v]) + rng.uniform(0.05, 1.0)
    bl = np.zeros(N)
    for v, p in enumerate(pid):
        bl[v] = h[p] - h[v]
    return pid, bl, dates
