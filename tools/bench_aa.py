#!/usr/bin/env python3
"""S-WAG of SURVEY.md 8(d) (BASELINE.json configs[4]): synthetic 20-state alignment,
n taxa x P patterns x K categories, T trees; times log_likelihoods and phylo_gradients of
the 20-state engine with inputs resident in HBM.

  python tools/bench_aa.py [--taxa 512] [--patterns 50000] [--trees 1] [--steps 5]
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np  # noqa: E402


def random_topology(n, rng):
    """uniform random-join unrooted topology, ids as the reference numbers them"""
    import tree_utils as TU
    return TU.random_topology(n, rng)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--taxa", type=int, default=512)
    ap.add_argument("--patterns", type=int, default=50000)
    ap.add_argument("--categories", type=int, default=4)
    ap.add_argument("--trees", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mode", choices=["both", "gradient", "loglik"], default="both")
    args = ap.parse_args()
    import torch
    import libsbn_amd as L
    sys.setrecursionlimit(100000)
    n, P, K, T = args.taxa, args.patterns, args.categories, args.trees
    N = 2 * n - 1
    rng = np.random.default_rng(47)
    tips = rng.integers(0, 20, size=(n, P)).astype(np.int32)
    w = np.ones(P)
    pids = np.stack([random_topology(n, rng) for _ in range(T)])
    bls = rng.exponential(0.1, size=(T, 2 * n - 2))
    bls[:, -1] = 0
    params = np.ones((T, 2))
    dev = torch.device("cuda", 0)
    eng = L.Engine(L.PhyloModelSpecification("WAG", f"weibull+{K}", "strict"), tips, w, device=0)
    d_pid = torch.from_numpy(pids).to(dev)
    d_bl = torch.from_numpy(bls).to(dev)
    d_par = torch.from_numpy(params).to(dev)
    d_ll = torch.empty(T, dtype=torch.float64, device=dev)
    d_g = torch.empty((T, N), dtype=torch.float64, device=dev)
    d_site = torch.empty(T, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    plv = K * P * 20 * 8
    out = {"taxa": n, "patterns": P, "categories": K, "trees": T}
    for mode in (["loglik", "gradient"] if args.mode == "both" else [args.mode]):
        grad = mode == "gradient"
        eng.reserve(T, grad)

        def step():
            if grad:
                eng.gradients_device(stream, T, d_pid.data_ptr(), d_bl.data_ptr(),
                                     d_par.data_ptr(), d_ll.data_ptr(), d_g.data_ptr(),
                                     d_site.data_ptr(), None)
            else:
                eng.log_likelihoods_device(stream, T, d_pid.data_ptr(), d_bl.data_ptr(),
                                           d_par.data_ptr(), d_ll.data_ptr())
        for _ in range(args.warmup):
            step()
        eng.check_status(stream)
        torch.cuda.synchronize()
        eng.profile_begin(args.steps)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        walk_ms = float(np.mean(eng.profile_collect(args.steps)))
        eng.check_status(stream)
        assert bool(torch.isfinite(d_ll).all())
        # SURVEY 8(d) models and what the kernels actually have to move / compute
        b_model = ((10 * n - 14) if grad else 2 * (n - 1)) * plv + (12 if grad else 4) * n * P
        # 20x20 products per (pattern, category): one per internal edge in the post-order pass,
        # two more in the pre-order pass, a third there for a node with two internal children
        kids = np.bincount(pids[0][n:][pids[0][n:] >= 0], minlength=pids.shape[1] + 1)
        mv = (3 * (n - 2) + int(np.maximum(kids - 1, 0).sum())) if grad else (n - 2)
        flops = mv * K * P * 800.0
        out[mode] = {"ms_per_step": 1e3 * dt, "trees_per_s": T / dt, "walk_ms": walk_ms,
                     "logL0": float(d_ll[0]),
                     "survey_model_GBps": b_model * T / (walk_ms * 1e-3) / 1e9,
                     "mfma_TFLOPs": flops * T / (walk_ms * 1e-3) / 1e12}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
