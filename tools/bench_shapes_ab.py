"""A/B over the standard alignment shapes (DS1-DS8, 100 x 500, fluA-sized one-category trees):
ms per `phylo_gradients` call of T random trees with the engine's default path against forced
alternatives (environment switches read at engine creation), interleaved in one process on one
box.  Prints, per shape and variant, ms per call, 1e9 taxa x patterns x trees / s, the fraction
of the measured FP64 matrix peak (71.1 TFLOP/s) that the SURVEY 8(d) flop count amounts to, and
the path the call took.  Every variant's log-likelihoods and gradients are compared with the
first variant's (bit-identical paths print `same`).
  python tools/bench_shapes_ab.py [--trees 1000] [--shapes 27x934x4,...] name[=ENV=val[,ENV=val]] ...
(profiles/r06_tree_size.txt is made with it.)"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch

import libsbn_amd as L
import tree_utils as TU

DEFAULT_SHAPES = "27x934x4,29x1195x4,36x1812x4,41x1137x4,50x378x4,50x1133x4,59x1824x4,64x1008x4,100x500x4,69x238x1,45x1000x1,45x1000x2,31x1000x4"
SWITCHES = ("MI_PHYLO_GRADIENT_WALK", "MI_PHYLO_GRADIENT_STORE", "MI_PHYLO_WALK3_ARENA", "MI_PHYLO_FUSED_SETUP",
            "MI_PHYLO_WALK3_K1", "MI_PHYLO_ARENA_NT", "MI_PHYLO_WALK_TILE_REGS", "MI_PHYLO_MACRO_SLOTS", "MI_PHYLO_TIP_TILES", "MI_PHYLO_FUSED_FENCE", "MI_PHYLO_FUSED_COLOCATE")
PEAK = 71.1e12
dev = torch.device("cuda", 0)


def run(n, P, K, T, env, reps):
    for k in SWITCHES:
        os.environ.pop(k, None)
    os.environ.update(env)
    site = "constant" if K == 1 else f"weibull+{K}"
    tips, w = TU.random_alignment(n, P, np.random.default_rng(n * 1000 + P))
    pids, bls = TU.random_trees(n, min(T, 50), np.random.default_rng(n + P), mean_bl=0.05)
    r = (T + len(pids) - 1) // len(pids)
    pids = np.ascontiguousarray(np.tile(pids, (r, 1))[:T]).astype(np.int32)
    bls = np.ascontiguousarray(np.tile(bls, (r, 1))[:T] * np.random.default_rng(5).uniform(0.5, 1.5, size=(T, 1)))
    N = 2 * n - 1
    eng = L.Engine(L.PhyloModelSpecification("JC69", site, "strict"), tips, w, device=0)
    params = np.ones((T, max(eng.param_count, 1)))
    d = [torch.from_numpy(x).to(dev) for x in (pids, bls, params)]
    d_ll = torch.zeros(T, dtype=torch.float64, device=dev)
    d_g = torch.zeros((T, N), dtype=torch.float64, device=dev)
    d_s = torch.zeros(T, dtype=torch.float64, device=dev)
    cs = torch.cuda.Stream()
    st = cs.cuda_stream

    def step():
        eng.gradients_device(st, T, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d_ll.data_ptr(),
                             d_g.data_ptr(), d_s.data_ptr() if K > 1 else None, None)

    with torch.cuda.stream(cs):
        for _ in range(2):
            step()
        eng.check_status(st)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cs)
            for _ in range(reps):
                step()
            e1.record(cs)
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps)
    path = eng.last_call_path()
    out = np.concatenate([d_ll.cpu().numpy(), d_g.cpu().numpy().ravel(), d_s.cpu().numpy()])
    eng.close()
    return best, path, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=1000)
    ap.add_argument("--shapes", default=DEFAULT_SHAPES)
    ap.add_argument("variants", nargs="*", default=["default"])
    a = ap.parse_args()
    variants = []
    for v in a.variants:
        name, _, rest = v.partition("=")
        env = dict(kv.split("=", 1) for kv in rest.split(",")) if rest else {}
        variants.append((name, env))
    T = a.trees
    print("# %d trees per call, JC69 + (constant | weibull+K); flops per tree F_G = 288 K P (n - 1) (DESIGN.md 4: 27.98 MF for DS1)" % T)
    for shape in a.shapes.split(","):
        n, P, K = (int(x) for x in shape.split("x"))
        flops = 288.0 * K * P * (n - 1) * T
        ref = None
        for name, env in variants:
            try:
                ms, path, out = run(n, P, K, T, env, 10 if T <= 100 else 4)
            except RuntimeError as exc:
                print("%-14s %-12s FAILED %s" % (shape, name, str(exc)[:100]))
                continue
            same = "ref" if ref is None else ("same" if np.array_equal(ref, out) else "maxdiff %.2e" % np.max(np.abs(ref - out)))
            if ref is None:
                ref = out
            print("%-14s %-12s %8.4f ms  %6.2f G units/s  frac %.3f  %-8s %s" % (
                shape, name, ms, n * P * T / ms / 1e6, flops / (ms * 1e-3) / PEAK, same, path), flush=True)


if __name__ == "__main__":
    main()
