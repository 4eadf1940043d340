"""The small (strong-scaling) step under switches: DS1 trees, `phylo_gradients` with the site
gradient (JC69 + weibull+4), replayed from a hipGraph -- 200 replays between two events, best of
five -- for every variant given (environment switches read at engine creation), interleaved in
one process on one box; outputs compared with the first variant's.
  python tools/bench_small_step.py [--trees 125] name[=ENV=val[,ENV=val]] ...
(the hand-off fences of the one-launch call, round 6.)"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import torch

import bench as B
import libsbn_amd as L

SWITCHES = ("MI_PHYLO_FUSED_SETUP", "MI_PHYLO_FUSED_FENCE", "MI_PHYLO_FUSED_COLOCATE", "MI_PHYLO_GRADIENT_WALK")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=125)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("variants", nargs="*", default=["default"])
    a = ap.parse_args()
    T = a.trees
    tips, w, pids, bls = B.ds1_workload(T)
    n, P = tips.shape
    N = 2 * n - 1
    dev = torch.device("cuda", 0)
    params = np.ones((T, 2))
    d_pid = torch.from_numpy(pids).to(dev)
    d_bl = torch.from_numpy(bls).to(dev)
    d_par = torch.from_numpy(params).to(dev)
    d_ll = torch.zeros(T, dtype=torch.float64, device=dev)
    d_g = torch.zeros((T, N), dtype=torch.float64, device=dev)
    d_s = torch.zeros(T, dtype=torch.float64, device=dev)
    variants = []
    for v in a.variants:
        name, _, rest = v.partition("=")
        variants.append((name, dict(kv.split("=", 1) for kv in rest.split(",")) if rest else {}))
    res = {name: [] for name, _ in variants}
    ref = None
    for rnd in range(a.rounds):
        for name, env in variants:
            for k in SWITCHES:
                os.environ.pop(k, None)
            os.environ.update(env)
            eng = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w, device=0)
            eng.reserve(T, True)
            gs = torch.cuda.Stream()

            def step(cs):
                eng.gradients_device(cs, T, d_pid.data_ptr(), d_bl.data_ptr(), d_par.data_ptr(), d_ll.data_ptr(),
                                     d_g.data_ptr(), d_s.data_ptr(), None)
            with torch.cuda.stream(gs):
                for _ in range(3):
                    step(gs.cuda_stream)
            torch.cuda.synchronize()
            eng.check_status()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=gs):
                step(torch.cuda.current_stream().cuda_stream)
            for _ in range(5):
                graph.replay()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(200):
                    graph.replay()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 200)
            eng.check_status()
            out = torch.cat([d_ll, d_g.ravel(), d_s]).cpu().numpy()
            if ref is None:
                ref = out
            res[name].append((best, eng.last_call_path(), np.array_equal(ref, out)))
            eng.close()
    for name, _ in variants:
        r = res[name]
        print("%-22s %s ms per %d-tree step  %s  %s" % (
            name, " ".join("%.4f" % x[0] for x in r), T, "same" if all(x[2] for x in r) else "DIFFER", r[0][1]), flush=True)


if __name__ == "__main__":
    main()
