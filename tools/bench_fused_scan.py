"""One-launch call against the four-launch sequence over batch sizes: the
phylo_gradients step (logL + branch gradient) of the DS1 engine replayed from a hipGraph, with
MI_PHYLO_FUSED_SETUP=1 and =0, results compared bit for bit.
  python tools/bench_fused_scan.py [JC69|GTR]      (profiles/r05_fused_scan.txt)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import libsbn_amd as L
import bench as B
tips, w, pids, bls = B.ds1_workload(4000)
n, P = tips.shape; N = 2 * n - 1
dev = torch.device("cuda", 0)
model = sys.argv[1] if len(sys.argv) > 1 else "JC69"
rng = np.random.default_rng(1)
if model == "GTR":
    import tree_utils as TU
    r, f = TU.random_gtr_params(len(pids), rng)
    params = np.hstack([r, f, np.ones((len(pids), 2))])
else:
    params = np.ones((len(pids), 2))
res = {}
for name, env in (("fused", "1"), ("unfused", "0")):
    os.environ["MI_PHYLO_FUSED_SETUP"] = env
    eng = L.Engine(L.PhyloModelSpecification(model, "weibull+4", "strict"), tips, w, device=0)
    for T in (1, 8, 32, 64, 125, 250, 500, 1000, 2000, 4000):
        d_pid = torch.from_numpy(pids[:T]).to(dev); d_bl = torch.from_numpy(bls[:T]).to(dev)
        d_par = torch.from_numpy(params[:T]).to(dev)
        d_ll = torch.zeros(T, dtype=torch.float64, device=dev); d_g = torch.zeros((T, N), dtype=torch.float64, device=dev)
        eng.reserve(T, True)
        gs = torch.cuda.Stream()
        def step(cs):
            eng.gradients_device(cs, T, d_pid.data_ptr(), d_bl.data_ptr(), d_par.data_ptr(), d_ll.data_ptr(), d_g.data_ptr(), None, None)
        with torch.cuda.stream(gs):
            for _ in range(3): step(gs.cuda_stream)
        torch.cuda.synchronize(); eng.check_status()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=gs): step(torch.cuda.current_stream().cuda_stream)
        for _ in range(5): graph.replay()
        torch.cuda.synchronize()
        reps = 200 if T <= 250 else 40
        best = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): graph.replay()
            e1.record(); torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) / reps)
        res[(name, T)] = (min(best), eng.last_call_info()[0], torch.cat([d_ll, d_g.ravel()]).cpu().numpy())
    eng.close()
for T in (1, 8, 32, 64, 125, 250, 500, 1000, 2000, 4000):
    a, b = res[("fused", T)], res[("unfused", T)]
    print("%s T=%5d fused %.4f ms (%s)  unfused %.4f ms (%s)  ratio %.3f  %s" % (model, T, a[0], a[1][-12:], b[0], b[1][-12:], a[0] / b[0], "equal" if np.array_equal(a[2], b[2]) else "DIFFER"))
