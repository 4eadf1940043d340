"""Soak of the one-launch call's in-kernel hand-off: for several batch sizes the
fused step is replayed from a hipGraph tens of thousands of times, part of the time beside
another engine's 1000-tree batches on a second stream (uneven load), and every replay's
outputs are compared bit for bit with the four-launch results -- on the device, so that no
replay goes unchecked.  Prints one line per batch size.
  python tools/soak_fused.py [replays per size, default 20000]"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import libsbn_amd as L
import bench as B

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
tips, w, pids, bls = B.ds1_workload(1000)
n, P = tips.shape; N = 2 * n - 1
rng = np.random.default_rng(7)
params = np.ones((1000, 2)); params[:, 0] = rng.uniform(0.3, 2.0, 1000)
dev = torch.device("cuda", 0)
os.environ["MI_PHYLO_FUSED_SETUP"] = "1"
fused = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w, device=0)
os.environ["MI_PHYLO_FUSED_SETUP"] = "0"
plain = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w, device=0)
side = torch.cuda.Stream()
b = [torch.from_numpy(x).to(dev) for x in (pids, bls, params)]
b_ll = torch.zeros(1000, dtype=torch.float64, device=dev); b_g = torch.zeros((1000, N), dtype=torch.float64, device=dev)
plain.reserve(1000, True)
total_bad = 0
for T in (1, 7, 64, 125, 300, 512):
    ref = plain.gradients(pids[:T], bls[:T], params[:T])
    r_ll = torch.from_numpy(np.array([g.log_likelihood for g in ref])).to(dev)
    r_g = torch.from_numpy(np.stack([g.gradient["branch_lengths"] for g in ref])).to(dev)
    r_s = torch.from_numpy(np.array([g.gradient["site_model"][0] for g in ref])).to(dev)
    d = [torch.from_numpy(x[:T]).to(dev) for x in (pids, bls, params)]
    d_ll = torch.zeros(T, dtype=torch.float64, device=dev); d_g = torch.zeros((T, N), dtype=torch.float64, device=dev)
    d_s = torch.zeros(T, dtype=torch.float64, device=dev)
    bad = torch.zeros(1, dtype=torch.int64, device=dev)
    fused.reserve(T, True)
    gs = torch.cuda.Stream()
    def step(cs):
        fused.gradients_device(cs, T, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d_ll.data_ptr(), d_g.data_ptr(), d_s.data_ptr(), None)
    with torch.cuda.stream(gs):
        step(gs.cuda_stream); step(gs.cuda_stream)
    torch.cuda.synchronize()
    assert fused.last_call_info()[0] == "gradient_walk_lut_fused_kernel", fused.last_call_info()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=gs):
        d_ll.zero_(); d_g.zero_(); d_s.zero_()
        step(torch.cuda.current_stream().cuda_stream)
        # (the check is part of the graph: every replay is compared, on the device)
        bad += (d_ll != r_ll).sum() + (d_g != r_g).sum() + (d_s != r_s).sum()
    for k in range(reps):
        if k % 5 == 0 and (k // 1000) % 2 == 0:  # beside some stretches: the other engine's big batch
            plain.gradients_device(side.cuda_stream, 1000, b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr(), b_ll.data_ptr(), b_g.data_ptr(), None, None)
        graph.replay()
    torch.cuda.synchronize()
    fused.check_status(); plain.check_status()
    nb = int(bad.item()); total_bad += nb
    print("T=%4d: %d graph replays of the one-launch step, %d mismatching output values" % (T, reps, nb), flush=True)
print("soak:", "OK" if total_bad == 0 else "MISMATCHES %d" % total_bad)
