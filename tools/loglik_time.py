"""ms per 1000-tree DS1 call of log_likelihoods (JC69+G4, GTR+G4) and of the GTR+G4 full gradient
(16 finite-difference passes): one library per process (MI_PHYLO_LIBRARY), for A/B runs of builds of
kernels_loglik.hip (profiles/r06_loglik_tip_lookup_timing.txt).
  [MI_PHYLO_LIBRARY=libsbn_amd/variants/x.so] python tools/loglik_time.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import bench as B, libsbn_amd as L, tree_utils as TU
T = int(os.environ.get("LL_TREES", "1000"))
tips, w, pids, bls = B.ds1_workload(T)
n, P = tips.shape; N = 2 * n - 1
dev = torch.device("cuda", 0)
out = []
for subst, mode in (("JC69", "loglik"), ("GTR", "loglik"), ("GTR", "grad")):
    eng = L.Engine(L.PhyloModelSpecification(subst, "weibull+4", "strict"), tips, w, device=0)
    rng = np.random.default_rng(5)
    params = np.ones((T, eng.param_count))
    if subst == "GTR":
        params = np.hstack([rng.dirichlet(10 * np.ones(6), T), rng.dirichlet(10 * np.ones(4), T), np.ones((T, 2))])
    d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (pids.astype(np.int32), bls, params)]
    d_ll = torch.zeros(T, dtype=torch.float64, device=dev); d_g = torch.zeros((T, N), dtype=torch.float64, device=dev)
    d_s = torch.zeros(T, dtype=torch.float64, device=dev); d_q = torch.zeros((T, 8), dtype=torch.float64, device=dev)
    cs = torch.cuda.Stream(); torch.cuda.set_stream(cs); st = cs.cuda_stream
    def step():
        if mode == "loglik":
            eng.log_likelihoods_device(st, T, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d_ll.data_ptr())
        else:
            eng.gradients_device(st, T, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d_ll.data_ptr(), d_g.data_ptr(),
                                 d_s.data_ptr(), d_q.data_ptr() if subst == "GTR" else None)
    for _ in range(3): step()
    eng.check_status(st); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): step()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 5)
    out.append("%s/%s %.4f ms (ll0 %.12g sum %.12g)" % (subst, mode, best, float(d_ll[0]), float(d_ll.sum())))
    eng.close()
print(os.environ.get("MI_PHYLO_LIBRARY", "default").split("/")[-1], " | ".join(out), flush=True)
