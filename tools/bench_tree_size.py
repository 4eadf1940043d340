"""Gradient throughput for random trees of a given size: python tools/bench_tree_size.py <taxa> <patterns>
(1000 trees, JC69 + 4 rate categories; used for DESIGN.md 4.1 / 6 and profiles/r01i_arena_*)."""
import sys, time
import os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import libsbn_amd as L, tree_utils as TU
n, P = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(3)
T = 1000
tips, w = TU.random_alignment(n, P, rng)
pids, bls = TU.random_trees(n, 50, rng, mean_bl=0.05)
pids = np.tile(pids, (T // 50, 1)); bls = np.tile(bls, (T // 50, 1)) * rng.uniform(0.5, 1.5, size=(T, 1))
N = 2 * n - 1
eng = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w, device=0)
dev = torch.device("cuda", 0)
params = np.ones((T, 2))
d_pid = torch.from_numpy(pids.astype(np.int32)).to(dev); d_bl = torch.from_numpy(bls).to(dev); d_par = torch.from_numpy(params).to(dev)
d_ll = torch.zeros(T, dtype=torch.float64, device=dev); d_site = torch.zeros(T, dtype=torch.float64, device=dev)
d_g = torch.zeros((T, N), dtype=torch.float64, device=dev)
st = torch.cuda.current_stream().cuda_stream
def step():
    eng.gradients_device(st, T, d_pid.data_ptr(), d_bl.data_ptr(), d_par.data_ptr(), d_ll.data_ptr(), d_g.data_ptr(), d_site.data_ptr(), None, False)
for _ in range(2): step()
eng.check_status(st); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("n=%d P=%d grad: %.3f ms/1000 trees (%.0f trees/s)" % (n, P, dt * 1e3, T / dt))
