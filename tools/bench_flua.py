"""fluA rooted (BASELINE.json configs[3]): ms per phylo_gradients call for T trees (argv[1]).
Used for the rocprofv3 summaries in profiles/ (per-kernel times of the rooted path)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, libsbn_amd as L
dev = torch.device("cuda:0")
Tf = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tips, w, pids, bls, rates, counts, hs, bs, rs = bench.flua_workload(Tf)
n, P = tips.shape; N = 2 * n - 1
eng = L.Engine(L.PhyloModelSpecification("JC69", "constant", "strict"), tips, w, device=0)
stream = torch.cuda.current_stream().cuda_stream
d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (pids, bls, np.ones((Tf, 1)), rates, counts, hs, bs, rs)]
ll = torch.empty(Tf, dtype=torch.float64, device=dev)
gr = torch.empty((Tf, n - 1), dtype=torch.float64, device=dev)
gc = torch.empty((Tf, N - 1), dtype=torch.float64, device=dev)
lib, h = eng._lib, eng._h
eng.reserve(Tf, True)
def call():
    rc = lib.mi_engine_gradients_rooted_device(h, stream, Tf, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
        d[3].data_ptr(), d[4].data_ptr(), d[5].data_ptr(), d[6].data_ptr(), d[7].data_ptr(), 0,
        ll.data_ptr(), gr.data_ptr(), gc.data_ptr(), None, None)
    assert rc == 0
for _ in range(3): call()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(50): call()
torch.cuda.synchronize()
print("T", Tf, "ms/call", (time.perf_counter() - t0) / 50 * 1e3, float(ll[0]))
