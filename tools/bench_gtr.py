#!/usr/bin/env python3
"""DS1 x 1000 trees, GTR+weibull+4, full phylo_gradients (BASELINE.json configs[2]) -- the call
profiled under rocprofv3 for profiles/r02_gtr_*."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import libsbn_amd as L  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
tips, w, pids, bls = bench.ds1_workload(T)
rng = np.random.default_rng(45)
gtr = np.hstack([rng.dirichlet(10 * np.ones(6), T), rng.dirichlet(10 * np.ones(4), T),
                 np.ones((T, 2))])
d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (pids, bls, gtr)]
N = 53
ll = torch.empty(T, dtype=torch.float64, device=dev)
g = torch.empty((T, N), dtype=torch.float64, device=dev)
site = torch.empty(T, dtype=torch.float64, device=dev)
sub = torch.empty((T, 8), dtype=torch.float64, device=dev)
eng = L.Engine(L.PhyloModelSpecification("GTR", "weibull+4", "strict"), tips, w, device=0)
stream = torch.cuda.current_stream().cuda_stream
eng.reserve(T, True)


def call():
    eng.gradients_device(stream, T, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                         ll.data_ptr(), g.data_ptr(), site.data_ptr(), sub.data_ptr())


for _ in range(2):
    call()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    call()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
eng.check_status(stream)
print(f"GTR+weibull+4 full phylo_gradients: {1e3 * dt:.3f} ms per {T} trees = {T / dt:.0f} trees/s; "
      f"checksum {float(ll.sum()):.6f} {float(sub.sum()):.9f} {float(site.sum()):.9f}")
