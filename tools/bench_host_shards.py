"""Host-pointer call of the headline batch (mi_engine_gradients_unrooted, numpy buffers) through
ONE handle with 1..4 logical shards on the same device (shard_devices = [-1] * D): the shards'
uploads, kernels and downloads run on streams of their own, so a shard's transfers and host
copies overlap the other shards' kernels.  Results are compared bit for bit with one shard's.
  python3 tools/bench_host_shards.py [trees] [steps]"""
import ctypes as C
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench as B  # noqa: E402
import libsbn_amd as L  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
tips, w, pids, bls = B.ds1_workload(T)
params = np.full((T, 1), 0.5)
spec = L.PhyloModelSpecification("JC69", "weibull+4", "none")
pid = np.ascontiguousarray(pids, dtype=np.int32)
bl = np.ascontiguousarray(bls, dtype=np.float64)
pr = np.ascontiguousarray(params, dtype=np.float64)
N = bl.shape[1] + 1  # (2 taxa - 1 gradient entries per tree)
ptr = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
first = None
for D in (1, 2, 3, 4, 1, 2):
    eng = L.Engine(spec, tips, w, shard_devices=[-1] * D)
    ll, g, site = np.empty(T), np.empty((T, N)), np.empty(T)

    def call():
        rc = eng._lib.mi_engine_gradients_unrooted(eng._h, T, ptr(pid), ptr(bl), ptr(pr), 0,
                                                   ptr(ll), ptr(g), ptr(site), None)
        if rc:
            raise RuntimeError(eng._check(rc))

    for _ in range(5):
        call()
    t0 = time.perf_counter()
    for _ in range(steps):
        call()
    dt = (time.perf_counter() - t0) / steps
    if first is None:
        first = (ll.copy(), g.copy(), site.copy())
    same = bool(np.array_equal(ll, first[0]) and np.array_equal(g, first[1]) and np.array_equal(site, first[2]))
    print(json.dumps({"shards": D, "trees": T, "ms_per_call": round(dt * 1e3, 4), "trees_per_s": round(T / dt),
                      "bit_identical_to_one_shard": same}))
    del eng
