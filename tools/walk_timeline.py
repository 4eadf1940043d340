"""Wave timeline of gradient_walk_kernel at T trees per call (argv[1], default 125).

Needs the diagnostic build (`make -C libsbn_amd/csrc timeline` -> libsbn_amd/variants/timeline.so):
every wave records s_memrealtime (constant 100 MHz, one time base for the whole device) and
s_memtime (shader clock) at its start and end.  Prints the launch's span, the mean wave
duration in microseconds and in shader cycles (their ratio is the clock the launch ran at),
wave durations by start time, and how many of the 2 048 wave slots are occupied over time --
the ramp, the plateau and the tail of a launch ("small batches")."""
import ctypes, os, sys
import numpy as np
os.environ["MI_PHYLO_LIBRARY"] = os.path.abspath("libsbn_amd/variants/timeline.so")
os.environ["MI_PHYLO_GRADIENT_WALK"] = "v2"  # (the instrumented kernel is the second generation's)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import libsbn_amd as L
from libsbn_amd import sharding
T = int(sys.argv[1]) if len(sys.argv) > 1 else 125
dev = torch.device("cuda", 0)
compute = torch.cuda.Stream(dev); torch.cuda.set_stream(compute)
tips, w, pids, bls = bench.ds1_workload(1000)
params = np.ones((len(pids), 2))
n, P = tips.shape; N = 2 * n - 1
eng = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w, device=0)
stream = torch.cuda.current_stream().cuda_stream
d_pid = torch.from_numpy(pids[:T]).to(dev); d_bl = torch.from_numpy(bls[:T]).to(dev); d_par = torch.from_numpy(params[:T]).to(dev)
blk = sharding.ResultBlocks(T, N, extra=1, device=dev)
eng.reserve(T, True)
for _ in range(8):
    eng.gradients_device(stream, T, d_pid.data_ptr(), d_bl.data_ptr(), d_par.data_ptr(), blk.log_likelihoods.data_ptr(), blk.branch_gradients.data_ptr(), blk.extras[0].data_ptr(), None)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["MI_PHYLO_LIBRARY"])
buf = np.zeros((65536, 8), dtype=np.int64)
rc = lib.mi_debug_walk_timeline(buf.ctypes.data_as(ctypes.c_void_p), 65536)
assert rc == 0, rc
buf = buf[buf[:, 6] != 0]  # the waves of the launch (the first 65 536 of them)
waves = len(buf)
cyc = (buf[:, 3] - buf[:, 0]).astype(float)
phase = np.diff(buf[:, :4].astype(float), axis=1)  # prologue | post-order | pre-order, shader cycles
rt = (buf[:, 6:8] - buf[:, 6].min()) * 0.01  # us, 100 MHz constant clock
dur = rt[:, 1] - rt[:, 0]
print("T %d: %d waves; launch span (first wave start -> last wave end) %.1f us; mean wave %.2f us = %.0f shader "
      "cycles (prologue %.0f, post-order %.0f, pre-order %.0f) => %.2f GHz; work / 2048 slots = %.1f us"
      % (T, waves, rt[:, 1].max(), dur.mean(), cyc.mean(), phase[:, 0].mean(), phase[:, 1].mean(),
         phase[:, 2].mean(), cyc.sum() / dur.sum() / 1e3, dur.sum() / 2048))
nb = 8
edges = np.linspace(0, rt[:, 0].max() + 1e-9, nb + 1)
print("  waves by start time (us): count, mean duration us, mean shader cycles")
for i in range(nb):
    sel = (rt[:, 0] >= edges[i]) & (rt[:, 0] < edges[i + 1] + (1e-9 if i == nb - 1 else 0))
    if sel.sum():
        print("  %7.1f-%7.1f %6d  %6.2f  %8.0f" % (edges[i], edges[i + 1], sel.sum(), dur[sel].mean(), cyc[sel].mean()))
ts = np.linspace(0, rt[:, 1].max(), 17)
print("  occupied wave slots at t (us):", " ".join("%.0f:%d" % (t, int(((rt[:, 0] <= t) & (rt[:, 1] > t)).sum())) for t in ts))
