"""Audit of the engine's path selection: for a grid of tree sizes, pattern counts, category
counts and batch sizes, times phylo_gradients with the default choice and with every forced
combination of walk generation (MI_PHYLO_GRADIENT_WALK) and store (MI_PHYLO_GRADIENT_STORE), and
prints the configurations where a forced combination beats the default by more than 3 %.
(Round 5 found the K < 3 rule this way: DESIGN.md 4.1.)
  python tools/audit_paths.py [quick]"""
import os, sys, itertools
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np, torch
import libsbn_amd as L, tree_utils as TU

dev = torch.device("cuda", 0)
rng = np.random.default_rng(11)

def time_cfg(n, P, K, T, env):
    for k in ("MI_PHYLO_GRADIENT_WALK", "MI_PHYLO_GRADIENT_STORE", "MI_PHYLO_LOGLIK_PATH", "MI_PHYLO_WALK3_K1",
              "MI_PHYLO_ARENA_NT", "MI_PHYLO_WALK_TILE_REGS"):
        os.environ.pop(k, None)
    os.environ.update(env)
    site = "constant" if K == 1 else f"weibull+{K}"
    tips, w = TU.random_alignment(n, P, np.random.default_rng(n * 1000 + P))
    pids, bls = TU.random_trees(n, min(T, 50), np.random.default_rng(n + P), mean_bl=0.05)
    reps = (T + len(pids) - 1) // len(pids)
    pids = np.ascontiguousarray(np.tile(pids, (reps, 1))[:T]).astype(np.int32)
    bls = np.ascontiguousarray(np.tile(bls, (reps, 1))[:T])
    N = 2 * n - 1
    try:
        eng = L.Engine(L.PhyloModelSpecification("JC69", site, "strict"), tips, w, device=0)
    except RuntimeError as exc:
        return None, str(exc)[:60]
    params = np.ones((T, max(eng.param_count, 1)))
    d = [torch.from_numpy(x).to(dev) for x in (pids, bls, params)]
    d_ll = torch.zeros(T, dtype=torch.float64, device=dev); d_g = torch.zeros((T, N), dtype=torch.float64, device=dev)
    d_s = torch.zeros(T, dtype=torch.float64, device=dev)
    # (an explicit stream: torch's default stream has the null handle, which the C ABI reads as
    # "the engine's own stream" -- the events below would then time nothing)
    cs = torch.cuda.Stream()
    torch.cuda.set_stream(cs)
    st = cs.cuda_stream
    def step():
        if os.environ.get("AUDIT_MODE") == "loglik":
            eng.log_likelihoods_device(st, T, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d_ll.data_ptr())
        else:
            eng.gradients_device(st, T, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d_ll.data_ptr(), d_g.data_ptr(),
                                 d_s.data_ptr() if K > 1 else None, None)
    try:
        for _ in range(2): step()
        eng.check_status(st); torch.cuda.synchronize()
        reps = 20 if T <= 100 else 5
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): step()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        path = eng.last_call_path()
        ll0 = float(d_ll[0])
    except RuntimeError as exc:
        eng.close(); return None, str(exc)[:60]
    eng.close()
    return ms, path + " ll0=%.10g" % ll0

quick = len(sys.argv) > 1
sizes = [int(x) for x in os.environ["AUDIT_SIZES"].split(",")] if os.environ.get("AUDIT_SIZES") else [8, 16, 27, 31, 36, 45, 64, 100]
grid = list(itertools.product(sizes, [200, 1000] if not quick else [300], [1, 2, 4], [16, 1000] if not quick else [1000]))
for n, P, K, T in grid:
    base, bpath = time_cfg(n, P, K, T, {})
    best, bestname, bestpath = base, "default", bpath
    if os.environ.get("AUDIT_MODE") == "loglik":
        variants = [("valu", {"MI_PHYLO_LOGLIK_PATH": "valu"}), ("mfma", {"MI_PHYLO_LOGLIK_PATH": "mfma"})]
    else:
        variants = [(f"{walk}/{store}", {"MI_PHYLO_GRADIENT_WALK": walk, "MI_PHYLO_GRADIENT_STORE": store})
                    for walk, store in itertools.product(("v2", "v3"), ("lds", "arena"))]
        # (round 6: the arena's non-temporal accesses are chosen by tiles per tree)
        if K == 1:
            # (until the tip codes were pre-tiled: the look-up walk only where its one-launch call applies)
            variants.append(("v3k1old/lds", {"MI_PHYLO_GRADIENT_STORE": "lds", "MI_PHYLO_WALK3_K1": "0"}))
        # (and its tile width by the engine's tile counts)
        variants += [("v3/%s/r%s" % (store, regs), {"MI_PHYLO_GRADIENT_WALK": "v3", "MI_PHYLO_GRADIENT_STORE": store,
                                                   "MI_PHYLO_WALK_TILE_REGS": regs})
                     for store in ("lds", "arena") for regs in ("3", "4")]
        variants += [("v3/arena/nt%s" % nt, {"MI_PHYLO_GRADIENT_WALK": "v3", "MI_PHYLO_GRADIENT_STORE": "arena", "MI_PHYLO_ARENA_NT": nt})
                     for nt in ("0", "1")]
    for vname, env in variants:
        ms, path = time_cfg(n, P, K, T, env)
        if ms is not None and ms < best:
            best, bestname, bestpath = ms, vname, path
    flag = "  <== default loses %.0f %%" % (100 * (base / best - 1)) if best < base * 0.97 else ""
    print("n=%3d P=%4d K=%d T=%4d default %.4f ms [%s] best %.4f ms (%s: %s)%s" % (
        n, P, K, T, base, bpath.split(" ll0")[0], best, bestname, bestpath.split(" ll0")[0], flag), flush=True)
