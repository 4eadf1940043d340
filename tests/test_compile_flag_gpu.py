"""The walk-kernel files are compiled with an LLVM option the AMDGPU pipeline leaves off by
default (-structurizecfg-skip-uniform-regions, libsbn_amd/csrc/Makefile): a miscompile under
it would be silent.  `make noskip` builds the same library WITHOUT the option
(libsbn_amd/variants/noskip.so); this test runs the same calls through both builds, each in
its own process (MI_PHYLO_LIBRARY selects the build), and compares every output bit for bit:
4-state gradients (JC69 and GTR finite differences, rescaled and not, both walk generations,
the arena store, K = 8), log-likelihoods, a rooted call and the 20-state kernels."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NOSKIP = os.path.join(REPO, "libsbn_amd", "variants", "noskip.so")

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "tests")); sys.path.insert(0, sys.argv[1])
import libsbn_amd as L
import oracle_lib as O, tree_utils as TU, aa_utils as A
out = {}
rng = np.random.default_rng(123)
st = O.load_struct("ds1_top100")
tips, w, pids, _ = O.struct_arrays(st)
T = 24
pids = pids[:T]
bls = rng.exponential(0.1, size=(T, pids.shape[1] + 1)); bls[:, -1] = 0
def grads(tag, eng, pr, **kw):
    g = eng.gradients(pids, bls, pr, **kw)
    out[tag + ".ll"] = np.array([x.log_likelihood for x in g])
    for k in g[0].gradient:
        out[tag + "." + k] = np.stack([x.gradient[k] for x in g])
jc = np.ones((T, 2)); jc[:, 0] = rng.uniform(0.4, 1.6, T)
for walk in ("v3", "v2"):
    os.environ["MI_PHYLO_GRADIENT_WALK"] = walk
    for store in ("", "arena"):
        if store: os.environ["MI_PHYLO_GRADIENT_STORE"] = store
        else: os.environ.pop("MI_PHYLO_GRADIENT_STORE", None)
        eng = L.Engine(L.PhyloModelSpecification("JC69", "weibull+4", "strict"), tips, w)
        for resc in (False, True):
            grads(f"jc.{walk}.{store}.{int(resc)}", eng, jc, rescaling=resc)
            assert walk != "v3" or store or eng.last_call_info()[0] in ("gradient_walk_lut_kernel", "gradient_walk_lut_fused_kernel")
        out[f"jc.ll.{walk}.{store}"] = eng.log_likelihoods(pids, bls, jc)
        eng.close()
os.environ.pop("MI_PHYLO_GRADIENT_WALK", None); os.environ.pop("MI_PHYLO_GRADIENT_STORE", None)
r, f = TU.random_gtr_params(T, rng)
gtr = np.hstack([r, f, jc])
eng = L.Engine(L.PhyloModelSpecification("GTR", "weibull+4", "strict"), tips, w)
grads("gtr", eng, gtr); eng.close()
eng = L.Engine(L.PhyloModelSpecification("JC69", "weibull+8", "strict"), tips, w)
grads("k8", eng, jc); eng.close()
# rooted
n = 12
rt, rw = TU.random_alignment(n, 77, rng)
trees = [TU.clocklike_rooted_tree(n, rng) for _ in range(3)]
rp, rb = np.stack([t[0] for t in trees]), np.stack([t[1] for t in trees])
stt = [O.time_tree_init(n, t[0], t[1], t[2]) for t in trees]
h, bd, ra = (np.stack([s[i] for s in stt]) for i in range(3))
rates = np.full((3, 2 * n - 2), 0.05)
eng = L.Engine(L.PhyloModelSpecification("JC69", "constant", "strict"), rt, rw)
g = eng.rooted_gradients(rp, rb, np.ones((3, 1)), rates, [1, 1, 1], h, bd, ra)
out["rooted.ratios"] = np.stack([x.gradient["ratios_root_height"] for x in g]); eng.close()
# 20 states
at, aw = A.random_aa_alignment(40, 700, rng)
ap, ab = TU.random_trees(40, 3, rng)
apr = A.params_for("weibull+4", 3, rng)
eng = L.Engine(L.PhyloModelSpecification("WAG", "weibull+4", "strict"), at, aw)
g = eng.gradients(ap, ab, apr)
out["aa.ll"] = np.array([x.log_likelihood for x in g])
out["aa.g"] = np.stack([x.gradient["branch_lengths"] for x in g])
out["aa.site"] = np.stack([x.gradient["site_model"] for x in g])
out["aa.ll2"] = eng.log_likelihoods(ap, ab, apr); eng.close()
np.savez(sys.argv[2], **out)
'''


def _run(tmp_path, name, library):
    env = dict(os.environ)
    env.pop("MI_PHYLO_LIBRARY", None)
    if library:
        env["MI_PHYLO_LIBRARY"] = library
    out = tmp_path / f"{name}.npz"
    r = subprocess.run([sys.executable, "-c", WORKER, REPO, str(out)], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return np.load(out)


def test_uniform_region_flag_does_not_change_a_bit(tmp_path):
    assert os.path.exists(NOSKIP), ("libsbn_amd/variants/noskip.so is missing: "
                                    "`make -C libsbn_amd/csrc noskip` (__graft_entry__.build does)")
    a = _run(tmp_path, "product", None)
    b = _run(tmp_path, "noskip", NOSKIP)
    assert sorted(a.files) == sorted(b.files) and len(a.files) > 30
    for k in a.files:
        assert np.all(np.isfinite(a[k])), k
        assert np.array_equal(a[k], b[k]), k
