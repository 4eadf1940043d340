"""Builds and runs the C++ Engine adapter example (the reference's doctests against
libsbn_amd/csrc/host/engine.hpp) on the GPU box."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_engine_adapter(tmp_path):
    exe = tmp_path / "engine_example"
    lib = os.path.join(REPO, "libsbn_amd")
    subprocess.run(["g++", "-std=c++17", "-O1", os.path.join(REPO, "tests/cpp/engine_example.cpp"),
                    "-L" + lib, "-lmi_phylo", "-lmi_phylo_host", "-Wl,-rpath," + lib,
                    "-o", str(exe)], check=True)
    out = subprocess.run([str(exe), os.path.join(REPO, "tests/golden/data")],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout


def test_tree_setup_kernels_agree(tmp_path):
    """The register-array tree-setup kernel (N <= 256, branch-free) must produce exactly
    what the general LDS kernel produces: status, macro counts, the Sethi-Ullman schedule
    with its LDS slots, and the half-storage gradient schedule."""
    import numpy as np
    import tree_utils as TU
    exe = tmp_path / "tree_setup_compare"
    lib = os.path.join(REPO, "libsbn_amd")
    hipcc = "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-Wno-unused-result",
                    os.path.join(REPO, "tests/cpp/tree_setup_compare.hip"),
                    "-L" + lib, "-lmi_phylo", "-Wl,-rpath," + lib, "-o", str(exe)], check=True)
    rng = np.random.default_rng(11)
    for n in (3, 4, 5, 6, 9, 17, 27, 32, 33, 50, 64, 65, 69, 100, 128):
        T = 40
        pids, _ = TU.random_trees(n, T, rng)
        pids[0] = TU.ladder_topology(n)
        trees = tmp_path / f"trees_{n}.bin"
        pids.astype(np.int32).tofile(trees)
        dumps = {}
        for mode in ("small", "lds"):
            env = dict(os.environ)
            if mode == "lds":
                env["MI_PHYLO_TREE_SETUP"] = "lds"
            out = tmp_path / f"dump_{n}_{mode}.bin"
            r = subprocess.run([str(exe), str(n), str(T), str(trees), str(out)], env=env,
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
            dumps[mode] = np.fromfile(out, dtype=np.int32)
        assert np.array_equal(dumps["small"], dumps["lds"]), f"n = {n}"
