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
    """The register-array tree-setup kernel (N <= 256, branch-free) and the workgroup-per-tree
    kernel (larger trees) must produce exactly what the sequential LDS kernel produces:
    status, macro counts, the Sethi-Ullman schedule with its LDS slots, and the half-storage
    gradient schedule."""
    import sys
    import numpy as np
    import tree_utils as TU
    sys.setrecursionlimit(20000)  # (tree_utils walks ladder trees recursively)
    exe = tmp_path / "tree_setup_compare"
    lib = os.path.join(REPO, "libsbn_amd")
    hipcc = "/opt/rocm/bin/hipcc"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-Wno-unused-result",
                    os.path.join(REPO, "tests/cpp/tree_setup_compare.hip"),
                    "-L" + lib, "-lmi_phylo", "-Wl,-rpath," + lib, "-o", str(exe)], check=True)
    rng = np.random.default_rng(11)

    def dump(n, T, trees, mode, rooted, slots="default"):
        env = dict(os.environ)
        env.pop("MI_PHYLO_TREE_SETUP", None)
        env.pop("MI_PHYLO_MACRO_SLOTS", None)
        if mode != "default":
            env["MI_PHYLO_TREE_SETUP"] = mode
        if slots != "default":
            env["MI_PHYLO_MACRO_SLOTS"] = slots
        out = tmp_path / f"dump_{n}_{mode}_{rooted}_{slots}.bin"
        fold = slots == "fold"  # (round 6: the slot assignment in the set-up launch itself)
        if fold:
            env.pop("MI_PHYLO_MACRO_SLOTS")
        r = subprocess.run([str(exe), str(n), str(T), str(trees), str(out), str(rooted), str(int(fold))], env=env,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        if fold:
            assert "folded=1" in r.stdout, r.stdout
        return np.fromfile(out, dtype=np.int32)

    # up to 64 nodes the register-array kernel is the default, above the workgroup kernel;
    # every kernel that applies to a size == the sequential LDS kernel
    for n in (3, 4, 5, 6, 9, 17, 27, 32, 33, 50, 64, 65, 69, 100, 128, 129, 200, 333, 512, 700):
        for rooted in (0, 1):
            if rooted and n not in (3, 9, 69, 129, 512):
                continue
            T = 40 if n <= 128 else 12
            pids, _ = TU.random_trees(n, T, rng, rooted=bool(rooted))
            pids[0] = TU.ladder_topology(n, rooted=bool(rooted))
            pids[1] = TU.balanced_topology(n, rooted=bool(rooted))
            trees = tmp_path / f"trees_{n}_{rooted}.bin"
            pids.astype(np.int32).tofile(trees)
            # (the dump ends with the macro order / LDS slots / arena indices of the arena variant:
            # workgroup-per-tree macro_slots kernel by default, the sequential one for `ref`)
            ref = dump(n, T, trees, "lds", rooted, slots="seq")
            assert np.array_equal(dump(n, T, trees, "default", rooted), ref), f"n = {n}"
            if n >= 4:  # (the workgroup kernel, forced below 65 nodes, with the arena's slot assignment folded in)
                assert np.array_equal(dump(n, T, trees, "wg", rooted, slots="fold"), ref), f"n = {n} (wg + slots)"
            if n <= 128:
                assert np.array_equal(dump(n, T, trees, "wg", rooted), ref), f"n = {n} (wg)"
                assert np.array_equal(dump(n, T, trees, "small", rooted), ref), f"n = {n} (small)"
    # invalid trees (one per batch: which bad tree is reported first is a race): the same
    # status from every kernel
    n, T = 40, 6
    for case in range(2):
        pids, _ = TU.random_trees(n, T, rng)
        if case == 0:
            pids[2, 5] = 3                # a parent that is a tip
        else:
            pids[4, 7] = pids[4, 60]      # an internal node with three children (or a bad id)
        trees = tmp_path / f"trees_bad{case}.bin"
        pids.astype(np.int32).tofile(trees)
        heads = [dump(n, T, trees, mode, 0)[:2] for mode in ("lds", "default", "wg", "small")]
        assert heads[0][0] != 0
        assert all(np.array_equal(heads[0], h) for h in heads[1:]), heads
