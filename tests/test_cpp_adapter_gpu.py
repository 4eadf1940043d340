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
