import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")
    # Some GPU tests hand torch device tensors to the *_device entry points.  torch ships its
    # own copy of the HIP runtime: when libmi_phylo.so (linked against /opt/rocm's) initialises
    # HIP first, torch's later initialisation reports "No HIP GPUs are available".  Loading
    # torch first makes both resolve to one runtime ("Python callers").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    # The native libraries are build products (git-ignored): on a fresh checkout build them
    # once (hipcc cross-compiles gfx950 without a GPU).  A failure surfaces in the tests
    # that need them -- nothing falls back to another implementation.
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    needed = [os.path.join(repo, "libsbn_amd", "libmi_phylo.so"),
              os.path.join(repo, "libsbn_amd", "libmi_phylo_host.so")]
    if not all(os.path.exists(p) for p in needed):
        import shutil
        import subprocess
        if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"):
            subprocess.run(["make", "-C", os.path.join(repo, "libsbn_amd", "csrc"), "all"],
                           check=False)
