import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "retinanet-tensorflow_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _ensure_built():
    """librn_hip.so missing or older than a kernel source / the header -> rebuild it (hipcc cross-compiles; no GPU needed)."""
    import glob
    import subprocess
    src = os.path.join(PKG, "csrc")
    so = os.path.join(PKG, "librn_hip.so")
    deps = [f for pat in ("*.hip", "*.h", "*.cpp", "Makefile") for f in glob.glob(os.path.join(src, pat))]
    deps.append(os.path.join(ROOT, "include", "rn_hip.h"))
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.run(["make", "-C", src, "-j8"], check=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
