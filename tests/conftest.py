import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built libraries (they are git-ignored): build what is MISSING, once, before collection.
    Building is not using -- the product library still refuses to run without its HIP code, and the CPU restatement
    stays the checker. Existing libraries are left alone (the GPU box receives them prebuilt)."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "tod_amd", "libtodhip.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tod_amd", "csrc"), "-s", "-j4"], check=True)
    if not os.path.exists(os.path.join(ROOT, "oracle", "libtod_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
