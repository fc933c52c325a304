import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


def pytest_sessionstart(session):
    """Build what is missing (hipcc cross-compiles gfx950 anywhere; gcc for the oracle).  The built libraries
    normally travel with the tree; this only covers a checkout that was never built."""
    import subprocess
    pkg = os.path.join(ROOT, "raytracing-in-windows-console_amd")
    needed = [os.path.join(pkg, "librtx_hip.so"), os.path.join(pkg, "headless_engine"),
              os.path.join(HERE, "gpu_checks", "libmath_check.so")]
    if not all(os.path.exists(p) for p in needed):
        subprocess.call(["make", "-C", pkg], stdout=subprocess.DEVNULL)
    if not os.path.exists(os.path.join(ROOT, "oracle", "librtx_oracle.so")):
        subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
