import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_sessionstart(session):
    """Build container only (where /root/reference exists): a libsbgm_hip.so older than its sources would travel to the GPU box
    as it is, so rebuild it before any test runs.  The GPU box uses the prebuilt file."""
    import subprocess
    csrc = os.path.join(ROOT, "sbgm_danra_amd", "csrc")
    if os.path.isdir("/root/reference") and subprocess.run(["make", "-q", "-C", csrc], capture_output=True).returncode != 0:
        subprocess.run(["make", "-C", csrc, "-j8"], check=True, capture_output=True)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm device (MI355X); run with -m gpu")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")
