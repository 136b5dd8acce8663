import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# UBD_GOLDEN_DIR: fixtures regenerated FROM THE REFERENCE by tests/golden/make_reference_golden.py (same file names and keys)
GOLDEN = os.environ.get("UBD_GOLDEN_DIR") or os.path.join(ROOT, "tests", "golden")


def _ensure_built():
    """The C-ABI library is built in-tree (hipcc cross-compiles gfx950 without a GPU); build it when a fresh
    checkout has not run __graft_entry__.build() yet."""
    lib = os.path.join(ROOT, "ubdvss_amd", "libubd_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.check_call(["bash", os.path.join(ROOT, "ubdvss_amd", "csrc", "build.sh")])


_ensure_built()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def manifest():
    import json
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)
