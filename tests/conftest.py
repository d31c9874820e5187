import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_sessionstart(session):
    """The C-ABI library is built in-tree (git-ignored); make sure it is current before anything imports
    owshen_amd.  A no-op when up to date; hipcc cross-compiles gfx950 without a GPU."""
    import subprocess
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "owshen_amd", "csrc")])
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "c")])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One og_ctx on cuda:0 for the whole GPU session.  No fallback: a missing library or
    device is a hard failure, never a skip."""
    from owshen_amd import api
    c = api.Context(0)
    yield c
    c.close()
