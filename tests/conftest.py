import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


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
