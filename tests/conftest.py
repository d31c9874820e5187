import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_sessionstart(session):
    """The C-ABI library is built in-tree (git-ignored); make sure it is current before anything imports
    owshen_amd.  A no-op when up to date; hipcc cross-compiles gfx950 without a GPU."""
    import fcntl
    import subprocess
    with open(os.path.join(ROOT, "owshen_amd", ".build.lock"), "w") as lk:   # (xdist: the controller and every worker start a session)
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "owshen_amd", "csrc")])
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle", "c")])


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One og_ctx on cuda:0 for the whole GPU session.  No fallback: a missing library or
    device is a hard failure, never a skip.  This is the SHIPPED library (owshen_amd/libowshen_gpu.so): it reads no A/B or
    test hook from the environment, so a test that sets one must use `ctx_hooks`."""
    from owshen_amd import api
    c = api.Context(0)
    yield c
    c.close()


def hooks_lib():
    """owshen_amd/libowshen_gpu_hooks.so: the same sources built with -DOG_AB_HOOKS (owshen_amd/csrc/ctx.h) -- the ~50 OG_*
    environment switches, the rejected kernel variants (msm_ab.hip.h) and the multi-device failure injection exist only there."""
    import ctypes as C
    from owshen_amd import _lib
    from owshen_amd._abi import bind
    path = os.path.join(os.path.dirname(_lib.LIB_PATH), "libowshen_gpu_hooks.so")
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: run `make -C owshen_amd/csrc`")
    global _HOOKS
    if _HOOKS is None:
        _HOOKS = bind(C.CDLL(path))
    return _HOOKS


_HOOKS = None


@pytest.fixture(scope="session")
def ctx_hooks():
    """an og_ctx of the HOOKS build on cuda:0: for the tests that switch launch forms / thresholds through OG_* variables"""
    from owshen_amd import api

    class HooksContext(api.Context):
        _lib = hooks_lib()

    c = HooksContext(0)
    yield c
    c.close()
