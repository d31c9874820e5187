"""ctypes loader for libowshen_gpu.so.  Fails loudly if the library is missing -- there is
no fallback path (the oracle under ``oracle/`` is test infrastructure, never imported here)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OWSHEN_GPU_LIB: another build of the same library (same-box A/B runs of two builds, tools/gpu_round.sh)
LIB_PATH = os.environ.get("OWSHEN_GPU_LIB") or os.path.join(_HERE, "libowshen_gpu.so")


class OwshenGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libowshen_gpu error {code}: {msg}")
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C owshen_amd/csrc`.  owshen_amd has no CPU fallback.")
    return C.CDLL(LIB_PATH)


lib = _load()

from ._abi import SIGNATURES, bind  # noqa: E402,F401

bind(lib)  # AttributeError here = header/library drift, fail loudly


def check(code):
    if code != 0:
        raise OwshenGpuError(code, lib.og_last_error().decode("utf-8", "replace"))
