"""TEST INFRASTRUCTURE ONLY -- numpy front end to tests/hipemu/_build/libowshen_emu.so.

That library is the SAME kernel source as the product (owshen_amd/csrc/*.hip) compiled for a
single-threaded CPU interpreter (tests/hipemu/hip/hip_runtime.h), so kernel and prover-glue
logic can be checked against the oracle where no GPU exists.  owshen_amd never loads it; the
`-m gpu` tests are the parity tests proper and run the gfx950 binary.  "Device" buffers here
are plain numpy arrays; `Ctx` reuses the host-side mirror (owshen_amd.api.Context) with the
buffer plumbing swapped, so the host logic under test is the product's own.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from owshen_amd import api
from owshen_amd._abi import bind

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "hipemu")
_SO = os.path.join(_DIR, "_build", "libowshen_emu.so")


def _load():
    # OG_EMU_LIB: another build of the same sources (tools/sanitize_emu.sh: -fsanitize=address / undefined with the runtime
    # LD_PRELOADed) -- the whole interpreter suite then runs on it
    if os.environ.get("OG_EMU_LIB"):
        return bind(C.CDLL(os.environ["OG_EMU_LIB"]))
    # (pytest-xdist workers import this module at the same time: one of them builds, the others wait for the lock)
    import fcntl
    os.makedirs(os.path.join(_DIR, "_build"), exist_ok=True)
    with open(os.path.join(_DIR, "_build", ".lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        subprocess.check_call(["make", "-s", "-j8", "-C", _DIR])
    return bind(C.CDLL(_SO))


lib = _load()

# The proof assembly's four products by a wave per chain (k_assemble_g1_muls_w9: the default for calls of up to 8 requests) is
# ~80 000 cross-lane rendezvous per chain on this interpreter, and most cases here prove a handful of proofs: they take the lane
# per chain instead (the bound is a hooks-build switch), the wave form has its own cases (test_emu_groth16.py) and runs in
# every `-m gpu` case of 8 requests or fewer.
os.environ.setdefault("OG_ASM_W9_MAX", "0")


class Ctx(api.Context):
    _lib = lib

    def __init__(self):
        h = C.c_void_p()
        self._check(self._lib.og_init(0, C.byref(h)))
        self._h = h

    def to_device(self, arr):
        arr = np.array(arr, copy=True, order="C")
        return arr if arr.dtype == np.uint8 else arr.view(np.uint8)

    def to_host(self, buf):
        return buf

    def empty(self, *shape):
        return np.empty(shape, dtype=np.uint8)

    def ptr(self, a):
        if a is None:
            return None
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)

    def _pre(self):
        pass
