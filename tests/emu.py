"""TEST INFRASTRUCTURE ONLY -- numpy front end to tests/hipemu/_build/libowshen_emu.so.

That library is the SAME kernel source as the product (owshen_amd/csrc/*.hip) compiled for a
single-threaded CPU interpreter (tests/hipemu/hip/hip_runtime.h), so kernel and prover-glue
logic can be checked against the oracle where no GPU exists.  owshen_amd never loads it; the
`-m gpu` tests are the parity tests proper and run the gfx950 binary.  "Device" buffers here
are plain numpy arrays.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from owshen_amd._abi import bind

_HERE = os.path.dirname(os.path.abspath(__file__))
_DIR = os.path.join(_HERE, "hipemu")
_SO = os.path.join(_DIR, "_build", "libowshen_emu.so")


def _load():
    subprocess.check_call(["make", "-s", "-j8", "-C", _DIR])
    return bind(C.CDLL(_SO))


lib = _load()


class EmuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libowshen_emu error {code}: {msg}")
        self.code = code


def check(code):
    if code != 0:
        raise EmuError(code, lib.og_last_error().decode("utf-8", "replace"))


def p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return C.c_void_p(a.ctypes.data)


def u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


class Ctx:
    def __init__(self):
        h = C.c_void_p()
        check(lib.og_init(0, C.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib.og_shutdown(self.h)
            self.h = None

    def field_op(self, field, op, a, b=None):
        a = u8(a)
        b = a if b is None else u8(b)
        out = np.empty_like(a)
        check(lib.og_field_op_d(self.h, field, {"add": 0, "sub": 1, "mul": 2, "inv": 3}[op], p(a), p(b), p(out), a.shape[0]))
        return out

    def mimc7_hash2(self, l, r):
        l, r = u8(l), u8(r)
        out = np.empty_like(l)
        check(lib.og_mimc7_hash2_d(self.h, p(l), p(r), p(out), l.shape[0]))
        return out

    def ntt(self, x, inverse=False, coset=False):
        x = u8(x)
        x3 = x if x.ndim == 3 else x[None]
        out = np.empty_like(x3)
        check(lib.og_ntt_fr_d(self.h, p(x3), p(out), x3.shape[1].bit_length() - 1, x3.shape[0], int(inverse), int(coset)))
        return out.reshape(x.shape)

    def h_poly(self, a, b, c):
        a, b, c = u8(a), u8(b), u8(c)
        a3 = a if a.ndim == 3 else a[None]
        out = np.empty_like(a3)
        check(lib.og_h_poly_d(self.h, p(a), p(b), p(c), a3.shape[1].bit_length() - 1, a3.shape[0], p(out)))
        return out.reshape(a.shape)

    def bases(self, group, points, window_bits=0, precompute=False):
        points = u8(points)
        h = C.c_void_p()
        check(lib.og_bases_create_d(self.h, group, p(points), points.shape[0], window_bits, int(precompute), C.byref(h)))
        return h

    def msm(self, bases_h, group, scalars, n=None):
        scalars = u8(scalars)
        s3 = scalars if scalars.ndim == 3 else scalars[None]
        batch, nn = s3.shape[0], s3.shape[1]
        n = nn if n is None else n
        out = np.zeros((batch, 64 if group == 1 else 128), dtype=np.uint8)
        check(lib.og_msm_d(self.h, bases_h, p(s3), n, batch, nn * 32, p(out)))
        return out
