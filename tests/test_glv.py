"""og_glv_decompose (owshen_amd/csrc/glv.h): the GLV split k = k1 + lambda k2 (mod r) the proof assembly of latency-bound calls
uses.  lambda and beta are RE-DERIVED here from the two moduli (tools/derive_glv.py: cube roots of unity, matched on the generator
with the oracle's group law), so the constants baked into the library are checked from outside; the halves must be below 2^127.
A host function: no GPU needed."""
import ctypes as C
import random

import pytest

from oracle.py import fields
from tools import derive_glv


@pytest.fixture(scope="module")
def lib():
    from owshen_amd._lib import lib as L
    return L


@pytest.fixture(scope="module")
def consts():
    return derive_glv.derive()


def _decompose(lib, k):
    out = (C.c_uint8 * 32)()
    rc = lib.og_glv_decompose((C.c_uint8 * 32).from_buffer_copy(int(k).to_bytes(32, "little")), out)
    if rc != 0:
        return None
    b = bytes(out)
    halves = []
    for h in (b[:16], b[16:]):
        v = int.from_bytes(h, "little")
        mag, neg = v & ((1 << 127) - 1), v >> 127
        halves.append(-mag if neg else mag)
    return halves


def test_endomorphism_constants_are_what_the_moduli_give(consts):
    assert pow(consts["lam"], 3, fields.R) == 1 and consts["lam"] != 1
    assert pow(consts["beta"], 3, fields.P) == 1 and consts["beta"] != 1
    from oracle.py.curve import G1, G1_GEN
    assert (derive_glv.R, derive_glv.Q) == (fields.R, fields.P)
    assert G1.mul(G1_GEN, consts["lam"]) == (consts["beta"] * G1_GEN[0] % fields.P, G1_GEN[1])   # phi(G) = lambda G, by the oracle's group law
    assert consts["lam"] == 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd            # the values glv.h carries
    assert consts["beta"] == 0x59e26bcea0d48bacd4f263f1acdb5c4f5763473177fffffe


def test_decomposition_is_short_and_exact(lib, consts):
    rnd = random.Random(5)
    ks = [0, 1, 2, fields.R - 1, fields.R - 2, fields.R // 2, consts["lam"], fields.R - consts["lam"], (1 << 253) - 1, 1 << 127, (1 << 128) - 1]
    ks += [rnd.randrange(fields.R) for _ in range(3000)]
    worst = 0
    for k in ks:
        got = _decompose(lib, k)
        assert got is not None, k
        k1, k2 = got
        assert (k1 + consts["lam"] * k2 - k) % fields.R == 0, k
        worst = max(worst, abs(k1).bit_length(), abs(k2).bit_length())
    assert worst <= 127
    # the library rounds down where the textbook rounds to nearest: the halves may differ from derive_glv's by a lattice vector,
    # never in what they represent; on most scalars they coincide
    same = sum(1 for k in ks[11:200] if tuple(_decompose(lib, k)) == derive_glv.decompose(k, consts))
    assert same >= 150


def test_non_canonical_scalars_are_refused(lib):
    assert _decompose(lib, fields.R) is None and _decompose(lib, (1 << 256) - 1) is None
    assert b"not canonical" in lib.og_last_error()
