"""The host's Fr arithmetic (owshen_amd/csrc/host_fr4.h: four 64-bit limbs, R' = 2^256 -- what og_set_host_chains walks the MiMC7
chains with) against Python integers: the constants it DERIVES from FrParams::N, and add / Montgomery product / to / from on edge
values and random ones."""
import ctypes as C
import random

import pytest

from oracle.py import fields

N = fields.R
RP = 1 << 256
A4, A13 = C.c_uint64 * 4, C.c_uint64 * 13


@pytest.fixture(scope="module")
def h4():
    from tests import emu
    f = emu.lib.emu_h4_op
    f.restype = None

    def limbs(v):
        return A4(*[(v >> (64 * i)) & (2**64 - 1) for i in range(4)])

    def call(op, a=0, b=0):
        out = A13()
        f(op, limbs(a), limbs(b), out)
        vals = list(out)
        return vals if op == 4 else sum(x << (64 * i) for i, x in enumerate(vals[:4]))
    return call


def test_derived_constants(h4):
    v = h4(4)
    n = sum(x << (64 * i) for i, x in enumerate(v[0:4]))
    assert n == N
    assert v[4] == (-pow(N, -1, 1 << 64)) % (1 << 64)
    assert sum(x << (64 * i) for i, x in enumerate(v[5:9])) == RP % N
    assert sum(x << (64 * i) for i, x in enumerate(v[9:13])) == RP * RP % N


def test_add_mul_to_from(h4):
    rnd = random.Random(4)
    rinv = pow(RP, -1, N)
    vals = [0, 1, 2, N - 1, N - 2, (N - 1) // 2, (1 << 253) - 1, (1 << 64) - 1, 1 << 64, (1 << 192) + 5] + [rnd.randrange(N) for _ in range(60)]
    for a in vals:
        for b in vals[:10] + [rnd.choice(vals)]:
            assert h4(0, a, b) == (a + b) % N, (a, b)
            assert h4(1, a, b) == a * b * rinv % N, (a, b)
        assert h4(3, a) == a * rinv % N
    for a in vals + [N, N + 1, 2 * N + 3, RP - 1, RP - N, (1 << 255) + 12345]:   # to_mont reduces any 256-bit value (what fe_to_mont does on the device)
        assert h4(2, a) == a * RP % N, a
