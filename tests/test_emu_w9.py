"""The wave-wide ("w9") Montgomery product (owshen_amd/csrc/field_w9.hip.h: one element over nine lanes of a wave) on the CPU
interpreter: raw limbs at the documented bounds (limbs < 2^31 on both sides, a * b < 169 N^2) against Python integers and
against the lane-local fe_mul, and the chained form through the C ABI (og_field_mulchain_lat_d) against og_field_mulchain_d."""
import ctypes as C
import random

import numpy as np
import pytest

from oracle.py import fields

MASK = (1 << 29) - 1
RR = 1 << 261
MODS = {0: fields.R, 1: fields.P}
A9, A10 = C.c_uint32 * 9, C.c_uint32 * 10


def _limbs(v):
    return [(v >> (29 * i)) & MASK for i in range(9)]


def _lazy(v, rnd):
    """limbs of v with carries pushed DOWN: limb i + k 2^29, limb i + 1 - k, as far as 2^31 - 1 allows"""
    l = _limbs(v)
    for i in range(8):
        k = min(l[i + 1], ((1 << 31) - 1 - l[i]) >> 29, rnd.randrange(4))
        l[i] += k << 29
        l[i + 1] -= k
    assert sum(x << (29 * i) for i, x in enumerate(l)) == v and all(x < (1 << 31) for x in l)
    return l


@pytest.fixture(scope="module")
def w9():
    from tests import emu
    f = emu.lib.emu_w9_mul
    f.restype = None
    g = emu.lib.emu_fe_op
    g.restype = None

    def call(field, a, b):
        out = A10()
        f(field, A9(*a), A9(*b), out)
        assert out[9] == 0, "a lane above the ninth is not zero"
        return list(out)[:9]

    def ref(field, a, b):
        out = A9()
        g(field, 2, A9(*a), A9(*b), out)
        return list(out)
    return call, ref


@pytest.mark.parametrize("field", [0, 1])
def test_w9_mul_value_and_limb_bounds(w9, field):
    call, ref = w9
    N = MODS[field]
    rinv = pow(RR, -1, N)
    rnd = random.Random(60 + field)
    vals = [0, 1, N - 1, N, 2 * N - 1, 8 * N - 1, 13 * N - 1, (1 << 253) - 1] + [rnd.randrange(13 * N) for _ in range(24)]
    for a in vals:
        for b in vals[:8] + [rnd.choice(vals)]:
            for la, lb in ((_limbs(a), _limbs(b)), (_lazy(a, rnd), _lazy(b, rnd))):
                r = call(field, la, lb)
                v = sum(x << (29 * i) for i, x in enumerate(r))
                assert all(x < (1 << 29) + 32 for x in r), (a, b, r)
                assert v < 2 * N and v % N == a * b * rinv % N, (a, b)
    # the same Montgomery digits as the lane-local product: identical VALUES (not only mod N) for operands fe_mul accepts
    for _ in range(40):
        a, b = rnd.randrange(8 * N), rnd.randrange(2 * N)
        r = call(field, _limbs(a), _limbs(b))
        want = ref(field, _limbs(a), _limbs(b))
        assert sum(x << (29 * i) for i, x in enumerate(r)) == sum(x << (29 * i) for i, x in enumerate(want))


@pytest.mark.parametrize("field", [0, 1])
@pytest.mark.parametrize("lazy", [0, 1])
def test_w9_mul_two_rows_and_the_32_bit_digit(field, lazy):
    """w9_mul<M, ROWS = true, LAZY>: rows 0 and 1 of the wave multiply the same uniform a by two different b, each row with its own
    Montgomery digit (DPP row_newbcast on the GPU); LAZY keeps all 32 bits of the digit: the same residue, below a b / (169 N) +
    8.01 N instead of 2 N (169 = 2^261 / N, rounded down), limbs below 2^29 + 32 either way, every other lane zero.  Operands up to
    the bounds a MiMC7 round built from it reaches (mimc7.hip.h w9_mimc7_round: t < 21 N, limbs < 2^31 + 2^9)"""
    from tests import emu
    f = emu.lib.emu_w9_mul_rows
    f.restype = None
    A19 = C.c_uint32 * 19
    N = MODS[field]
    rinv = pow(RR, -1, N)
    rnd = random.Random(70 + 2 * field + lazy)
    top = 21 * N
    vals = [0, 1, N - 1, N, 2 * N - 1, top - 1] + [rnd.randrange(top) for _ in range(20)]
    for a in vals:
        for _ in range(3):
            b0, b1 = rnd.choice(vals), rnd.choice(vals)
            if lazy == 0 and (a * b0 >= 169 * N * N or a * b1 >= 169 * N * N):
                continue
            la, l0, l1 = _lazy(a, rnd), _lazy(b0, rnd), _lazy(b1, rnd)
            out = A19()
            f(field, lazy, A9(*la), A9(*l0), A9(*l1), out)
            assert out[18] == 0, "a lane outside the two rows' nine limbs is not zero"
            for b, r in ((b0, list(out)[:9]), (b1, list(out)[9:18])):
                v = sum(x << (29 * i) for i, x in enumerate(r))
                assert all(x < (1 << 29) + 32 for x in r), (a, b, r)
                assert v % N == a * b * rinv % N, (a, b)
                bound = a * b // (169 * N) + 8 * N + N // 50 if lazy else 2 * N
                assert v < bound, (a, b, v // N)


@pytest.mark.parametrize("field", [0, 1])
def test_w9_chain_equals_lane_local_chain(field):
    from tests import emu
    ctx = emu.Ctx()
    try:
        rnd = random.Random(70 + field)
        N = MODS[field]
        n, iters = 3, 5
        xs = [rnd.randrange(N) for _ in range(n)]
        ys = [rnd.randrange(N) for _ in range(n)]
        to = lambda vs: np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vs), dtype=np.uint8).reshape(len(vs), 32)
        y = ctx.to_device(to(ys))
        x0 = ctx.to_device(to(xs))
        ctx.field_mulchain(field, x0, y, iters)
        for form in (0, 1):
            x = ctx.to_device(to(xs))
            ctx.field_mulchain_lat(field, form, x, y, iters)
            assert bytes(x) == bytes(x0), form
        rinv = pow(RR, -1, N)
        for i in range(n):
            v = int.from_bytes(bytes(x0[i]), "little")
            assert v % N == xs[i] * pow(ys[i] * rinv, iters, N) % N
    finally:
        ctx.close()
