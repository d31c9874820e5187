"""The 9 x 29-bit Montgomery field layer (owshen_amd/csrc/field.hip.h) driven limb by limb on the CPU
interpreter with adversarial operands: values up to the documented bounds (< 2N, and up to 8N for
products), all-ones limbs, 0 / N / 2N-1, against Python integers."""
import ctypes as C
import random

import pytest

from oracle.py import fields

MASK = (1 << 29) - 1
RR = 1 << 261
MODS = {0: fields.R, 1: fields.P}


@pytest.fixture(scope="module")
def fe():
    from tests import emu
    f = emu.lib.emu_fe_op
    f.restype = None
    A9 = C.c_uint32 * 9

    def call(field, op, a, b=0):
        out = A9()
        f(field, op, A9(*[(a >> (29 * i)) & MASK for i in range(9)]), A9(*[(b >> (29 * i)) & MASK for i in range(9)]), out)
        limbs = list(out)
        return limbs, sum(v << (29 * i) for i, v in enumerate(limbs))
    return call


def _samples(N, rnd, bound_mult=2):
    top = bound_mult * N - 1
    vals = [0, 1, N - 1, N, N + 1, top, top - 1, (1 << 253) - 1, sum(MASK << (29 * i) for i in range(8)) % (bound_mult * N)]
    vals += [rnd.randrange(bound_mult * N) for _ in range(40)]
    return vals


@pytest.mark.parametrize("field", [0, 1])
def test_add_sub_neg_stay_almost_reduced(fe, field):
    N = MODS[field]
    rnd = random.Random(field)
    vals = _samples(N, rnd)
    for a in vals:
        for b in vals[:12] + [rnd.choice(vals)]:
            for op, want in ((0, a + b), (1, a - b)):
                limbs, v = fe(field, op, a, b)
                assert all(x <= MASK for x in limbs) and v < 2 * N and v % N == want % N, (op, a, b)
        limbs, v = fe(field, 5, a)
        assert v < 2 * N and (v + a) % N == 0


@pytest.mark.parametrize("field", [0, 1])
def test_mul_sqr_bounds_and_values(fe, field):
    N = MODS[field]
    rnd = random.Random(10 + field)
    rinv = pow(RR, -1, N)
    vals = _samples(N, rnd, 8)  # products tolerate operands up to 8N
    for a in vals:
        for b in vals[:10] + [rnd.choice(vals)]:
            limbs, v = fe(field, 2, a, b)
            assert all(x <= MASK for x in limbs) and v < 2 * N and v == v % (2 * N) and v % N == a * b * rinv % N, (a, b)
        limbs, v = fe(field, 3, a)
        assert all(x <= MASK for x in limbs) and v < 2 * N and v % N == a * a * rinv % N


@pytest.mark.parametrize("field", [0, 1])
def test_from_mont_is_canonical_and_words_roundtrip(fe, field):
    N = MODS[field]
    rnd = random.Random(20 + field)
    rinv = pow(RR, -1, N)
    for a in _samples(N, rnd):
        _, v = fe(field, 4, a)
        assert v < N and v == a * rinv % N
        _, w = fe(field, 7, a)
        assert w == a
    for a in (0, N, 5, N + 5):
        limbs, _ = fe(field, 8, a, a % N)
        assert limbs[0] == 1 and limbs[1] == (1 if a % N == 0 else 0)


@pytest.mark.parametrize("field", [0, 1])
def test_inverse(fe, field):
    N = MODS[field]
    rnd = random.Random(30 + field)
    for a in [1, N - 1, N + 3] + [rnd.randrange(1, 2 * N) for _ in range(4)]:
        if a % N == 0:
            continue
        _, v = fe(field, 6, a)  # Montgomery inverse: v = a^-1 R^2 ... in Montgomery terms (aR)^-1 R
        am = a  # treat a as the Montgomery form of x = a R^-1
        x = am * pow(RR, -1, N) % N
        assert v < 2 * N and v % N == pow(x, -1, N) * RR % N


@pytest.mark.parametrize("field", [0, 1])
def test_fused_products_with_lazy_operands(fe, field):
    """fe_mul_add / fe_mul_add4 / lazy negation and doubling: worst-case column sums (all-ones limbs) stay exact"""
    N = MODS[field]
    rnd = random.Random(40 + field)
    rinv = pow(RR, -1, N)
    vals = _samples(N, rnd)  # normalized operands < 2N
    for a in vals:
        for b in vals[:10] + [rnd.choice(vals)]:
            limbs, v = fe(field, 9, a, b)
            assert all(x <= MASK for x in limbs) and v < 2 * N and v % N == (a * b + (4 * N - b) * a) * rinv % N
            limbs, v = fe(field, 10, a, b)
            assert all(x <= MASK for x in limbs) and v < 2 * N and v % N == (2 * a * b - a * a - b * b) * rinv % N
            limbs, v = fe(field, 11, a, b)
            assert all(x <= MASK for x in limbs) and v < 2 * N and v % N == 2 * a * b * rinv % N


@pytest.mark.parametrize("field", [0, 1])
def test_weak_forms(fe, field):
    """carry-only subtraction / a + 2b / the 3N-4N-5N zero test / 4N-based lazy negation"""
    N = MODS[field]
    rnd = random.Random(50 + field)
    rinv = pow(RR, -1, N)
    vals = _samples(N, rnd)  # < 2N
    for a in vals:
        for b in vals[:12] + [rnd.choice(vals), a, (a + N) % (2 * N)]:
            limbs, v = fe(field, 12, a, b)
            assert all(x <= MASK for x in limbs) and v == a - b + 4 * N
            limbs, v = fe(field, 13, a, b)
            assert all(x <= MASK for x in limbs) and v == a + 2 * b
            limbs, _ = fe(field, 14, a, b)
            assert limbs[0] == (1 if (a - b) % N == 0 else 0), (a, b)
            limbs, v = fe(field, 15, a, b)
            assert all(x <= MASK for x in limbs) and v < 2 * N and v % N == (a * a + (4 * N - b) * a) * rinv % N
    # the weak difference feeds products with operands up to 6N / 10N: worst-case magnitudes stay exact
    big = [6 * N - 1, 10 * N - 1, 6 * N - 12345]
    for a in big:
        for b in big:
            limbs, v = fe(field, 2, a, b)
            assert v < 2 * N and v % N == a * b * rinv % N


def test_fq2_products_with_weak_and_saturated_operands():
    """Karatsuba-in-columns Fq2 product, squaring and the fused forms at the documented operand bounds: weak values up to
    6N / 10N and limb patterns saturated at 2^29 - 1 (the worst case for the 64-bit column sums)."""
    from tests import emu
    f = emu.lib.emu_fq2_op
    f.restype = None
    A18 = C.c_uint32 * 18
    N = fields.P
    rinv = pow(RR, -1, N)

    def pack(x):
        return A18(*([(x[0] >> (29 * i)) & MASK for i in range(9)] + [(x[1] >> (29 * i)) & MASK for i in range(9)]))

    def call(op, a, b=(0, 0), c=(0, 0), d=(0, 0)):
        out = A18()
        f(op, pack(a), pack(b), pack(c), pack(d), out)
        v = list(out)
        assert all(x <= MASK for x in v)
        r0 = sum(x << (29 * i) for i, x in enumerate(v[:9]))
        r1 = sum(x << (29 * i) for i, x in enumerate(v[9:]))
        assert r0 < 2 * N and r1 < 2 * N
        return r0 % N, r1 % N

    def mul(x, y):
        return ((x[0] * y[0] - x[1] * y[1]) % N, (x[0] * y[1] + x[1] * y[0]) % N)

    def scale(x):
        return (x[0] * rinv % N, x[1] * rinv % N)
    rnd = random.Random(77)
    sat = lambda top: (top << 232) | ((1 << 232) - 1)   # every limb below the top saturated
    w6, w10, w2 = (6 * N) >> 232, (10 * N) >> 232, (2 * N) >> 232
    big6 = [6 * N - 1, sat(w6 - 1), 6 * N - rnd.randrange(1 << 200)]
    big2 = [2 * N - 1, sat(w2 - 1), rnd.randrange(2 * N)]
    big10 = [10 * N - 1, sat(w10 - 1)]
    for a0 in big6:
        for a1 in big6[:2]:
            a = (a0, a1)
            assert call(1, a) == scale(mul(a, a))
            for b in [(big2[0], big2[1]), (big6[1], big6[0]), (big2[2], big2[2])]:
                assert call(0, a, b) == scale(mul(a, b))          # a may be weak when it is the second operand's partner
                assert call(0, b, a) == scale(mul(b, a))
            # f_mul_sub(R, D, Y1, PPP): a, b < 6N; c, d < 2N
            c, d = (big2[0], big2[1]), (big2[1], big2[2])
            bb = (big6[2], big6[1])
            want = scale(tuple((u - v) % N for u, v in zip(mul(a, bb), mul(c, d))))
            assert call(2, a, bb, c, d) == want
            # f_sqr_sub(R, PP, W): a < 6N, c < 2N, d < 10N
            dd = (big10[0], big10[1])
            want = scale(tuple((u - v) % N for u, v in zip(mul(a, a), mul(c, dd))))
            assert call(3, a, (0, 0), c, dd) == want


@pytest.mark.parametrize("field", [0, 1])
def test_mul_plus_equals_mul_then_weak_sub(fe, field):
    """fe_mul_plus (the subtrahend rides in the high columns of the Montgomery reduction): limb-for-limb the value of
    fe_mul followed by the weak subtraction, for operands at the bounds the group law uses (a < 2N normalized; x < 2N)"""
    N = MODS[field]
    rnd = random.Random(30 + field)
    vals = _samples(N, rnd)
    for a in vals:
        for b in vals[:10] + [rnd.choice(vals)]:
            _, sq = fe(field, 3, a)                      # a^2 / R, < 2N
            limbs, v = fe(field, 16, a, b)
            assert all(x <= MASK for x in limbs) and v == sq + 4 * N - b, (a, b)
            _, want = fe(field, 12, sq, b)               # fe_sub_weak(a^2 / R, b)
            assert v == want
            limbs, v = fe(field, 17, a, b)               # a lazily negated multiplication operand + the same as addend
            rinv = pow(RR, -1, N)
            assert all(x <= MASK for x in limbs) and v < 6 * N and v % N == ((4 * N - a) * b * rinv + 4 * N - a) % N


@pytest.mark.parametrize("field", [0, 1])
def test_mul_with_one_operand_of_limbs_up_to_2_31(field):
    """ADVICE r4 / field.hip.h's fe_mul contract: ONE operand may have limbs up to 2^31 and a value up to 42 N (the lazy butterflies
    of the radix-4 NTT).  The HOST form of the column walk (what og_verify and this interpreter run; the gfx950 asm text gets the
    same operands in tests/test_mont_asm.py) with all limbs at the bound against the largest normalized partners: exact
    Montgomery products, normalized, < 2N"""
    from tests import emu
    from tests.test_mont_asm import worst_lazy31
    f = emu.lib.emu_fe_op
    f.restype = None
    A9 = C.c_uint32 * 9
    N = MODS[field]
    rnd = random.Random(31 + field)
    lim = lambda v: [(v >> (29 * i)) & MASK for i in range(9)]
    val = lambda l: sum(x << (29 * i) for i, x in enumerate(l))
    worst = worst_lazy31(N)
    lazies = [worst] + [[rnd.randrange(1 << 31) for _ in range(8)] + [rnd.randrange(worst[8] + 1)] for _ in range(30)]
    partners = [lim(2 * N - 1), [MASK] * 8 + [lim(2 * N - 1)[8]], lim(0), lim(1)] + [lim(rnd.randrange(2 * N)) for _ in range(20)]
    for la in lazies:
        for pb in partners:
            for x, y in ((la, pb), (pb, la)):
                out = A9()
                f(field, 2, A9(*x), A9(*y), out)
                total = val(x) * val(y)
                m = (-total * pow(N, -1, RR)) % RR
                got = list(out)
                assert all(v <= MASK for v in got) and val(got) == (total + m * N) // RR and val(got) < 2 * N, (x, y)
