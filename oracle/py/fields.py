"""BN254 base/scalar fields on Python ints.  TEST INFRASTRUCTURE ONLY.

Fr is the reference's ``Fp``:
``/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11``
(modulus, generator 7, 32-byte little-endian repr).  Fq and the tower
Fq2/Fq6/Fq12 follow EIP-196/197 (alt_bn128); the reference has no such code
 -- parity unpinned.
"""

# scalar field (reference Fp, mod.rs:8)
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
# base field (EIP-196)
P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
# BN parameter x: P = 36x^4+36x^3+24x^2+6x+1, R = 36x^4+36x^3+18x^2+6x+1
BN_X = 4965661367192848881
FR_GENERATOR = 7  # mod.rs:9
FR_TWO_ADICITY = 28

assert P == 36 * BN_X**4 + 36 * BN_X**3 + 24 * BN_X**2 + 6 * BN_X + 1
assert R == 36 * BN_X**4 + 36 * BN_X**3 + 18 * BN_X**2 + 6 * BN_X + 1
assert (R - 1) % (1 << FR_TWO_ADICITY) == 0 and ((R - 1) >> FR_TWO_ADICITY) & 1


def inv(a, m):
    return pow(a, -1, m)


def fr_to_bytes(x):
    """32-byte little-endian canonical repr (mod.rs:10)."""
    return int(x % R).to_bytes(32, "little")


def fr_from_bytes(b):
    x = int.from_bytes(b, "little")
    if x >= R:
        raise ValueError("non-canonical Fr")
    return x


def fq_to_bytes(x):
    return int(x % P).to_bytes(32, "little")


def fq_from_bytes(b):
    x = int.from_bytes(b, "little")
    if x >= P:
        raise ValueError("non-canonical Fq")
    return x


def fr_root_of_unity(log_n):
    """Primitive 2^log_n-th root of unity derived from generator 7."""
    assert 0 <= log_n <= FR_TWO_ADICITY
    return pow(FR_GENERATOR, (R - 1) >> log_n, R)


# ---- Fq2 = Fq[u]/(u^2+1), elements are (c0, c1) ---------------------------

def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return ((-a[0]) % P, (-a[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sqr(a):
    return ((a[0] + a[1]) * (a[0] - a[1]) % P, 2 * a[0] * a[1] % P)


def f2_scale(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f2_inv(a):
    d = inv((a[0] * a[0] + a[1] * a[1]) % P, P)
    return (a[0] * d % P, (-a[1]) * d % P)


def f2_conj(a):
    return (a[0], (-a[1]) % P)


F2_ZERO = (0, 0)
F2_ONE = (1, 0)
XI = (9, 1)  # non-residue for the Fq6 tower, EIP-197


def f2_mul_xi(a):
    # (a0 + a1 u)(9 + u) = 9a0 - a1 + (a0 + 9a1) u
    return ((9 * a[0] - a[1]) % P, (a[0] + 9 * a[1]) % P)


# ---- Fq6 = Fq2[v]/(v^3 - xi), elements are (c0, c1, c2) of Fq2 -------------

F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)


def f6_add(a, b):
    return (f2_add(a[0], b[0]), f2_add(a[1], b[1]), f2_add(a[2], b[2]))


def f6_sub(a, b):
    return (f2_sub(a[0], b[0]), f2_sub(a[1], b[1]), f2_sub(a[2], b[2]))


def f6_neg(a):
    return (f2_neg(a[0]), f2_neg(a[1]), f2_neg(a[2]))


def f6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    t0 = f2_mul(a0, b0)
    t1 = f2_mul(a1, b1)
    t2 = f2_mul(a2, b2)
    c0 = f2_add(t0, f2_mul_xi(f2_sub(f2_mul(f2_add(a1, a2), f2_add(b1, b2)), f2_add(t1, t2))))
    c1 = f2_add(f2_sub(f2_mul(f2_add(a0, a1), f2_add(b0, b1)), f2_add(t0, t1)), f2_mul_xi(t2))
    c2 = f2_add(f2_sub(f2_mul(f2_add(a0, a2), f2_add(b0, b2)), f2_add(t0, t2)), t1)
    return (c0, c1, c2)


def f6_mul_v(a):
    """multiply by v: (c0,c1,c2) -> (xi*c2, c0, c1)"""
    return (f2_mul_xi(a[2]), a[0], a[1])


def f6_inv(a):
    a0, a1, a2 = a
    t0 = f2_sub(f2_sqr(a0), f2_mul_xi(f2_mul(a1, a2)))
    t1 = f2_sub(f2_mul_xi(f2_sqr(a2)), f2_mul(a0, a1))
    t2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    d = f2_add(f2_mul(a0, t0), f2_mul_xi(f2_add(f2_mul(a2, t1), f2_mul(a1, t2))))
    di = f2_inv(d)
    return (f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di))


# ---- Fq12 = Fq6[w]/(w^2 - v), elements are (c0, c1) of Fq6 -----------------

F12_ONE = (F6_ONE, F6_ZERO)


def f12_mul(a, b):
    t0 = f6_mul(a[0], b[0])
    t1 = f6_mul(a[1], b[1])
    c0 = f6_add(t0, f6_mul_v(t1))
    c1 = f6_sub(f6_mul(f6_add(a[0], a[1]), f6_add(b[0], b[1])), f6_add(t0, t1))
    return (c0, c1)


def f12_sqr(a):
    return f12_mul(a, a)


def f12_conj(a):
    return (a[0], f6_neg(a[1]))


def f12_inv(a):
    t = f6_sub(f6_mul(a[0], a[0]), f6_mul_v(f6_mul(a[1], a[1])))
    ti = f6_inv(t)
    return (f6_mul(a[0], ti), f6_neg(f6_mul(a[1], ti)))


def f12_pow(a, e):
    r = F12_ONE
    b = a
    while e:
        if e & 1:
            r = f12_mul(r, b)
        b = f12_sqr(b)
        e >>= 1
    return r
