"""Restatement of the reference's BabyJubJub file -- the ONLY real anchor.

TEST INFRASTRUCTURE ONLY.  Follows, function by function,
``/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs``:
constants :174-189, affine add :28-44, double :56-67, multiply :68-78,
compress/decompress :82-98, verify :99-115, projective add/double :118-164,
placeholder hash :202-204, sign :210-236.  Used to pin the Fr modulus, generator
and little-endian repr that cross the C-ABI seam.
"""
from .fields import R, inv

A = 168700
D = 168696
BASE = (5299619240641551281634865583518297030282874472190772894086521144482721001553,
        16950150798460657717958625567821834550301663161624707787222815936182638968203)
ORDER = 21888242871839275222246405745257275088614511777268538073601725287587578984328


def is_on_curve(p):
    x, y = p
    return (y * y + A * x * x) % R == (1 + D * x * x % R * y * y) % R


def affine_double(p):  # mod.rs:56-67
    x, y = p
    xx = inv((A * x * x + y * y) % R, R)
    yy = inv((2 - A * x * x - y * y) % R, R)
    return (2 * x * y * xx % R, (y * y - A * x * x) * yy % R)


def affine_add(p, q):  # mod.rs:28-44
    if p == q:
        return affine_double(p)
    t = D * p[0] * q[0] % R * p[1] * q[1] % R
    xx = inv((1 + t) % R, R)
    yy = inv((1 - t) % R, R)
    return ((p[0] * q[1] + p[1] * q[0]) * xx % R, (p[1] * q[1] - A * p[0] * q[0]) * yy % R)


def proj_zero():
    return (0, 1, 0)


def proj_to_affine(p):  # mod.rs:165-171
    if p[2] % R == 0:
        return (0, 1)
    zi = inv(p[2], R)
    return (p[0] * zi % R, p[1] * zi % R)


def proj_double(p):  # mod.rs:152-164
    if p[2] % R == 0:
        return proj_zero()
    X, Y, Z = p
    b = (X + Y) ** 2 % R
    c = X * X % R
    d = Y * Y % R
    e = A * c % R
    f = (e + d) % R
    h = Z * Z % R
    j = (f - 2 * h) % R
    return ((b - c - d) * j % R, f * (e - d) % R, f * j % R)


def proj_add(p, q):  # mod.rs:118-143
    if p[2] % R == 0:
        return q
    if q[2] % R == 0:
        return p
    if proj_to_affine(p) == proj_to_affine(q):
        return proj_double(p)
    a = p[2] * q[2] % R
    b = a * a % R
    c = p[0] * q[0] % R
    d = p[1] * q[1] % R
    e = D * c % R * d % R
    f = (b - e) % R
    g = (b + e) % R
    return (a * f % R * ((p[0] + p[1]) * (q[0] + q[1]) - c - d) % R,
            a * g % R * (d - A * c) % R,
            f * g % R)


def multiply(p, scalar):  # mod.rs:68-78 (MSB-first over the 254-bit repr)
    res = proj_zero()
    pp = (p[0], p[1], 1)
    for i in reversed(range(254)):
        res = proj_double(res)
        if (scalar >> i) & 1:
            res = proj_add(res, pp)
    return proj_to_affine(res)


def fr_sqrt(a):
    """Tonelli-Shanks in Fr (ff's sqrt; either root -- caller fixes parity)."""
    a %= R
    if a == 0:
        return 0
    if pow(a, (R - 1) // 2, R) != 1:
        return None
    s, q = 28, (R - 1) >> 28
    z = pow(7, q, R)
    m, c, t, r = s, z, pow(a, q, R), pow(a, (q + 1) // 2, R)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % R
            i += 1
        b = pow(c, 1 << (m - i - 1), R)
        m, c, t, r = i, b * b % R, t * b * b % R, r * b % R
    return r


def compress(p):  # mod.rs:82-84
    return (p[0], bool(p[1] & 1))


def decompress(c):  # mod.rs:88-98
    x, odd = c
    i = inv((1 - D * x * x) % R, R)
    y = fr_sqrt(i * (1 - A * x * x) % R)
    if y is None:
        raise ValueError("Cannot take sqrt")
    if bool(y & 1) != odd:
        y = (-y) % R
    return (x, y)


def placeholder_hash(inp):  # mod.rs:202-204 -- product of inputs; NOT MiMC7
    r = 1
    for v in inp:
        r = r * v % R
    return r


def sign(sk, randomness, message):  # mod.rs:210-236
    pk = decompress(compress(multiply(BASE, sk)))
    r = placeholder_hash([randomness, message])
    rr = multiply(BASE, r)
    h = placeholder_hash([rr[0], rr[1], pk[0], pk[1], message])
    s = (r + h * sk) % ORDER
    if s >= R:
        raise ValueError("Invalid repr")
    return (rr, s)


def verify(pk_compressed, message, sig):  # mod.rs:99-115
    pk = decompress(pk_compressed)
    rr, s = sig
    if not is_on_curve(pk) or not is_on_curve(rr):
        return False
    h = placeholder_hash([rr[0], rr[1], pk[0], pk[1], message])
    sb = multiply(BASE, s)
    r_plus_ha = affine_add(multiply(pk, h), rr)
    return r_plus_ha == sb


# ---- the same scheme on the real hash: MultiMiMC7 replaces the placeholder product --------------
def mimc7_hash(inp):
    from . import mimc7
    return mimc7.multi_hash([v % R for v in inp], 0)


def sign_mimc7(sk, randomness, message):  # mod.rs:210-236 with hash := MultiMiMC7
    pk = decompress(compress(multiply(BASE, sk)))
    r = mimc7_hash([randomness, message])
    rr = multiply(BASE, r)
    h = mimc7_hash([rr[0], rr[1], pk[0], pk[1], message])
    s = (r + h * sk) % ORDER
    if s >= R:
        raise ValueError("Invalid repr")
    return (rr, s)


def verify_mimc7(pk, message, sig):  # mod.rs:99-115 with hash := MultiMiMC7; pk affine
    rr, s = sig
    if not is_on_curve(pk) or not is_on_curve(rr):
        return False
    h = mimc7_hash([rr[0], rr[1], pk[0], pk[1], message])
    return affine_add(multiply(pk, h), rr) == multiply(BASE, s)
