"""Radix-2 Fr NTT and the Groth16 H-polynomial pipeline.  TEST INFRASTRUCTURE ONLY.

No reference counterpart; the field is the reference's ``Fp``
(``.../babyjubjub/mod.rs:7-11``: generator 7 => 2^k-th roots of unity
w_k = 7^((r-1)/2^k)).  Coset generator for the quotient evaluation is g = 7
(arkworks' ``GENERATOR`` convention).
"""
from .fields import R, inv, fr_root_of_unity, FR_GENERATOR


def dft_naive(a, omega):
    n = len(a)
    return [sum(a[j] * pow(omega, i * j, R) for j in range(n)) % R for i in range(n)]


def _bitrev(a):
    n = len(a)
    lg = n.bit_length() - 1
    out = list(a)
    for i in range(n):
        j = int(bin(i)[2:].zfill(lg)[::-1], 2) if lg else 0
        if j > i:
            out[i], out[j] = out[j], out[i]
    return out


def ntt(a, inverse=False):
    """natural order in, natural order out; evaluates at w^i (w^-i and 1/n if inverse)."""
    n = len(a)
    assert n & (n - 1) == 0
    lg = n.bit_length() - 1
    w = fr_root_of_unity(lg)
    if inverse:
        w = inv(w, R)
    a = _bitrev([x % R for x in a])
    m = 1
    while m < n:
        wm = pow(w, n // (2 * m), R)
        for s in range(0, n, 2 * m):
            t = 1
            for j in range(m):
                u = a[s + j]
                v = a[s + j + m] * t % R
                a[s + j] = (u + v) % R
                a[s + j + m] = (u - v) % R
                t = t * wm % R
        m *= 2
    if inverse:
        ninv = inv(n, R)
        a = [x * ninv % R for x in a]
    return a


def coset_ntt(a, g=FR_GENERATOR):
    """evaluate polynomial with coefficients a at g*w^i."""
    gi = 1
    out = []
    for x in a:
        out.append(x * gi % R)
        gi = gi * g % R
    return ntt(out)


def coset_intt(e, g=FR_GENERATOR):
    """inverse of coset_ntt."""
    a = ntt(e, inverse=True)
    ginv = inv(g, R)
    gi = 1
    out = []
    for x in a:
        out.append(x * gi % R)
        gi = gi * ginv % R
    return out


def h_poly(a_evals, b_evals, c_evals, g=FR_GENERATOR):
    """h = (A*B - C)/Z over a size-d domain: 3 iNTT, 3 coset NTT, pointwise, 1 coset iNTT.

    Returns the d coefficients of h (deg <= d-2, so the last is 0)."""
    d = len(a_evals)
    ac = ntt(a_evals, inverse=True)
    bc = ntt(b_evals, inverse=True)
    cc = ntt(c_evals, inverse=True)
    ae = coset_ntt(ac, g)
    be = coset_ntt(bc, g)
    ce = coset_ntt(cc, g)
    zinv = inv((pow(g, d, R) - 1) % R, R)  # Z(g w^i) = g^d - 1, constant on the coset
    he = [(x * y - z) % R * zinv % R for x, y, z in zip(ae, be, ce)]
    return coset_intt(he, g)
