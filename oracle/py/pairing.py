"""BN254 optimal-ate pairing (EIP-197 equation).  TEST INFRASTRUCTURE ONLY.

Stands in for the Solidity verifier that ``north_star`` names but the reference
does not contain (``/root/reference/contracts/src/Owshen.sol:66-78`` is an ECDSA
gate).  ``pairing_check`` evaluates exactly what the
0x08 precompile evaluates: prod e(P_i, Q_i) == 1.
Tower: Fq2 = Fq[u]/(u^2+1), Fq6 = Fq2[v]/(v^3-(9+u)), Fq12 = Fq6[w]/(w^2-v).
"""
from .fields import (P, R, BN_X, f2_add, f2_sub, f2_mul, f2_sqr, f2_neg, f2_inv, f2_scale, f2_conj,
                     F2_ZERO, F2_ONE, XI, F6_ZERO, F12_ONE, f12_mul, f12_sqr, f12_pow)
from .curve import G1, G2

ATE_LOOP = 6 * BN_X + 2
FINAL_EXP = (P**12 - 1) // R


def _f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r


_G12 = _f2_pow(XI, (P - 1) // 3)
_G13 = _f2_pow(XI, (P - 1) // 2)
_G22 = _f2_pow(XI, (P * P - 1) // 3)
_G23 = _f2_pow(XI, (P * P - 1) // 2)


def _frob_twist(q):
    return (f2_mul(f2_conj(q[0]), _G12), f2_mul(f2_conj(q[1]), _G13))


def _frob2_twist(q):
    return (f2_mul(q[0], _G22), f2_mul(q[1], _G23))


def _line(t, lam, p):
    """line through twist point t with twist-slope lam, evaluated at G1 point p."""
    xp, yp = p
    c0 = ((yp % P, 0), F2_ZERO, F2_ZERO)
    c1 = (f2_neg(f2_scale(lam, xp)), f2_sub(f2_mul(lam, t[0]), t[1]), F2_ZERO)
    return (c0, c1)


def _vertical(t, p):
    # x_P - x_T w^2  (only reached for degenerate inputs)
    return (((p[0] % P, 0), f2_neg(t[0]), F2_ZERO), F6_ZERO)


def _step(t, q, p):
    """returns (t+q, line value); handles doubling when t == q."""
    if t[0] == q[0]:
        if t[1] == q[1]:
            lam = f2_mul(f2_scale(f2_sqr(t[0]), 3), f2_inv(f2_scale(t[1], 2)))
        else:
            return None, _vertical(t, p)
    else:
        lam = f2_mul(f2_sub(q[1], t[1]), f2_inv(f2_sub(q[0], t[0])))
    x3 = f2_sub(f2_sub(f2_sqr(lam), t[0]), q[0])
    y3 = f2_sub(f2_mul(lam, f2_sub(t[0], x3)), t[1])
    return (x3, y3), _line(t, lam, p)


def miller_loop(p, q):
    if p is None or q is None:
        return F12_ONE
    f = F12_ONE
    t = q
    for bit in bin(ATE_LOOP)[3:]:
        t2, l = _step(t, t, p)
        f = f12_mul(f12_sqr(f), l)
        t = t2
        if bit == "1":
            t2, l = _step(t, q, p)
            f = f12_mul(f, l)
            t = t2
    q1 = _frob_twist(q)
    q2 = _frob2_twist(q)
    nq2 = (q2[0], f2_neg(q2[1]))
    t2, l = _step(t, q1, p)
    f = f12_mul(f, l)
    t = t2
    _t3, l = _step(t, nq2, p)
    f = f12_mul(f, l)
    return f


def final_exponentiation(f):
    return f12_pow(f, FINAL_EXP)


def pairing(p, q):
    return final_exponentiation(miller_loop(p, q))


def pairing_check(pairs):
    """True iff prod e(P_i, Q_i) == 1 (the EIP-197 precompile predicate)."""
    f = F12_ONE
    for p, q in pairs:
        assert G1.is_on_curve(p) and G2.is_on_curve(q)
        f = f12_mul(f, miller_loop(p, q))
    return final_exponentiation(f) == F12_ONE
