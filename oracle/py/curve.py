"""BN254 G1 / G2 group law on Python ints.  TEST INFRASTRUCTURE ONLY.

No reference counterpart (no G1/G2 in the snapshot);
follows EIP-196/197: G1: y^2 = x^3 + 3 over Fq, generator (1, 2);
G2: y^2 = x^3 + 3/(9+u) over Fq2.  Points are affine tuples (x, y) or None
for the point at infinity.  Jacobian (X, Y, Z) used internally.
"""
from .fields import P, R, inv, f2_add, f2_sub, f2_mul, f2_sqr, f2_neg, f2_inv, f2_scale, F2_ZERO, F2_ONE

G1_GEN = (1, 2)
G1_B = 3

G2_GEN = (
    (10857046999023057135944570762232829481370756359578518086990519993285655852781,
     11559732032986387107991004021392285783925812861821192530917403151452391805634),
    (8495653923123431417604973247489272438418190587263600148770280649306958101930,
     4082367875863433681332203403145435568316851327593401208105741076214120093531),
)
# b' = 3 / (9 + u)
G2_B = f2_scale(f2_inv((9, 1)), 3)


class _Fq:
    zero = 0
    one = 1
    add = staticmethod(lambda a, b: (a + b) % P)
    sub = staticmethod(lambda a, b: (a - b) % P)
    mul = staticmethod(lambda a, b: a * b % P)
    sqr = staticmethod(lambda a: a * a % P)
    neg = staticmethod(lambda a: (-a) % P)
    inv = staticmethod(lambda a: inv(a, P))
    is_zero = staticmethod(lambda a: a % P == 0)


class _Fq2:
    zero = F2_ZERO
    one = F2_ONE
    add = staticmethod(f2_add)
    sub = staticmethod(f2_sub)
    mul = staticmethod(f2_mul)
    sqr = staticmethod(f2_sqr)
    neg = staticmethod(f2_neg)
    inv = staticmethod(f2_inv)
    is_zero = staticmethod(lambda a: a[0] % P == 0 and a[1] % P == 0)


class Group:
    """Short-Weierstrass a=0 group over field F with Jacobian internals."""

    def __init__(self, F, b, gen):
        self.F = F
        self.b = b
        self.gen = gen

    def is_on_curve(self, pt):
        if pt is None:
            return True
        F = self.F
        x, y = pt
        return F.sub(F.sqr(y), F.add(F.mul(F.sqr(x), x), self.b)) == F.zero

    # --- jacobian ---
    def to_jac(self, pt):
        if pt is None:
            return (self.F.one, self.F.one, self.F.zero)
        return (pt[0], pt[1], self.F.one)

    def to_affine(self, j):
        F = self.F
        X, Y, Z = j
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def jdbl(self, j):
        F = self.F
        X, Y, Z = j
        if F.is_zero(Z):
            return j
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        t = F.sub(F.sqr(F.add(X, B)), F.add(A, C))
        D = F.add(t, t)
        E = F.add(F.add(A, A), A)
        Fv = F.sqr(E)
        X3 = F.sub(Fv, F.add(D, D))
        C8 = F.add(C, C)
        C8 = F.add(C8, C8)
        C8 = F.add(C8, C8)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
        Z3 = F.mul(F.add(Y, Y), Z)
        return (X3, Y3, Z3)

    def jadd(self, p, q):
        F = self.F
        X1, Y1, Z1 = p
        X2, Y2, Z2 = q
        if F.is_zero(Z1):
            return q
        if F.is_zero(Z2):
            return p
        Z1Z1 = F.sqr(Z1)
        Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(Y1, F.mul(Z2, Z2Z2))
        S2 = F.mul(Y2, F.mul(Z1, Z1Z1))
        if U1 == U2:
            if S1 == S2:
                return self.jdbl(p)
            return (F.one, F.one, F.zero)
        H = F.sub(U2, U1)
        Rr = F.sub(S2, S1)
        HH = F.sqr(H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.sqr(Rr), HHH), F.add(V, V))
        Y3 = F.sub(F.mul(Rr, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    # --- affine API ---
    def neg(self, pt):
        if pt is None:
            return None
        return (pt[0], self.F.neg(pt[1]))

    def add(self, p, q):
        return self.to_affine(self.jadd(self.to_jac(p), self.to_jac(q)))

    def double(self, p):
        return self.to_affine(self.jdbl(self.to_jac(p)))

    def jmul(self, j, k):
        F = self.F
        k %= R
        acc = (F.one, F.one, F.zero)
        for bit in bin(k)[2:] if k else "":
            acc = self.jdbl(acc)
            if bit == "1":
                acc = self.jadd(acc, j)
        return acc

    def mul(self, pt, k):
        return self.to_affine(self.jmul(self.to_jac(pt), k))

    def msm_naive(self, scalars, points):
        """sum_i s_i * P_i by independent double-and-add (the definition)."""
        acc = self.to_jac(None)
        for s, p in zip(scalars, points):
            if p is None or s % R == 0:
                continue
            acc = self.jadd(acc, self.jmul(self.to_jac(p), s))
        return self.to_affine(acc)

    def msm(self, scalars, points, c=None):
        """Unsigned-window Pippenger; same value as msm_naive, faster for tests."""
        n = len(scalars)
        if n == 0:
            return None
        if c is None:
            c = max(2, min(16, n.bit_length() - 2))
        nwin = (254 + c - 1) // c
        inf = self.to_jac(None)
        total = inf
        jp = [self.to_jac(p) for p in points]
        ss = [s % R for s in scalars]
        for w in reversed(range(nwin)):
            for _ in range(c):
                total = self.jdbl(total)
            buckets = {}
            sh = w * c
            mask = (1 << c) - 1
            for s, p in zip(ss, jp):
                d = (s >> sh) & mask
                if d:
                    b = buckets.get(d)
                    buckets[d] = p if b is None else self.jadd(b, p)
            run = inf
            acc = inf
            prev = None
            for d in sorted(buckets, reverse=True):
                if prev is not None:
                    # acc += run * (prev - d)
                    acc = self.jadd(acc, self.jmul(run, prev - d)) if prev - d > 1 else self.jadd(acc, run)
                run = self.jadd(run, buckets[d])
                prev = d
            if prev is not None:
                acc = self.jadd(acc, self.jmul(run, prev)) if prev > 1 else self.jadd(acc, run)
            total = self.jadd(total, acc)
        return self.to_affine(total)

    def fixed_base_table(self, base=None, c=8):
        """table[w][d] = d * 2^(c*w) * base (affine), for fast setup."""
        base = self.gen if base is None else base
        nwin = (254 + c - 1) // c
        tab = []
        cur = self.to_jac(base)
        for _ in range(nwin):
            row = [None]
            acc = self.to_jac(None)
            for _d in range(1, 1 << c):
                acc = self.jadd(acc, cur)
                row.append(acc)
            tab.append(row)
            for _k in range(c):
                cur = self.jdbl(cur)
        return (c, tab)

    def fixed_base_mul(self, table, k):
        c, tab = table
        k %= R
        acc = self.to_jac(None)
        w = 0
        mask = (1 << c) - 1
        while k:
            d = k & mask
            if d:
                acc = self.jadd(acc, tab[w][d])
            k >>= c
            w += 1
        return self.to_affine(acc)


G1 = Group(_Fq, G1_B, G1_GEN)
G2 = Group(_Fq2, G2_B, G2_GEN)

assert G1.is_on_curve(G1_GEN)
assert G2.is_on_curve(G2_GEN)


# ---- byte encodings used at the C-ABI seam (32-byte LE limbs) ----

def g1_to_bytes(pt):
    """affine x||y, each 32-byte LE; infinity = 64 zero bytes."""
    if pt is None:
        return bytes(64)
    return int(pt[0]).to_bytes(32, "little") + int(pt[1]).to_bytes(32, "little")


def g1_from_bytes(b):
    x = int.from_bytes(b[0:32], "little")
    y = int.from_bytes(b[32:64], "little")
    if x == 0 and y == 0:
        return None
    return (x, y)


def g2_to_bytes(pt):
    """affine x.c0||x.c1||y.c0||y.c1, each 32-byte LE; infinity = zeros."""
    if pt is None:
        return bytes(128)
    (x0, x1), (y0, y1) = pt
    return b"".join(int(v).to_bytes(32, "little") for v in (x0, x1, y0, y1))


def g2_from_bytes(b):
    v = [int.from_bytes(b[i * 32:(i + 1) * 32], "little") for i in range(4)]
    if not any(v):
        return None
    return ((v[0], v[1]), (v[2], v[3]))
