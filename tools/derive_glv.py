#!/usr/bin/env python3
"""Derives the GLV constants of BN254 G1 used by the proof assembly (owshen_amd/csrc/glv.h) from the two moduli alone, and
prints them: beta (a cube root of unity in Fq), lambda (the matching cube root of unity in Fr: (beta x, y) = lambda (x, y)), a
reduced basis (a1, b1), (a2, b2) of the lattice {(a, b): a + b lambda = 0 mod r}, and the rounding multipliers
g1 = floor(2^320 b2 / r), g2 = floor(2^320 (-b1) / r).  Nothing here is typed from memory: everything is recomputed and checked
on the generator.  tests/test_glv.py re-derives lambda the same way and checks the library's decomposition against it."""
# Self-contained on purpose: only tests/, smoke() and bench.py's CPU leg may touch oracle/, and this is a developer tool.
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617   # BN254 group order (the reference's Fp)
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583   # BN254 base field
G1_GEN = (1, 2)                                                                      # y^2 = x^3 + 3


def _add(p1, p2):
    if p1 is None:
        return p2
    if p2 is None:
        return p1
    (x1, y1), (x2, y2) = p1, p2
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        m = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        m = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (m * m - x1 - x2) % Q
    return x3, (m * (x1 - x3) - y1) % Q


def _mul(pt, k):
    acc = None
    while k:
        if k & 1:
            acc = _add(acc, pt)
        pt = _add(pt, pt)
        k >>= 1
    return acc


def cube_roots_of_unity(p):
    for g in range(2, 50):
        w = pow(g, (p - 1) // 3, p)
        if w != 1:
            return w, w * w % p
    raise SystemExit("no cube root of unity found")


def derive():
    betas, lambdas = cube_roots_of_unity(Q), cube_roots_of_unity(R)
    x, y = G1_GEN
    pair = None
    for b in betas:
        for lam in lambdas:
            if _mul(G1_GEN, lam) == (b * x % Q, y):
                pair = (b, lam)
    assert pair, "no (beta, lambda) pair acts as the endomorphism"
    beta, lam = pair
    # extended Euclid on (r, lambda): remainders r_i = s_i r + t_i lambda; stop around sqrt(r)
    rs, ts = [R, lam], [0, 1]
    while rs[-1] != 0:
        qq = rs[-2] // rs[-1]
        rs.append(rs[-2] - qq * rs[-1])
        ts.append(ts[-2] - qq * ts[-1])
    import math
    sq = math.isqrt(R)
    l = max(i for i in range(len(rs)) if rs[i] >= sq)
    a1, b1 = rs[l + 1], -ts[l + 1]
    cand = [(rs[l], -ts[l]), (rs[l + 2], -ts[l + 2])]
    a2, b2 = min(cand, key=lambda v: v[0] * v[0] + v[1] * v[1])
    for a, b in ((a1, b1), (a2, b2)):
        assert (a + b * lam) % R == 0
    det = a1 * b2 - a2 * b1
    assert abs(det) == R, det
    if det < 0:
        a2, b2 = -a2, -b2
    g1 = (b2 << 320) // R
    g2 = ((-b1) << 320) // R
    return dict(beta=beta, lam=lam, a1=a1, b1=b1, a2=a2, b2=b2, g1=g1, g2=g2)


def decompose(k, c):
    """the library's arithmetic, in Python integers (floor instead of round: still short)"""
    c1 = (k * c["g1"]) >> 320
    c2 = (k * c["g2"]) >> 320
    k1 = k - c1 * c["a1"] - c2 * c["a2"]
    k2 = -c1 * c["b1"] - c2 * c["b2"]
    return k1, k2


if __name__ == "__main__":
    c = derive()
    import random
    rnd = random.Random(1)
    worst = 0
    for _ in range(20000):
        k = rnd.randrange(R)
        k1, k2 = decompose(k, c)
        assert (k1 + c["lam"] * k2 - k) % R == 0
        worst = max(worst, abs(k1).bit_length(), abs(k2).bit_length())
    for k in (0, 1, R - 1, R // 2, c["lam"]):
        k1, k2 = decompose(k, c)
        assert (k1 + c["lam"] * k2 - k) % R == 0
        worst = max(worst, abs(k1).bit_length(), abs(k2).bit_length())
    print("worst |k_i| bits over the samples:", worst)
    for name, v in c.items():
        print(f"{name} = {v} ({v:#x})")
