"""Groth16 over BN254 on a generic R1CS: setup / prove / verify.  TEST INFRASTRUCTURE ONLY.

No reference counterpart (the snapshot has no prover; the
withdraw seam ``/root/reference/src/services/api_services/withdraw.rs:27-71`` is
ECDSA based).  Follows Groth 2016 with arkworks query naming:

    A = alpha + sum z_i A_i(tau) + r delta                       (G1)
    B = beta  + sum z_i B_i(tau) + s delta                       (G2, and a G1 copy)
    C = sum_{i>l} z_i L_i + sum_j h_j H_j + s A + r B1 - r s delta   (G1)
    verify: e(A,B) = e(alpha,beta) e(sum_{i<=l} z_i IC_i, gamma) e(C, delta)

R1CS: ``constraints`` is a list of (a_row, b_row, c_row) dicts {wire: coeff};
wire 0 is the constant 1, wires 1..n_pub are public, the rest private.
The evaluation domain has d = pow2 >= n_constraints + n_pub + 1 points; the extra
rows are arkworks' input-consistency rows (A-row = wire i, B = C = 0).
Randomness (r, s) is ALWAYS an explicit input so proofs are reproducible.
"""
from .fields import R, inv, fr_root_of_unity
from .curve import G1, G2, G1_GEN, G2_GEN, g1_to_bytes, g2_to_bytes, g1_from_bytes, g2_from_bytes
from . import ntt as _ntt
from .pairing import pairing_check


class R1CS:
    def __init__(self, n_wires, n_pub, constraints):
        self.n_wires = n_wires
        self.n_pub = n_pub
        self.constraints = constraints

    @property
    def domain_size(self):
        need = len(self.constraints) + self.n_pub + 1
        d = 1
        while d < need:
            d *= 2
        return d

    def rows(self):
        """all QAP rows: the constraints followed by the input-consistency rows."""
        rows = list(self.constraints)
        for i in range(self.n_pub + 1):
            rows.append(({i: 1}, {}, {}))
        return rows

    def is_satisfied(self, z):
        assert len(z) == self.n_wires and z[0] == 1
        for k, (a, b, c) in enumerate(self.constraints):
            av = sum(v * z[i] for i, v in a.items()) % R
            bv = sum(v * z[i] for i, v in b.items()) % R
            cv = sum(v * z[i] for i, v in c.items()) % R
            if av * bv % R != cv:
                return False
        return True

    def abc_evals(self, z):
        d = self.domain_size
        ae, be, ce = [0] * d, [0] * d, [0] * d
        for k, (a, b, c) in enumerate(self.rows()):
            ae[k] = sum(v * z[i] for i, v in a.items()) % R
            be[k] = sum(v * z[i] for i, v in b.items()) % R
            ce[k] = sum(v * z[i] for i, v in c.items()) % R
        return ae, be, ce


def lagrange_at(tau, d):
    """[L_k(tau)] for the size-d radix-2 domain, plus Z(tau) = tau^d - 1."""
    w = fr_root_of_unity(d.bit_length() - 1)
    z = (pow(tau, d, R) - 1) % R
    out = []
    wk = 1
    dinv = inv(d, R)
    for _ in range(d):
        out.append(z * dinv % R * wk % R * inv((tau - wk) % R, R) % R)
        wk = wk * w % R
    return out, z


def qap_at(r1cs, tau):
    """(a_i, b_i, c_i) = (A_i(tau), B_i(tau), C_i(tau)) per wire, and Z(tau)."""
    d = r1cs.domain_size
    lag, z = lagrange_at(tau, d)
    m = r1cs.n_wires
    a, b, c = [0] * m, [0] * m, [0] * m
    for k, (ra, rb, rc) in enumerate(r1cs.rows()):
        lk = lag[k]
        for i, v in ra.items():
            a[i] = (a[i] + v * lk) % R
        for i, v in rb.items():
            b[i] = (b[i] + v * lk) % R
        for i, v in rc.items():
            c[i] = (c[i] + v * lk) % R
    return a, b, c, z


def setup_scalars(r1cs, tau, alpha, beta, gamma, delta):
    """The discrete logs of every key element (so the GPU fixed-base setup can be checked)."""
    a, b, c, z = qap_at(r1cs, tau)
    d = r1cs.domain_size
    l = r1cs.n_pub
    ginv, dinv = inv(gamma, R), inv(delta, R)
    kk = [(beta * a[i] + alpha * b[i] + c[i]) % R for i in range(r1cs.n_wires)]
    ic = [kk[i] * ginv % R for i in range(l + 1)]
    lq = [kk[i] * dinv % R for i in range(l + 1, r1cs.n_wires)]
    hq = []
    t = z * dinv % R
    for _ in range(d - 1):
        hq.append(t)
        t = t * tau % R
    return {"a": a, "b": b, "l": lq, "h": hq, "ic": ic}


def setup(r1cs, tau, alpha, beta, gamma, delta):
    sc = setup_scalars(r1cs, tau, alpha, beta, gamma, delta)
    t1 = G1.fixed_base_table(G1_GEN)
    t2 = G2.fixed_base_table(G2_GEN)
    m1 = lambda k: G1.fixed_base_mul(t1, k)
    m2 = lambda k: G2.fixed_base_mul(t2, k)
    pk = {
        "alpha_g1": m1(alpha), "beta_g1": m1(beta), "beta_g2": m2(beta),
        "delta_g1": m1(delta), "delta_g2": m2(delta),
        "a_query": [m1(k) for k in sc["a"]],
        "b_g1_query": [m1(k) for k in sc["b"]],
        "b_g2_query": [m2(k) for k in sc["b"]],
        "l_query": [m1(k) for k in sc["l"]],
        "h_query": [m1(k) for k in sc["h"]],
    }
    vk = {
        "alpha_g1": pk["alpha_g1"], "beta_g2": pk["beta_g2"],
        "gamma_g2": m2(gamma), "delta_g2": pk["delta_g2"],
        "ic": [m1(k) for k in sc["ic"]],
    }
    return pk, vk


def prove(pk, r1cs, z, r, s, msm1=None, msm2=None):
    """returns (A, B, C) affine.  msm1/msm2 default to the oracle's Pippenger."""
    msm1 = msm1 or G1.msm
    msm2 = msm2 or G2.msm
    assert len(z) == r1cs.n_wires
    l = r1cs.n_pub
    ae, be, ce = r1cs.abc_evals(z)
    h = _ntt.h_poly(ae, be, ce)
    d = r1cs.domain_size
    assert h[d - 1] == 0
    A = G1.add(G1.add(pk["alpha_g1"], msm1(z, pk["a_query"])), G1.mul(pk["delta_g1"], r))
    B2 = G2.add(G2.add(pk["beta_g2"], msm2(z, pk["b_g2_query"])), G2.mul(pk["delta_g2"], s))
    B1 = G1.add(G1.add(pk["beta_g1"], msm1(z, pk["b_g1_query"])), G1.mul(pk["delta_g1"], s))
    C = G1.add(msm1(z[l + 1:], pk["l_query"]), msm1(h[:d - 1], pk["h_query"]))
    C = G1.add(C, G1.mul(A, s))
    C = G1.add(C, G1.mul(B1, r))
    C = G1.add(C, G1.neg(G1.mul(pk["delta_g1"], r * s % R)))
    return A, B2, C


def verify(vk, public_inputs, proof):
    A, B, C = proof
    if len(public_inputs) + 1 != len(vk["ic"]):
        return False
    for pt, grp in ((A, G1), (B, G2), (C, G1)):
        if pt is None or not grp.is_on_curve(pt):
            return False
    acc = vk["ic"][0]
    for x, pt in zip(public_inputs, vk["ic"][1:]):
        acc = G1.add(acc, G1.mul(pt, x))
    return pairing_check([
        (G1.neg(A), B),
        (vk["alpha_g1"], vk["beta_g2"]),
        (acc, vk["gamma_g2"]),
        (C, vk["delta_g2"]),
    ])


# ---- wire formats -----------------------------------------------------------

def proof_to_bytes(proof):
    """256 B: A (x||y) || B (x.c0||x.c1||y.c0||y.c1) || C (x||y), 32-byte LE limbs."""
    A, B, C = proof
    return g1_to_bytes(A) + g2_to_bytes(B) + g1_to_bytes(C)


def proof_from_bytes(b):
    assert len(b) == 256
    return g1_from_bytes(b[0:64]), g2_from_bytes(b[64:192]), g1_from_bytes(b[192:256])


def proof_to_evm_calldata(proof):
    """8 x uint256 big-endian, G2 as (x.c1, x.c0, y.c1, y.c0) -- snarkjs/EIP-197 order."""
    A, B, C = proof
    vals = [A[0], A[1], B[0][1], B[0][0], B[1][1], B[1][0], C[0], C[1]]
    return b"".join(int(v).to_bytes(32, "big") for v in vals)
