"""og_verify (the product's CPU Groth16 verifier; needs no GPU) against the oracle: accepts what the oracle
pairing check accepts, rejects wrong inputs, tampered proofs, non-canonical encodings, points off the curve and
G2 points outside the r-torsion.  Runs in the CPU suite through the real libowshen_gpu.so."""
import random
import struct

import numpy as np
import pytest

from oracle.py import fields, groth16 as og16
from oracle.py.curve import G1, G2, G1_GEN, G2_GEN, g1_to_bytes, g2_to_bytes
from tests.r1cs_util import random_r1cs

P, R = fields.P, fields.R


def _vk_blob(vk_o):
    ic = b"".join(g1_to_bytes(p) for p in vk_o["ic"])
    return (b"OWVK0001" + struct.pack("<Q", len(vk_o["ic"]) - 1) + g1_to_bytes(vk_o["alpha_g1"]) + g2_to_bytes(vk_o["beta_g2"]) +
            g2_to_bytes(vk_o["gamma_g2"]) + g2_to_bytes(vk_o["delta_g2"]) + ic)


@pytest.fixture(scope="module")
def instance():
    n_pub = 3
    n_wires, cons, z = random_r1cs(17, n_pub, seed=21)
    ro = og16.R1CS(n_wires, n_pub, cons)
    rnd = random.Random(8)
    pk, vk = og16.setup(ro, *(rnd.randrange(1, R) for _ in range(5)))
    proofs = [og16.prove(pk, ro, z, rnd.randrange(R), rnd.randrange(R)), og16.prove(pk, ro, z, 0, 0)]
    return vk, _vk_blob(vk), z[1:n_pub + 1], proofs


def test_accepts_valid_proofs_like_the_oracle(instance):
    from owshen_amd import groth16 as g16
    vk, blob, pub, proofs = instance
    for pr in proofs:
        assert og16.verify(vk, pub, pr)
        assert g16.verify(blob, pub, og16.proof_to_bytes(pr))


def test_rejects_wrong_inputs_and_tampered_proofs(instance):
    from owshen_amd import groth16 as g16
    vk, blob, pub, proofs = instance
    pb = og16.proof_to_bytes(proofs[0])
    for i in range(len(pub)):
        bad = list(pub)
        bad[i] = (bad[i] + 1) % R
        assert not g16.verify(blob, bad, pb)
    A, B, C = proofs[0]
    assert not g16.verify(blob, pub, og16.proof_to_bytes((A, B, G1.add(C, G1_GEN))))   # valid points, wrong proof
    assert not g16.verify(blob, pub, og16.proof_to_bytes((G1.neg(A), B, C)))
    assert not g16.verify(blob, pub, og16.proof_to_bytes((A, G2.neg(B), C)))
    for pos in (0, 40, 70, 130, 200, 255):                                              # single bit flips
        t = bytearray(pb)
        t[pos] ^= 1
        assert not g16.verify(blob, pub, bytes(t))
    assert not g16.verify(blob, pub, b"\0" * 256)                                       # points at infinity


def test_rejects_non_canonical_and_out_of_group_encodings(instance):
    from owshen_amd import groth16 as g16
    vk, blob, pub, proofs = instance
    A, B, C = proofs[0]
    pb = bytearray(og16.proof_to_bytes(proofs[0]))
    # A.x + p is the same residue but not a canonical encoding: must be rejected, not reduced
    if A[0] + P < 1 << 256:
        t = bytearray(pb)
        t[0:32] = (A[0] + P).to_bytes(32, "little")
        assert not g16.verify(blob, pub, bytes(t))
    # a public input >= r
    assert not g16.verify(blob, [pub[0] + R] + list(pub[1:]), bytes(pb)) if pub[0] + R < 1 << 256 else True
    # a point on the twist curve but outside the r-torsion (cofactor > 1): found by solving y^2 = x^3 + b' in Fq2
    from oracle.py.fields import f2_add, f2_mul, f2_sqr, f2_inv, XI
    bt = f2_mul((3, 0), f2_inv(XI))

    def fq_sqrt(a):
        s = pow(a, (P + 1) // 4, P)
        return s if s * s % P == a % P else None

    def f2_sqrt(a):
        a0, a1 = a
        n = fq_sqrt((a0 * a0 + a1 * a1) % P)
        if n is None:
            return None
        for sn in (n, (-n) % P):
            x0 = fq_sqrt((a0 + sn) * pow(2, -1, P) % P)
            if x0:
                x1 = a1 * pow(2 * x0, -1, P) % P
                if f2_sqr((x0, x1)) == (a0 % P, a1 % P):
                    return (x0, x1)
        return None
    q = None
    for xr in range(1, 200):
        x = (xr, 1)
        y = f2_sqrt(f2_add(f2_mul(f2_sqr(x), x), bt))
        if y is not None and G2.is_on_curve((x, y)) and G2.add(G2.mul((x, y), R - 1), (x, y)) is not None:
            q = (x, y)
            break
    assert q is not None, "no out-of-subgroup twist point found"
    assert not g16.verify(blob, pub, og16.proof_to_bytes((A, q, C)))
    # a G1 point off the curve
    assert not g16.verify(blob, pub, og16.proof_to_bytes(((A[0], (A[1] + 1) % P), B, C)))


def test_malformed_verifying_key_is_an_error(instance):
    from owshen_amd import groth16 as g16
    from owshen_amd.api import OwshenGpuError
    vk, blob, pub, proofs = instance
    pb = og16.proof_to_bytes(proofs[0])
    for bad in (blob[:-1], b"XXXXXXXX" + blob[8:], blob[:16] + b"\x01" + blob[17:]):
        with pytest.raises(OwshenGpuError):
            g16.verify(bad, pub, pb)
    with pytest.raises(OwshenGpuError):
        g16.verify(blob, pub[:-1], pb)


def test_vk_to_bytes_roundtrip(instance):
    from owshen_amd import groth16 as g16
    vk, blob, pub, proofs = instance
    d = {"alpha_g1": g1_to_bytes(vk["alpha_g1"]), "beta_g2": g2_to_bytes(vk["beta_g2"]), "gamma_g2": g2_to_bytes(vk["gamma_g2"]),
         "delta_g2": g2_to_bytes(vk["delta_g2"]),
         "ic": np.frombuffer(b"".join(g1_to_bytes(p) for p in vk["ic"]), dtype=np.uint8).reshape(-1, 64)}
    assert g16.vk_to_bytes(d) == blob


# ---- named algebraic known answers (EIP-197) against the product's verifier --------------------------------------

def _f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = fields.f2_mul(r, a)
        a = fields.f2_sqr(a)
        e >>= 1
    return r


def _f2_sqrt(a):
    """square root in Fq2 (q = 3 mod 4), None if a is not a square"""
    q = fields.P
    a1 = _f2_pow(a, (q - 3) // 4)
    alpha = fields.f2_mul(a1, fields.f2_mul(a1, a))
    a0 = fields.f2_mul(fields.f2_conj(alpha), alpha)
    if a0 == (q - 1, 0):
        return None
    x0 = fields.f2_mul(a1, a)
    if alpha == (q - 1, 0):
        return fields.f2_mul((0, 1), x0)
    b = _f2_pow(fields.f2_add((1, 0), alpha), (q - 1) // 2)
    return fields.f2_mul(b, x0)


def test_eip197_bilinearity_and_subgroup_check_against_og_verify():
    """e(a G1, b G2) = e(G1, G2)^(ab), exercised through the product's verifier: with alpha = al G1, beta = be G2,
    gamma = ga G2, delta = de G2, IC_0 = i0 G1 and a proof (A, B, C) = (a G1, b G2, c G1), the EIP-197 predicate holds iff
    a b = al be + i0 ga + c de (mod r).  Every point comes from the Python curve oracle.  Then: a B on the twist but
    outside the r-torsion must be rejected."""
    from owshen_amd import groth16 as g16
    rnd = random.Random(197)
    al, be, ga, de, i0, a, b = (rnd.randrange(1, fields.R) for _ in range(7))
    c = (a * b - al * be - i0 * ga) * pow(de, -1, fields.R) % fields.R
    vk = {"alpha_g1": g1_to_bytes(G1.mul(G1_GEN, al)), "beta_g2": g2_to_bytes(G2.mul(G2_GEN, be)),
          "gamma_g2": g2_to_bytes(G2.mul(G2_GEN, ga)), "delta_g2": g2_to_bytes(G2.mul(G2_GEN, de)),
          "ic": np.frombuffer(g1_to_bytes(G1.mul(G1_GEN, i0)), dtype=np.uint8).reshape(1, 64)}
    vkb = g16.vk_to_bytes(vk)
    proof = g1_to_bytes(G1.mul(G1_GEN, a)) + g2_to_bytes(G2.mul(G2_GEN, b)) + g1_to_bytes(G1.mul(G1_GEN, c))
    assert g16.verify(vkb, [], proof)
    bad = g1_to_bytes(G1.mul(G1_GEN, (a + 1) % fields.R)) + proof[64:]
    assert not g16.verify(vkb, [], bad)
    # swap the roles: (b G1, a G2) pairs to the same value
    proof2 = g1_to_bytes(G1.mul(G1_GEN, b)) + g2_to_bytes(G2.mul(G2_GEN, a)) + proof[192:]
    assert g16.verify(vkb, [], proof2)
    # a twist point outside the r-torsion (the twist's cofactor is huge, so a random twist point almost surely is)
    bt = fields.f2_scale(fields.f2_inv((9, 1)), 3)
    xq = None
    for t in range(2, 50):
        x = (t, 1)
        y = _f2_sqrt(fields.f2_add(fields.f2_mul(fields.f2_sqr(x), x), bt))
        if y is not None:
            xq = (x, y)
            break
    assert xq is not None and G2.is_on_curve(xq) and G2.add(G2.mul(xq, fields.R - 1), xq) is not None
    off = proof[:64] + g2_to_bytes(xq) + proof[192:]
    assert not g16.verify(vkb, [], off)
