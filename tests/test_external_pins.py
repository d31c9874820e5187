"""External pins: every constant SURVEY.md section 8 (rows N1 / N3, 8c) lists as verified is asserted BY NAME
against the Python oracle, the C oracle's tables, the product's 9 x 29-bit tables -- and, on the GPU, through og_field_op_d /
og_msm_d.  The values below are typed from SURVEY.md (which took them from the public record: EIP-196/197, the reference's
`Fp` derive at /root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11), NOT computed by this repository, so a
wrong modulus, generator, twist or Montgomery constant anywhere in the three implementations fails here.

Also: the bilinearity identities e(a G1, b G2) = e(G1, G2)^(a b) and e(G1, G2)^r = 1 checked against the PRODUCT's verifier
(og_verify, in both libraries) with hand-built verifying keys.  Round 5: a PUBLISHED pairing-check vector (EIP-197 / go-ethereum's
bn256 pairing precompile tests, "jeff1") through the oracle's precompile model, the second engine and -- embedded in the Groth16
predicate -- the product's verifier in both libraries."""
import os
import random
import re

import numpy as np
import pytest

from oracle.py import fields
from oracle.py.curve import G1, G2, G1_GEN, G2_GEN, g1_to_bytes, g2_to_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# ---- SURVEY.md 8a-N1 (R = 2^256) -----------------------------------------------------------------------------------
FR_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617   # R1: the reference's Fp
FQ_MODULUS = 21888242871839275222246405745257275088696311157297823662689037894645226208583   # N1: EIP-196 base field
FR_GENERATOR = 7
FR_ROOT_2_28 = 1748695177688661943023146337482803886740723238769601073607632802312037301404    # R1: 2^28-th root from g = 7
FR_R_MOD = 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb
FR_R2_MOD = 0x0216d0b17f4e44a58c49833d53bb808553fe3ab1e35c59e31bb8e645ae216da7
FR_NEG_INV_64 = 0xc2e1f593efffffff
FQ_R_MOD = 0x0e0a77c19a07df2f666ea36f7879462c0a78eb28f5c70b3dd35d438dc58f0d9d
FQ_R2_MOD = 0x06d89f71cab8351f47ab1eff0a417ff6b5e71911d44501fbf32cfc5b538afa89
FQ_NEG_INV_64 = 0x87d20782e4866389
# ---- SURVEY.md 8a-N3 -------------------------------------------------------------------------------------------------
G2_GEN_X = (10857046999023057135944570762232829481370756359578518086990519993285655852781,
            11559732032986387107991004021392285783925812861821192530917403151452391805634)
G2_GEN_Y = (8495653923123431417604973247489272438418190587263600148770280649306958101930,
            4082367875863433681332203403145435568316851327593401208105741076214120093531)
# ---- SURVEY.md 8c ----------------------------------------------------------------------------------------------------
BN_X = 4965661367192848881
# EIP-196 test vector quoted in DESIGN.md section 0: 2 * (1, 2)
G1_DOUBLE = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
             9918110051302171585080402603319702774565515993150576347155970296011118125764)


def test_named_constants_are_mutually_consistent():
    """the typed values satisfy the identities SURVEY.md states for them (a typo above cannot hide)"""
    R256 = 1 << 256
    assert FR_R_MOD == R256 % FR_MODULUS and FR_R2_MOD == R256 * R256 % FR_MODULUS
    assert FQ_R_MOD == R256 % FQ_MODULUS and FQ_R2_MOD == R256 * R256 % FQ_MODULUS
    assert (FR_MODULUS * FR_NEG_INV_64 + 1) % (1 << 64) == 0 and (FQ_MODULUS * FQ_NEG_INV_64 + 1) % (1 << 64) == 0
    assert FR_NEG_INV_64 & 0xffffffff == 0xefffffff and FQ_NEG_INV_64 & 0xffffffff == 0xe4866389
    assert FQ_MODULUS == 36 * BN_X**4 + 36 * BN_X**3 + 24 * BN_X**2 + 6 * BN_X + 1
    assert FR_MODULUS == 36 * BN_X**4 + 36 * BN_X**3 + 18 * BN_X**2 + 6 * BN_X + 1
    assert pow(FR_ROOT_2_28, 1 << 28, FR_MODULUS) == 1 and pow(FR_ROOT_2_28, 1 << 27, FR_MODULUS) == FR_MODULUS - 1
    assert FR_ROOT_2_28 == pow(FR_GENERATOR, (FR_MODULUS - 1) >> 28, FR_MODULUS)


def test_python_oracle_uses_the_named_constants():
    assert (fields.R, fields.P, fields.BN_X, fields.FR_GENERATOR) == (FR_MODULUS, FQ_MODULUS, BN_X, FR_GENERATOR)
    assert G1_GEN == (1, 2) and G2_GEN == (G2_GEN_X, G2_GEN_Y)
    assert G1.mul(G1_GEN, 2) == G1_DOUBLE
    # the twist: y^2 = x^3 + 3 / (9 + u) holds for the named generator
    x, y, F = G2_GEN_X, G2_GEN_Y, G2.F
    bprime = F.mul((3, 0), F.inv((9, 1)))
    assert F.sub(F.mul(y, y), F.add(F.mul(F.mul(x, x), x), bprime)) == (0, 0)
    assert G2.mul(G2_GEN, FR_MODULUS) is None and G1.mul(G1_GEN, FR_MODULUS) is None
    assert fields.fr_root_of_unity(28) == FR_ROOT_2_28


def _c_define(text, section, name):
    """value of `#define name` inside the Fq (0) / Fr (1) parameter block of oracle/c/oracle.c"""
    vals = re.findall(r"#define %s\s+(.+)" % name, text)
    return vals[section]


def _limbs64(s):
    w = [int(x.rstrip("ULL"), 16) for x in re.findall(r"0x[0-9a-fA-F]+ULL", s)]
    return sum(v << (64 * i) for i, v in enumerate(w))


def test_c_oracle_tables_are_the_named_constants():
    text = open(os.path.join(ROOT, "oracle", "c", "oracle.c")).read()
    for section, mod, r1, r2, ninv in ((0, FQ_MODULUS, FQ_R_MOD, FQ_R2_MOD, FQ_NEG_INV_64), (1, FR_MODULUS, FR_R_MOD, FR_R2_MOD, FR_NEG_INV_64)):
        assert _limbs64(_c_define(text, section, "FIELD_N")) == mod
        assert _limbs64(_c_define(text, section, "FIELD_ONE")) == r1
        assert _limbs64(_c_define(text, section, "FIELD_R2")) == r2
        assert int(_c_define(text, section, "FIELD_INV").rstrip("ULL"), 16) == ninv


def _limbs29(s):
    w = [int(x.rstrip("u"), 16) for x in re.findall(r"0x[0-9a-fA-F]+u", s)]
    assert len(w) == 9 and all(v < (1 << 30) for v in w)
    return sum(v << (29 * i) for i, v in enumerate(w))


def test_product_tables_are_the_named_constants():
    """owshen_amd/csrc/field.hip.h holds the same fields in 9 x 29-bit limbs with R = 2^261 = 2^5 * 2^256: its tables must be
    the named R = 2^256 constants moved by that factor"""
    text = open(os.path.join(ROOT, "owshen_amd", "csrc", "field.hip.h")).read()
    for struct, mod, r1, r2, ninv in (("FqParams", FQ_MODULUS, FQ_R_MOD, FQ_R2_MOD, FQ_NEG_INV_64), ("FrParams", FR_MODULUS, FR_R_MOD, FR_R2_MOD, FR_NEG_INV_64)):
        body = text[text.index("struct %s {" % struct):]
        body = body[:body.index("\n};")]
        tab = {n: _limbs29(v) for n, v in re.findall(r"constexpr uint32_t (\w+)\[9\] = \{([^}]*)\}", body)}
        assert tab["N"] == mod and tab["N2"] == 2 * mod
        assert tab["ONE"] == r1 * 32 % mod                       # 2^261 mod N
        assert tab["R2"] == r2 * 1024 % mod                      # 2^522 mod N
        inv29 = int(re.search(r"INV = (0x[0-9a-f]+)u", body).group(1), 16)
        assert inv29 == ninv & ((1 << 29) - 1)                   # -N^-1 mod 2^29
        for k in (3, 4, 5):
            assert tab["N%d" % k] == k * mod
        assert tab["NEG4"] == 4 * mod and tab["NEG8"] == 8 * mod   # borrow-inflated limbs, same value


def test_eddsa_base_point_is_the_references():
    """the BabyJubJub BASE point words embedded in eddsa.hip are the reference's (mod.rs:177-188), on curve and of prime order"""
    from oracle.py import babyjubjub as bj
    assert bj.is_on_curve(bj.BASE) and bj.multiply(bj.BASE, bj.ORDER // 8) == (0, 1)   # prime order: the "Base8" point
    text = open(os.path.join(ROOT, "owshen_amd", "csrc", "eddsa.hip")).read()
    for name, want in (("BX", bj.BASE[0]), ("BY", bj.BASE[1])):
        words = re.search(r"%s\[8\] = \{([^}]*)\}" % name, text).group(1)
        got = sum(int(w.rstrip("u"), 16) << (32 * i) for i, w in enumerate(re.findall(r"0x[0-9a-f]+u", words)))
        assert got == want, name


def _vk_blob(alpha, beta, gamma, delta, ic):
    return (b"OWVK0001" + (len(ic) - 1).to_bytes(8, "little") + g1_to_bytes(alpha) + g2_to_bytes(beta) + g2_to_bytes(gamma) +
            g2_to_bytes(delta) + b"".join(g1_to_bytes(p) for p in ic))


def _verifiers():
    from owshen_amd import groth16 as g16, verify_only
    return (("libowshen_gpu.so og_verify", lambda vk, pub, pr: g16.verify(vk, pub, pr)),
            ("libowshen_verify.so og_verify", lambda vk, pub, pr: verify_only.verify(vk, pub, pr)))


def test_bilinearity_against_the_products_verifier():
    """e(A, B) = e(alpha, beta) e(IC_0, gamma) e(C, delta) with alpha = (a b mod r) G1, beta = gamma = delta = G2, IC_0 = k G1,
    C = -k G1, A = a G1, B = b G2: accepted iff e(a G1, b G2) = e(G1, G2)^(a b) -- including when a b wraps around r, i.e.
    e(G1, G2)^r = 1.  A key with (a b + 1) in alpha must be rejected."""
    rnd = random.Random(1997)
    for name, verify in _verifiers():
        for trial in range(3):
            a = rnd.randrange(2, FR_MODULUS)
            b = rnd.randrange(FR_MODULUS - 1000, FR_MODULUS) if trial == 0 else rnd.randrange(2, FR_MODULUS)
            if trial == 0:
                assert a * b >= FR_MODULUS                      # the exponent wraps: uses e(G1, G2)^r = 1
            k = rnd.randrange(1, FR_MODULUS)
            A, B = G1.mul(G1_GEN, a), G2.mul(G2_GEN, b)
            ic0, C = G1.mul(G1_GEN, k), G1.neg(G1.mul(G1_GEN, k))
            proof = g1_to_bytes(A) + g2_to_bytes(B) + g1_to_bytes(C)
            good = _vk_blob(G1.mul(G1_GEN, a * b % FR_MODULUS), G2_GEN, G2_GEN, G2_GEN, [ic0])
            bad = _vk_blob(G1.mul(G1_GEN, (a * b + 1) % FR_MODULUS), G2_GEN, G2_GEN, G2_GEN, [ic0])
            assert verify(good, [], proof), name
            assert not verify(bad, [], proof), name
        # linearity in the public input: vk_x = IC_0 + x IC_1 with IC_1 = G1 shifts the accepted alpha by x
        x = rnd.randrange(1, FR_MODULUS)
        a, b = rnd.randrange(2, FR_MODULUS), rnd.randrange(2, FR_MODULUS)
        proof = g1_to_bytes(G1.mul(G1_GEN, a)) + g2_to_bytes(G2.mul(G2_GEN, b)) + g1_to_bytes(G1.neg(G1.mul(G1_GEN, x)))
        vk = _vk_blob(G1.mul(G1_GEN, a * b % FR_MODULUS), G2_GEN, G2_GEN, G2_GEN, [G1.mul(G1_GEN, FR_MODULUS - 7), G1.add(G1_GEN, G1.mul(G1_GEN, 7 * pow(x, -1, FR_MODULUS) % FR_MODULUS))])
        # vk_x = -7 G1 + x (1 + 7 / x) G1 = x G1;  C = -x G1 cancels it
        assert verify(vk, [x], proof), name
        assert not verify(vk, [(x + 1) % FR_MODULUS], proof), name


# ---- a PUBLISHED pairing vector against every verifier of this repository -----------------------------------------------
# The first test case of go-ethereum's bn256 pairing precompile ("jeff1", core/vm/testdata/precompiles/bn256Pairing.json; the
# same bytes are in the EIP-197 test suites): two (G1, G2) pairs whose pairing product is 1 -- the second G2 point is the
# generator.  Typed from the public record, NOT produced by this repository; and self-validating: a mis-typed digit would leave
# a point off the curve, outside the r-torsion, or break the product (checked first, with the Python oracle).
EIP197_JEFF1 = """
1c76476f4def4bb94541d57ebba1193381ffa7aa76ada664dd31c16024c43f59 3034dd2920f673e204fee2811c678745fc819b55d3e9d294e45c9b03a76aef41
209dd15ebff5d46c4bd888e51a93cf99a7329636c63514396b4a452003a35bf7 04bf11ca01483bfa8b34b43561848d28905960114c8ac04049af4b6315a41678
2bb8324af6cfc93537a2ad1a445cfd0ca2a71acd7ac41fadbf933c2a51be344d 120a2a4cf30c1bf9845f20c6fe39e07ea2cce61f0c9bb048165fe5e4de877550
111e129f1cf1097710d41c4ac70fcdfa5ba2023c6ff1cbeac322de49d1b6df7c 2032c61a830e3c17286de9462bf242fca2883585b93870a73853face6a6bf411
198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2 1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed
090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b 12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa
"""


def _jeff1():
    w = [int(x, 16) for x in EIP197_JEFF1.split()]
    assert len(w) == 12
    # precompile order: G1.x, G1.y, G2.x_imag, G2.x_real, G2.y_imag, G2.y_real
    p1, q1 = (w[0], w[1]), ((w[3], w[2]), (w[5], w[4]))
    p2, q2 = (w[6], w[7]), ((w[9], w[8]), (w[11], w[10]))
    return w, p1, q1, p2, q2


def test_published_eip197_pairing_vector_against_the_oracle_model():
    from oracle.py import evm_model
    w, p1, q1, p2, q2 = _jeff1()
    assert q2 == G2_GEN                                               # the second G2 point of the vector IS the generator
    assert G1.is_on_curve(p1) and G1.is_on_curve(p2) and G2.is_on_curve(q1)
    assert G2.mul(q1, FR_MODULUS) is None                            # in the r-torsion
    assert evm_model.ec_pairing(w) == 1
    bad = list(w)
    bad[6], bad[7] = 1, 2                                             # another (valid) G1 point in the second pair
    assert evm_model.ec_pairing(bad) == 0


def test_published_eip197_pairing_vector_against_the_products_verifier():
    """e(P1, Q1) e(P2, Q2) = 1 embedded in the Groth16 predicate og_verify evaluates, e(-A, B) e(alpha, beta) e(IC_0, gamma)
    e(C, delta) = 1: A = -P1, B = Q1, alpha = P2, beta = Q2 and the other two pairs cancelling (IC_0 = G1, C = -G1, gamma = delta =
    G2).  The PRODUCT's verifier -- both libraries -- must accept it, and refuse it with the two G1 points exchanged or with
    one bit of A changed: its Miller loop and final exponentiation agree with a vector this repository did not make."""
    _w, p1, q1, p2, q2 = _jeff1()
    vk = _vk_blob(p2, q2, G2_GEN, G2_GEN, [G1_GEN])
    proof = g1_to_bytes(G1.neg(p1)) + g2_to_bytes(q1) + g1_to_bytes(G1.neg(G1_GEN))
    swapped = g1_to_bytes(G1.neg(p2)) + g2_to_bytes(q1) + g1_to_bytes(G1.neg(G1_GEN))     # e(P2, Q1) e(P2, Q2) != 1
    for name, verify in _verifiers():
        assert verify(vk, [], proof), name
        assert not verify(vk, [], swapped), name
        assert not verify(_vk_blob(p1, q2, G2_GEN, G2_GEN, [G1_GEN]), [], proof), name
        t = bytearray(proof)
        t[3] ^= 0x10
        assert not verify(vk, [], bytes(t)), name


def test_published_eip197_pairing_vector_against_the_second_engine():
    """the same embedding through the second, differently built pairing (oracle/js/bn254_pairing_second.js, V8 BigInt)"""
    import shutil
    import struct
    if shutil.which("node") is None:
        pytest.skip("no node binary")
    from tests import test_second_engine as se
    _w, p1, q1, p2, q2 = _jeff1()
    dec = lambda pt: [str(pt[0]), str(pt[1])]
    dec2 = lambda q: [[str(q[0][0]), str(q[0][1])], [str(q[1][0]), str(q[1][1])]]
    vk = {"alpha": dec(p2), "beta": dec2(q2), "gamma": dec2(G2_GEN), "delta": dec2(G2_GEN), "ic": [dec(G1_GEN)]}
    good = {"a": dec(G1.neg(p1)), "b": dec2(q1), "c": dec(G1.neg(G1_GEN))}
    bad = {"a": dec(G1.neg(p2)), "b": dec2(q1), "c": dec(G1.neg(G1_GEN))}
    res = se._run_pairing({"vk": vk, "proofs": [{"public": [], "proof": good}, {"public": [], "proof": bad}]})
    assert res["proofs"] == [True, False]


# ---- on the GPU: the named constants through the C ABI -------------------------------------------------------------------

@pytest.mark.gpu
def test_gpu_field_ops_reproduce_the_named_montgomery_identities(ctx):
    """og_field_op_d on the canonical integers R mod N and R^2 mod N: (R mod N)^2 = R^2 mod N, (R mod N)(R^-1) = 1, ... with both
    operands and the expected results typed from SURVEY.md, in both fields"""
    from owshen_amd import api
    for field, mod, r1, r2 in ((api.FR, FR_MODULUS, FR_R_MOD, FR_R2_MOD), (api.FQ, FQ_MODULUS, FQ_R_MOD, FQ_R2_MOD)):
        a = ctx.to_device(api.ints_to_bytes([r1, r1, r2, mod - 1, FR_ROOT_2_28 % mod]))
        b = ctx.to_device(api.ints_to_bytes([r1, 1, 1, mod - 1, FR_ROOT_2_28 % mod]))
        got = api.bytes_to_ints(ctx.to_host(ctx.field_op(field, "mul", a, b)))
        assert got[:4] == [r2, r1, r2, 1]
        inv = api.bytes_to_ints(ctx.to_host(ctx.field_op(field, "inv", a)))
        assert inv[0] * r1 % mod == 1 and inv[3] == mod - 1
        s = api.bytes_to_ints(ctx.to_host(ctx.field_op(field, "add", a, b)))
        assert s[0] == 2 * r1 % mod and s[3] == mod - 2
    # the 2^28-th root of unity, squared 27 times on the GPU, is -1
    x = ctx.to_device(api.ints_to_bytes([FR_ROOT_2_28]))
    for _ in range(27):
        x = ctx.field_op(api.FR, "mul", x, x)
    assert api.bytes_to_ints(ctx.to_host(x)) == [FR_MODULUS - 1]


@pytest.mark.gpu
def test_gpu_msm_reproduces_the_named_group_constants(ctx):
    """og_msm_d over the named generators: 2 G1 is the EIP-196 vector, r G1 = r G2 = O, (r - 1) G2 = -G2"""
    from owshen_amd import api
    g1 = np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8).reshape(1, 64)
    g2 = np.frombuffer(g2_to_bytes((G2_GEN_X, G2_GEN_Y)), dtype=np.uint8).reshape(1, 128)
    b1 = api.Bases(ctx, 1, ctx.to_device(np.repeat(g1, 2, axis=0)), 8, False)
    b2 = api.Bases(ctx, 2, ctx.to_device(np.repeat(g2, 2, axis=0)), 8, False)
    sc = lambda a, b: ctx.to_device(api.ints_to_bytes([a, b]))
    assert b1.msm(sc(1, 1))[0].tobytes() == g1_to_bytes(G1_DOUBLE)
    assert b1.msm(sc(FR_MODULUS - 1, 1))[0].tobytes() == bytes(64)           # (r - 1) G + G = r G = O
    assert b2.msm(sc(FR_MODULUS - 1, 1))[0].tobytes() == bytes(128)
    neg_g2 = (G2_GEN_X, ((FQ_MODULUS - G2_GEN_Y[0]) % FQ_MODULUS, (FQ_MODULUS - G2_GEN_Y[1]) % FQ_MODULUS))
    assert b2.msm(sc(FR_MODULUS - 1, 0))[0].tobytes() == g2_to_bytes(neg_g2)
    b1.close()
    b2.close()
