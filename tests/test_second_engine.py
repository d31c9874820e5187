"""A second, independent implementation of the MiMC7 layer of the oracle: oracle/js/mimc7_second.js, written from the
circomlib text alone for V8's BigInt (node 12 is in the image) with its own Keccak-f[1600].  The Python oracle must agree
with it on the round constants, the permutation, MultiMiMC7 / hash2 and the depth-32 zero-hash chain -- and both must
reproduce the public go-iden3-crypto test vectors (mimc7/mimc7_test.go; recalled, not fetched: there is no network).
Parity for the Groth16 layer stays "unpinned" (DESIGN.md 0); this removes "one author, one big-integer engine" from N5.

The same for the pairing and the Groth16 equation: oracle/js/bn254_pairing_second.js (a flat Fp12 = Fp[w]/(w^12 - 18 w^6 + 82),
the G2 point untwisted into E(Fp12), Frobenius as honest p-th powers, a naive final exponentiation -- none of the Python
oracle's tower formulas) must give the Python oracle's e(G1, G2) coefficient by coefficient, and must accept the proofs the
PRODUCT's kernels produce (on the CPU interpreter here): a random circuit's and a withdraw statement's, and refuse them for
another public input or with a flipped byte."""
import json
import os
import shutil
import subprocess

import pytest

from oracle.py import fields, mimc7
from oracle.py.keccak import keccak256

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node") or shutil.which("nodejs")

pytestmark = pytest.mark.skipif(NODE is None, reason="node is not installed")


@pytest.fixture(scope="module")
def js():
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "mimc7_second.js")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout)


def test_second_keccak_agrees_and_matches_published_answers(js):
    assert js["keccak_empty"] == keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert js["keccak_abc"] == keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    assert js["keccak_mimc"] == keccak256(b"mimc").hex()
    assert (js["rc0"], js["rc23"]) == ("1", "8000000080008008")     # first / last Keccak round constants (FIPS 202, table 5)


def test_second_engine_round_constants(js):
    assert int(js["c1"]) == mimc7.CONSTANTS[1] and int(js["c2"]) == mimc7.CONSTANTS[2] and int(js["c90"]) == mimc7.CONSTANTS[90]
    assert int(js["constants_sum"]) == sum(mimc7.CONSTANTS) % fields.R        # all 91 of them


def test_second_engine_hashes_and_public_vectors(js):
    # go-iden3-crypto mimc7_test.go: MIMC7Hash(1, 2); Hash([1, 2], nil); Hash([12, 45, 78, 41], nil)
    published = {"hash_1_2": 10594780656576967754230020536574539122676596303354946869887184401991294982664,
                 "multihash_1_2": 5233261170300319370386085858846328736737478911451874673953613863492170606314,
                 "multihash_12_45_78_41": 18226366069841799622585958305961373004333097209608110160936134895615261821931}
    ours = {"hash_1_2": mimc7.permute(1, 2), "multihash_1_2": mimc7.multi_hash([1, 2]),
            "multihash_12_45_78_41": mimc7.multi_hash([12, 45, 78, 41])}
    for k, v in published.items():
        assert int(js[k]) == ours[k] == v, k
    assert mimc7.hash2(1, 2) == published["multihash_1_2"]                      # the tree's two-to-one hash IS MultiMiMC7([l, r], 0)


def test_second_engine_zero_hash_chain_depth_32(js):
    z = mimc7.zero_hashes(32)
    assert [int(x) for x in js["zeros"]] == list(z)[:33]
    assert len(set(z)) == len(z)


# ---- the pairing and whole proofs ------------------------------------------------------------------------------------------

def _flat_fp12(t):
    """oracle/py tower element ((a0, a1, a2), (b0, b1, b2)), a_j / b_j in Fp2 = (x, y) -> 12 coefficients in powers of w, where
    the tower's v = w^2, its w = w, and u = w^6 - 9: (x + y u) v^j w^k sits at w^(2j + k) as (x - 9 y) and at w^(2j + k + 6) as y"""
    out = [0] * 12
    for k, c6 in enumerate(t):
        for j, (x, y) in enumerate(c6):
            m = 2 * j + k
            out[m] = (out[m] + x - 9 * y) % fields.P
            out[m + 6] = (out[m + 6] + y) % fields.P
    return out


def _g1(b):
    return [str(int.from_bytes(b[0:32], "little")), str(int.from_bytes(b[32:64], "little"))]


def _g2(b):
    v = [str(int.from_bytes(b[i * 32:(i + 1) * 32], "little")) for i in range(4)]
    return [[v[0], v[1]], [v[2], v[3]]]


def _proof_json(p):
    p = bytes(p)
    return {"a": _g1(p[0:64]), "b": _g2(p[64:192]), "c": _g1(p[192:256])}


def _vk_json(vk):
    return {"alpha": _g1(vk["alpha_g1"]), "beta": _g2(vk["beta_g2"]), "gamma": _g2(vk["gamma_g2"]), "delta": _g2(vk["delta_g2"]),
            "ic": [_g1(vk["ic"][i].tobytes()) for i in range(vk["ic"].shape[0])]}


def _run_pairing(req):
    out = subprocess.run([NODE, os.path.join(ROOT, "oracle", "js", "bn254_pairing_second.js")], input=json.dumps(req),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr
    return json.loads(out.stdout)


def test_second_engine_pairing_value_equals_the_tower_oracle():
    from oracle.py import pairing
    from oracle.py.curve import G1_GEN, G2_GEN
    js = _run_pairing({})
    assert [int(c) for c in js["e_g1_g2"]] == _flat_fp12(pairing.pairing(G1_GEN, G2_GEN))
    assert js["e_not_one"] and js["e_order_r"] and js["bilinear_g1"]


def test_second_engine_verifies_the_products_proofs():
    """proofs made by the product's kernels (CPU interpreter), keys from the product's own set-up: a random 40-gate circuit and a
    depth-1 withdraw statement (six public inputs)"""
    import random
    import numpy as np
    from owshen_amd import circuit, groth16 as g16
    from tests import emu, withdraw_cases as wc
    from tests.r1cs_util import random_r1cs
    ctx = emu.Ctx()
    try:
        rnd = random.Random(12)
        # a random circuit, two public inputs
        n_pub = 2
        n_wires, cons, z = random_r1cs(40, n_pub, seed=21, bool_every=4)
        blob, vk = g16.setup(ctx, g16.R1CS.from_constraints(n_wires, n_pub, cons), *(rnd.randrange(1, fields.R) for _ in range(5)))
        pk = g16.ProvingKey(ctx, blob)
        wit = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in z), dtype=np.uint8).reshape(1, -1, 32).copy()
        proof = pk.prove_batch(wit, [(rnd.randrange(fields.R), rnd.randrange(fields.R))])[0].tobytes()
        pk.close()
        pub = [str(z[1]), str(z[2])]
        flipped = bytearray(proof)
        flipped[200] ^= 1                                   # C.x: no longer on the curve (or another point)
        res = _run_pairing({"vk": _vk_json(vk), "proofs": [
            {"public": pub, "proof": _proof_json(proof)},
            {"public": [pub[0], str((z[2] + 1) % fields.R)], "proof": _proof_json(proof)},
            {"public": pub, "proof": _proof_json(flipped)}]})
        assert res["proofs"] == [True, False, False]
        # a withdraw statement: inputs -> witness -> proof inside the product, six public inputs handed back by the call
        depth = 1
        _r1, _blob, wvk, wpk, close = wc._key(ctx, depth, 2, 3)
        packed = np.stack([wc._pack(circuit, wc._inputs(rnd, depth))])
        proofs, pubs = circuit.prove_from_inputs(ctx, wpk, depth, ctx.to_device(packed), [(rnd.randrange(fields.R), rnd.randrange(fields.R))],
                                                 2, 3, return_public=True)
        close()
        wpub = [str(int.from_bytes(pubs[0][i].tobytes(), "little")) for i in range(6)]
        other = list(wpub)
        other[2] = str((int(other[2]) + 1) % fields.R)      # someone else's recipient
        res = _run_pairing({"vk": _vk_json(wvk), "proofs": [{"public": wpub, "proof": _proof_json(proofs[0].tobytes())},
                                                            {"public": other, "proof": _proof_json(proofs[0].tobytes())}]})
        assert res["proofs"] == [True, False]
    finally:
        ctx.close()
