"""A second, independent implementation of the MiMC7 layer of the oracle: oracle/js/mimc7_second.js, written from the
circomlib text alone for V8's BigInt (node 12 is in the image) with its own Keccak-f[1600].  The Python oracle must agree
with it on the round constants, the permutation, MultiMiMC7 / hash2 and the depth-32 zero-hash chain -- and both must
reproduce the public go-iden3-crypto test vectors (mimc7/mimc7_test.go; recalled, not fetched: there is no network).
Parity for the Groth16 layer stays "unpinned" (DESIGN.md 0); this removes "one author, one big-integer engine" from N5."""
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
