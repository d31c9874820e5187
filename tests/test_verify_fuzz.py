"""og_verify takes bytes from the network: `burn_tx` (/root/reference/src/blockchain/tx/burn_tx.rs:11-32, patched) hands it whatever a
withdraw request carried.  Whatever those bytes are it must answer -- accept, refuse (`ok = 0`) or name a malformed verifying key
(OG_ERR_INVALID) -- and never crash or accept: random proofs, every single-byte corruption class of a valid proof (incl. all 256
positions), public inputs at and above r, verifying keys truncated / extended / with a lying n_pub / with corrupted points.
Host-only library (libowshen_verify.so): no GPU, no torch."""
import random
import struct

import pytest

from oracle.py import fields, groth16 as og16
from oracle.py.curve import g1_to_bytes, g2_to_bytes
from tests.r1cs_util import random_r1cs

R = fields.R


@pytest.fixture(scope="module")
def inst():
    n_pub = 2
    n_wires, cons, z = random_r1cs(9, n_pub, seed=5)
    ro = og16.R1CS(n_wires, n_pub, cons)
    rnd = random.Random(99)
    pk, vk = og16.setup(ro, *(rnd.randrange(1, R) for _ in range(5)))
    proof = og16.proof_to_bytes(og16.prove(pk, ro, z, rnd.randrange(R), rnd.randrange(R)))
    blob = (b"OWVK0001" + struct.pack("<Q", n_pub) + g1_to_bytes(vk["alpha_g1"]) + g2_to_bytes(vk["beta_g2"]) + g2_to_bytes(vk["gamma_g2"]) +
            g2_to_bytes(vk["delta_g2"]) + b"".join(g1_to_bytes(p) for p in vk["ic"]))
    return blob, [int(v) for v in z[1:1 + n_pub]], proof


def _answer(blob, pub, proof):
    """True / False / "invalid" -- anything else (a crash) ends the test run"""
    from owshen_amd import verify_only
    try:
        return verify_only.verify(blob, pub, proof)
    except ValueError:
        return "invalid"


def test_proof_bytes(inst):
    blob, pub, proof = inst
    assert _answer(blob, pub, proof) is True
    rnd = random.Random(1)
    for pos in range(256):                                   # one flipped bit at every position: never accepted
        t = bytearray(proof)
        t[pos] ^= 1 << rnd.randrange(8)
        assert _answer(blob, pub, bytes(t)) is False, pos
    for _ in range(200):                                     # random bytes; random bytes below p in every coordinate
        assert _answer(blob, pub, rnd.randbytes(256)) is False
        coords = b"".join(rnd.randrange(fields.P).to_bytes(32, "little") for _ in range(8))
        assert _answer(blob, pub, coords) is False
    for fill in (0x00, 0xFF):
        assert _answer(blob, pub, bytes([fill]) * 256) is False
    # each of A, B, C replaced by the point at infinity (all zero) / by another proof's element
    for lo, hi in ((0, 64), (64, 192), (192, 256)):
        assert _answer(blob, pub, proof[:lo] + bytes(hi - lo) + proof[hi:]) is False
    # coordinates that are the right value + p (non-canonical encodings of the same point)
    for k in range(8):
        v = int.from_bytes(proof[32 * k:32 * k + 32], "little") + fields.P
        if v < 1 << 256:
            assert _answer(blob, pub, proof[:32 * k] + v.to_bytes(32, "little") + proof[32 * k + 32:]) is False, k


def test_public_inputs(inst):
    blob, pub, proof = inst
    assert _answer(blob, [pub[0] + R, pub[1]], proof) in (False, "invalid")          # x + r is not x at this boundary
    assert _answer(blob, [R, pub[1]], proof) in (False, "invalid")
    assert _answer(blob, [(1 << 256) - 1, pub[1]], proof) in (False, "invalid")
    assert _answer(blob, pub[:1], proof) in (False, "invalid")                       # too few / too many
    assert _answer(blob, pub + [0], proof) in (False, "invalid")
    assert _answer(blob, [], proof) in (False, "invalid")


def test_verifying_key_bytes(inst):
    blob, pub, proof = inst
    rnd = random.Random(2)
    for cut in (0, 7, 8, 15, 16, 17, 80, len(blob) - 64, len(blob) - 1):
        assert _answer(blob[:cut], pub, proof) in (False, "invalid"), cut
    assert _answer(blob + b"\0", pub, proof) in (False, "invalid")
    assert _answer(blob + blob[-64:], pub, proof) in (False, "invalid")              # one IC point too many for its n_pub
    for n in (0, 1, 3, 1 << 20, (1 << 64) - 1):                                      # n_pub that the length does not bear out
        assert _answer(blob[:8] + struct.pack("<Q", n) + blob[16:], pub, proof) in (False, "invalid"), n
    assert _answer(b"OWVK0002" + blob[8:], pub, proof) in (False, "invalid")
    for pos in range(16, len(blob), 7):                                              # corrupted key points: refused or named, never accepted
        t = bytearray(blob)
        t[pos] ^= 1 << rnd.randrange(8)
        assert _answer(bytes(t), pub, proof) in (False, "invalid"), pos
    for _ in range(50):
        junk = blob[:16] + rnd.randbytes(len(blob) - 16)
        assert _answer(junk, pub, proof) in (False, "invalid")
    assert _answer(blob, pub, proof) is True                                         # and the library is still in working order
