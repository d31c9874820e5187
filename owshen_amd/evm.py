"""EVM-side encodings of the withdraw path: the words `contracts/WithdrawVerifier.sol` consumes.

The library's byte formats are little-endian (the reference's `Fp::to_repr()`,
/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11) with G2 as x.c0 | x.c1 | y.c0 | y.c1; the EVM
precompiles (EIP-196 / EIP-197) take 32-byte big-endian words with the imaginary part of an Fq2 element FIRST.  This module
is the only place where that conversion happens.  Pure byte shuffling: no arithmetic, and no import of the GPU library
(a host that only submits proofs needs neither torch nor HIP).
"""
import struct


def proof_to_evm_calldata(proof):
    """256-byte proof -> 8 x uint256 big-endian, G2 as (x.c1, x.c0, y.c1, y.c0): the argument order of
    a snarkjs-style Solidity verifier / the EIP-197 precompile."""
    proof = bytes(proof)
    assert len(proof) == 256
    f = [proof[i * 32:(i + 1) * 32][::-1] for i in range(8)]  # LE -> BE
    return f[0] + f[1] + f[3] + f[2] + f[5] + f[4] + f[6] + f[7]

VK_MAGIC = b"OWVK0001"


def _be(le32):
    return bytes(le32)[::-1]


def g1_words(p64):
    """64-byte LE G1 point -> [x, y] as ints"""
    return [int.from_bytes(p64[0:32], "little"), int.from_bytes(p64[32:64], "little")]


def g2_words(p128):
    """128-byte LE G2 point (x.c0 | x.c1 | y.c0 | y.c1) -> [x.c1, x.c0, y.c1, y.c0] as ints"""
    c = [int.from_bytes(p128[32 * i:32 * i + 32], "little") for i in range(4)]
    return [c[1], c[0], c[3], c[2]]


def vk_to_evm_words(vk_blob):
    """"OWVK0001" blob -> the 14 + 2 (n_pub + 1) uint256 words of WithdrawVerifier's constructor:
    alpha (2) | beta (4) | gamma (4) | delta (4) | IC_0 .. IC_n_pub (2 each)"""
    vk_blob = bytes(vk_blob)
    assert vk_blob[:8] == VK_MAGIC
    n_pub = struct.unpack("<Q", vk_blob[8:16])[0]
    o = 16
    assert len(vk_blob) == o + 64 + 3 * 128 + (n_pub + 1) * 64
    words = g1_words(vk_blob[o:o + 64])
    for k in range(3):
        words += g2_words(vk_blob[o + 64 + 128 * k:o + 64 + 128 * (k + 1)])
    ic = o + 64 + 384
    for i in range(n_pub + 1):
        words += g1_words(vk_blob[ic + 64 * i:ic + 64 * (i + 1)])
    return words


def public_inputs_to_evm_words(public_inputs):
    """n_pub x 32-byte LE field elements (or ints) -> list of ints"""
    out = []
    for x in public_inputs:
        out.append(int(x) if isinstance(x, int) else int.from_bytes(bytes(x), "little"))
    return out


def proof_words(proof256):
    """256-byte proof -> the 8 uint256 words of verifyProof's `proof` argument"""
    cd = proof_to_evm_calldata(bytes(proof256))
    return [int.from_bytes(cd[32 * i:32 * i + 32], "big") for i in range(8)]


def vk_constructor_calldata(vk_blob):
    """ABI encoding of the constructor argument uint256[28] (static array: 28 words back to back)"""
    words = vk_to_evm_words(vk_blob)
    assert len(words) == 28, "WithdrawVerifier is written for n_pub = 6"
    return b"".join(w.to_bytes(32, "big") for w in words)


def verify_calldata(proof256, public_inputs):
    """ABI encoding of verifyProof(uint256[8], uint256[6]) arguments (selector not included)"""
    ws = proof_words(proof256) + public_inputs_to_evm_words(public_inputs)
    assert len(ws) == 14
    return b"".join(w.to_bytes(32, "big") for w in ws)
