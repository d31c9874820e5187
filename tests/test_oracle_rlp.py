"""the `Burn` RLP body with the proof fields of ffi/burn_proof.patch, pinned on the oracle side
(oracle/py/rlp_burn.py restates /root/reference/src/types/tx/custom.rs:111-212)."""
import pytest

from oracle.py import rlp_burn as rb

BURN_ID = bytes(range(32))
ADDR = bytes.fromhex("00112233445566778899aabbccddeeff00112233")
TOKEN = (bytes.fromhex("a0b86991c6218b36c1d19d4a2e9eb0ce3606eb48"), 6, "USDC")


def test_rlp_primitives_match_the_ethereum_spec_examples():
    # the examples of the RLP specification (ethereum.org "RLP" page): "dog", ["cat","dog"], "", [], 15, 1024, a 56-byte string
    assert rb.rlp_encode(b"dog") == b"\x83dog"
    assert rb.rlp_encode([b"cat", b"dog"]) == b"\xc8\x83cat\x83dog"
    assert rb.rlp_encode(b"") == b"\x80" and rb.rlp_encode([]) == b"\xc0"
    assert rb.rlp_encode(b"\x0f") == b"\x0f" and rb.rlp_encode(b"\x04\x00") == b"\x82\x04\x00"
    s = b"Lorem ipsum dolor sit amet, consectetur adipisicing elit"
    assert rb.rlp_encode(s) == b"\xb8\x38" + s
    assert rb.rlp_decode(rb.rlp_encode([[], [[]], [[], [[]]]])) == [[], [[]], [[], [[]]]]


@pytest.mark.parametrize("token", [None, TOKEN])
def test_reference_shape_round_trips_and_has_the_reference_list_length(token):
    enc = rb.encode_burn(BURN_ID, "eth", 10**18, token, ADDR)
    items = rb.rlp_decode(enc)
    assert len(items) == (4 if token is None else 7) + 2          # custom.rs:113,116,127 with dl = 2
    d = rb.decode_burn(enc)
    assert d == {"burn_id": BURN_ID, "amount": 10**18, "token": token, "network": "eth", "calldata_address": ADDR, "proof": None}
    assert items[3] == (10**18).to_bytes(32, "little")            # U256::as_le_bytes (custom.rs:119)


@pytest.mark.parametrize("token", [None, TOKEN])
def test_patched_shape_carries_the_256_byte_proof(token):
    proof = bytes((7 * i + 3) % 256 for i in range(256))
    root, nh = 0x1234 << 200, 0x5678 << 100
    enc = rb.encode_burn(BURN_ID, "bsc", 5, token, ADDR, (proof, root, nh))
    items = rb.rlp_decode(enc)
    assert len(items) == (4 if token is None else 7) + 5          # the patch: dl = 5 with a proof
    assert items[-3] == proof and items[-2] == root.to_bytes(32, "little") and items[-1] == nh.to_bytes(32, "little")
    # a 256-byte string takes the long-string header b9 01 00
    assert (b"\xb9\x01\x00" + proof) in enc
    d = rb.decode_burn(enc)
    assert d["proof"] == (proof, root, nh) and d["calldata_address"] == ADDR and d["network"] == "bsc"
    # the unpatched prefix is unchanged: a node that ignores the tail reads the same burn
    base = rb.rlp_decode(rb.encode_burn(BURN_ID, "bsc", 5, token, ADDR))
    assert items[: len(base)] == base


def test_golden_native_burn_with_proof():
    """pins the byte layout (a change here is a consensus change)"""
    enc = rb.encode_burn(BURN_ID, "eth", 1, None, ADDR, (bytes(256), 2, 3))
    assert enc[:2] == b"\xf9\x01"                                  # long list, 2-byte length
    assert enc.hex().startswith("f901ac846275726ea0000102")
    assert len(enc) == 3 + 0x01ac


def test_malformed_proof_tail_is_rejected():
    items = rb.rlp_decode(rb.encode_burn(BURN_ID, "eth", 1, None, ADDR, (bytes(256), 2, 3)))
    with pytest.raises(ValueError):
        rb.decode_burn(rb.rlp_encode(items[:-1]))                 # proof without nullifier hash
    with pytest.raises(ValueError):
        rb.decode_burn(rb.rlp_encode(items[:-3] + [bytes(255)] + items[-2:]))
