"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

RLP encoding of the reference's `Burn` transaction body, restated from
/root/reference/src/types/tx/custom.rs:111-153 (encode) and :155-212 (decode), plus the three trailing fields the
integration patch (ffi/burn_proof.patch) adds so that a burn can carry its withdraw proof:

    [ "burn", burn_id(32), "native", amount_le(32), network, calldata_address(20)                         ]   reference, Token::Native
    [ "burn", burn_id(32), "erc20", amount_le(32), token_address(20), decimals_le(32), symbol, network, calldata_address(20) ]
    [ ...the same..., proof(256), root(32 LE), nullifier_hash(32 LE) ]                                         patched

List lengths follow custom.rs:113,116,127 (`4 + dl` / `7 + dl`, dl = 2 when calldata is present): the patch makes dl = 5
when a proof is attached.  Field elements keep the reference's 32-byte little-endian form
(/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:10); U256 amounts are `as_le_bytes()` (custom.rs:119).
"""


def rlp_encode(item):
    """Ethereum RLP: bytes -> string item, list -> list item"""
    if isinstance(item, (bytes, bytearray)):
        b = bytes(item)
        if len(b) == 1 and b[0] < 0x80:
            return b
        return _length_prefix(len(b), 0x80) + b
    payload = b"".join(rlp_encode(x) for x in item)
    return _length_prefix(len(payload), 0xC0) + payload


def _length_prefix(n, offset):
    if n < 56:
        return bytes([offset + n])
    nb = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return bytes([offset + 55 + len(nb)]) + nb


def rlp_decode(data):
    item, rest = _decode_one(bytes(data))
    if rest:
        raise ValueError("trailing bytes after the RLP item")
    return item


def _decode_one(b):
    if not b:
        raise ValueError("empty input")
    p = b[0]
    if p < 0x80:
        return b[:1], b[1:]
    if p < 0xB8:
        n = p - 0x80
        return b[1:1 + n], b[1 + n:]
    if p < 0xC0:
        ln = p - 0xB7
        n = int.from_bytes(b[1:1 + ln], "big")
        return b[1 + ln:1 + ln + n], b[1 + ln + n:]
    if p < 0xF8:
        n, start = p - 0xC0, 1
    else:
        ln = p - 0xF7
        n, start = int.from_bytes(b[1:1 + ln], "big"), 1 + ln
    payload, out = b[start:start + n], []
    if len(payload) != n:
        raise ValueError("truncated list")
    while payload:
        x, payload = _decode_one(payload)
        out.append(x)
    return out, b[start + n:]


def encode_burn(burn_id, network, amount, token=None, calldata_address=None, proof=None):
    """burn_id: 32 bytes; network: "eth" | "bsc"; amount: int; token: None (native) or (address20, decimals:int, symbol:str);
    calldata_address: 20 bytes or None; proof: None or (proof256, root:int, nullifier_hash:int) -- needs calldata_address."""
    assert len(burn_id) == 32 and network in ("eth", "bsc")
    assert proof is None or calldata_address is not None, "a proof binds the recipient: it needs the calldata address"
    items = [b"burn", bytes(burn_id)]
    if token is None:
        items += [b"native", amount.to_bytes(32, "little")]
    else:
        addr, decimals, symbol = token
        assert len(addr) == 20
        items += [b"erc20", amount.to_bytes(32, "little"), bytes(addr), decimals.to_bytes(32, "little"), symbol.encode()]
    items.append(network.encode())
    if calldata_address is not None:
        assert len(calldata_address) == 20
        items.append(bytes(calldata_address))
    if proof is not None:
        pf, root, nh = proof
        assert len(pf) == 256
        items += [bytes(pf), root.to_bytes(32, "little"), nh.to_bytes(32, "little")]
    return rlp_encode(items)


def decode_burn(data):
    """inverse of encode_burn -> dict (mirrors the index arithmetic of custom.rs:155-212 + the patch)"""
    it = rlp_decode(data)
    if not isinstance(it, list) or it[0] != b"burn":
        raise ValueError("not a burn")
    out = {"burn_id": it[1], "amount": int.from_bytes(it[3], "little")}
    if it[2] == b"native":
        out["token"], network_idx = None, 4
    elif it[2] == b"erc20":
        out["token"], network_idx = (it[4], int.from_bytes(it[5], "little"), it[6].decode()), 7
    else:
        raise ValueError("unknown token type")
    out["network"] = it[network_idx].decode()
    cd = network_idx + 1
    out["calldata_address"] = it[cd] if len(it) > cd else None
    out["proof"] = None
    if len(it) > cd + 1:
        if len(it) != cd + 4 or len(it[cd + 1]) != 256 or len(it[cd + 2]) != 32 or len(it[cd + 3]) != 32:
            raise ValueError("malformed proof fields")
        out["proof"] = (it[cd + 1], int.from_bytes(it[cd + 2], "little"), int.from_bytes(it[cd + 3], "little"))
    return out
