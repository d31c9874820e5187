"""contracts/WithdrawVerifier.sol as a model.  TEST INFRASTRUCTURE ONLY.

No solc in this image, so the verifier contract's exact word layout is modelled: a line-by-line Python transcription of
`verifyProof` whose precompiles 0x06 / 0x07 / 0x08 are the EIP-196 / EIP-197 definitions evaluated by the oracle
(oracle/py/curve.py, pairing.py).  tests/test_evm_words.py pins the emitted words against it; bench.py's `plumbing` workload
(BASELINE.json configs[0]) runs a CPU-made proof through it.  The seam it stands in for:
/root/reference/contracts/src/Owshen.sol:66-78 (`_processWithdraw`, today an ECDSA check)."""
from . import fields, pairing
from .curve import G1, G2

Q, R = fields.P, fields.R


# ---- the three precompiles, per EIP-196 / EIP-197 (inputs and outputs are big-endian words) ----------------------
def _g1(x, y):
    if x >= Q or y >= Q:
        raise ValueError("coordinate not a field element")
    if x == 0 and y == 0:
        return None
    if not G1.is_on_curve((x, y)):
        raise ValueError("not on curve")
    return (x, y)


def ec_add(w):      # 0x06
    p = G1.add(_g1(w[0], w[1]), _g1(w[2], w[3]))
    return [0, 0] if p is None else [p[0], p[1]]


def ec_mul(w):      # 0x07
    p = G1.mul(_g1(w[0], w[1]), w[2])
    return [0, 0] if p is None else [p[0], p[1]]


def ec_pairing(w):  # 0x08: k tuples (G1.x, G1.y, G2.x_imag, G2.x_real, G2.y_imag, G2.y_real)
    assert len(w) % 6 == 0
    pairs = []
    for k in range(0, len(w), 6):
        p = _g1(w[k], w[k + 1])
        xi, xr, yi, yr = w[k + 2:k + 6]
        if max(xi, xr, yi, yr) >= Q:
            raise ValueError("coordinate not a field element")
        q = None if (xi | xr | yi | yr) == 0 else ((xr, xi), (yr, yi))
        if q is not None and (not G2.is_on_curve(q) or G2.add(G2.mul(q, R - 1), q) is not None):
            raise ValueError("G2 point not in the r-torsion")
        if p is not None and q is not None:
            pairs.append((p, q))
    return 1 if pairing.pairing_check(pairs) else 0


def verify_proof_model(vk, proof, inp):
    """line-by-line transcription of WithdrawVerifier.verifyProof (vk: 28 words, proof: 8, inp: 6)"""
    if any(x >= Q for x in proof):
        return False
    acc = [vk[14], vk[15]]
    try:
        for i in range(6):
            if inp[i] >= R:
                return False
            term = ec_mul([vk[16 + 2 * i], vk[17 + 2 * i], inp[i]])
            acc = ec_add([acc[0], acc[1], term[0], term[1]])
        neg_ay = 0 if proof[0] == 0 and proof[1] == 0 else (Q - proof[1]) % Q
        p = [proof[0], neg_ay, proof[2], proof[3], proof[4], proof[5],
             vk[0], vk[1], vk[2], vk[3], vk[4], vk[5],
             acc[0], acc[1], vk[6], vk[7], vk[8], vk[9],
             proof[6], proof[7], vk[10], vk[11], vk[12], vk[13]]
        return ec_pairing(p) == 1
    except ValueError:      # a failing precompile call makes staticcall return 0
        return False
