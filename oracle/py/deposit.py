"""The deposit statement -- specification + witness generator.  TEST INFRASTRUCTURE ONLY.

No reference counterpart: the snapshot's deposit is a ledger credit on the word of an L1 transaction hash
(``/root/reference/src/services/api_services/deposit.rs:32-154`` -> ``/root/reference/src/blockchain/tx/mint_tx.rs:11-49``) and
contains no circuit.  BASELINE.json's north_star names "deposit/withdraw circuits"; this file DEFINES the deposit
half in the same plain form as oracle/py/withdraw.py (whose helpers it reuses), and the product's builders
(owshen_amd/circuit.py, og_deposit_r1cs in keygen.hip) and the HIP witness kernel (witness.hip) are checked against it.

Statement (public: commitment, depositor):
    "I know (nullifier, secret) such that commitment = H(nullifier, secret)"
with H = MultiMiMC7 2-to-1 (oracle/py/mimc7.py).  `commitment` is the inner commitment c a depositor hands to the ledger, which
forms the note leaf H(c, H(amount, token)) itself from the asset it credits (oracle/py/withdraw.py, "Deposit side").  What the
proof adds: a c nobody knows the opening of -- a typo, a truncated hex string, somebody else's c copied from the mempool, whose
nullifier the copier does not know -- makes a note that can never be withdrawn; with the proof the ledger appends only notes whose
depositor demonstrably holds (nullifier, secret).  `depositor` -- the account the reference's DepositRequest credits
(``deposit.rs:19-24``: `address`) -- is bound by a square constraint (a public input that occurs in no constraint has the point at
infinity as its IC base and would not be bound), so a proof lifted from another request does not verify under a different address.

Wire order (the contract the implementations share):
    0 one | 1 commitment | 2 depositor                              (n_pub = 2)
    3 nullifier | 4 secret | 5 depositor^2
    6.. the hash gadget: perm0 91 x (t^2, t^4, t^6, t^7) | k1 | perm1 91 x (t^2, t^4, t^6, t^7)     (its `out` is wire 1)
735 wires, 731 constraints (the square, then the gadget's 730), NTT domain 2^10.
"""
from .fields import R
from . import mimc7
from .withdraw import _CS, _hash2

N_PUB = 2
N_WIRES = 6 + 729
N_CONSTRAINTS = 1 + 730


def shape():
    """(n_wires, n_constraints)"""
    return N_WIRES, N_CONSTRAINTS


def commitment_of(nullifier, secret):
    return mimc7.hash2(nullifier, secret)


def build(nullifier, secret, depositor):
    """returns (n_wires, n_pub, constraints, witness z)."""
    cs = _CS()
    w_c = cs.alloc(commitment_of(nullifier, secret))
    w_dep = cs.alloc(depositor)
    w_null, w_sec = cs.alloc(nullifier), cs.alloc(secret)
    w_dsq = cs.alloc(depositor * depositor)
    cs.enforce({w_dep: 1}, {w_dep: 1}, {w_dsq: 1})
    out = _hash2(cs, {w_null: 1}, {w_sec: 1}, out_wire=w_c)
    assert out == w_c and len(cs.z) == N_WIRES and len(cs.constraints) == N_CONSTRAINTS
    assert all(0 <= v < R for v in cs.z)
    return N_WIRES, N_PUB, cs.constraints, cs.z
