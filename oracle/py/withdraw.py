"""The withdraw circuit -- specification + witness generator.  TEST INFRASTRUCTURE ONLY.

No reference counterpart: the snapshot's withdraw is an ECDSA-authorised burn
(``/root/reference/src/services/api_services/withdraw.rs:27-71``,
``/root/reference/src/blockchain/tx/burn_tx.rs:11-32``) and contains no circuit
(SURVEY.md 0.1).  This file DEFINES the statement the prover path works on, in the plainest
possible form (a constraint-system builder that allocates wires and evaluates them as it
goes); the product's vectorised builder (owshen_amd/circuit.py) and the HIP witness kernel
(owshen_amd/csrc/witness.hip) are checked against it.

Statement (public: root, nullifier_hash, recipient, amount, token, chain_id):
    "I know (nullifier, secret, path) such that
        leaf = H(H(nullifier, secret), H(amount, token)) is under `root` at `index` (depth D MiMC7 tree),
        nullifier_hash = H(nullifier, 0)"
with H = MultiMiMC7 2-to-1 (oracle/py/mimc7.py).  `recipient` and `chain_id` are bound by a square constraint each (a
public input that occurs in no constraint has the point at infinity as its IC base and would not be bound by the proof).

What the six inputs bind, against the gate this replaces (/root/reference/contracts/src/Owshen.sol:69 signs
keccak256(abi.encode(msg.sender, _tokenAddress, _amount, _id, block.chainid))): recipient = msg.sender, token, amount,
chain id -- and nullifier_hash takes the place of the replay id `_id`.  `token` is the ERC-20 address as an integer (0 for
the native coin, /root/reference/contracts/src/Owshen.sol:51-57), and it sits INSIDE the leaf: a note is a claim on
`amount` of `token`, so a note of a worthless token cannot be withdrawn as another asset.

Deposit side (who forms the leaf): the depositor hands over only the inner commitment c = H(nullifier, secret); the
ledger -- the sequencer's mint path, /root/reference/src/blockchain/tx/mint_tx.rs:11-49, which already knows the token and
the amount it credits -- computes leaf = H(c, H(amount, token)) itself and appends it (og_mimc7_append_d).  The asset half of
the leaf is therefore never user-claimed and nothing about the ASSET needs proving; what is left to prove at deposit time --
that the depositor knows the opening of c -- is the deposit statement of oracle/py/deposit.py (round 6).

Wire order (the contract the three implementations share):
    0 one | 1 root | 2 nullifier_hash | 3 recipient | 4 amount | 5 token | 6 chain_id      (n_pub = 6)
    7 nullifier | 8 secret | 9..9+D-1 siblings | 9+D..9+2D-1 index bits | recipient^2 | chain_id^2
    hash gadgets in the order  inner, asset, leaf, nullifier_hash, level 0 .. level D-1
      level l first allocates `left_l`; every gadget then allocates
      perm0: 91 x (t^2, t^4, t^6, t^7) | k1 | perm1: 91 x (t^2, t^4, t^6, t^7) | out
      (`out` is not allocated when it is a public wire: nullifier_hash, root)
    padding (synthetic, sizes the statement to BASELINE.json's "MSM ~2^20, NTT 2^17"):
      n_pad3 gates  (p + 1) * (q + 2) = w       with fresh p, q, w
      n_pad2 gates  (p + 1) * (prev + 2) = w    with fresh p, w; prev = the previous gate's w, or the
                    constant-one wire at the start of every 64-gate segment
      free pad values: x = seed + (wire index); v = x^5; every pad wire with index % 5 == 0 is the parity bit
      of v instead (boolean wires, ~10 % of the padding like a real witness).
Constraint order follows the same sequence; see `build`.
"""
from .fields import R
from . import mimc7

N_PUB = 6
PAD_SEGMENT = 64


class _CS:
    def __init__(self):
        self.z = [1]
        self.constraints = []

    def alloc(self, value):
        self.z.append(value % R)
        return len(self.z) - 1

    def val(self, lc):
        return sum(c * self.z[w] for w, c in lc.items()) % R

    def enforce(self, a, b, c):
        assert self.val(a) * self.val(b) % R == self.val(c), "witness generator bug"
        self.constraints.append((dict(a), dict(b), dict(c)))


def _lc_add(*lcs):
    out = {}
    for lc in lcs:
        for w, c in lc.items():
            out[w] = (out.get(w, 0) + c) % R
    return {w: c for w, c in out.items() if c}


def _perm(cs, x_lc, k_lc):
    """91 rounds; returns the wire holding x_91."""
    cur = x_lc
    for i in range(mimc7.N_ROUNDS):
        t = _lc_add(cur, k_lc, {0: mimc7.CONSTANTS[i]})
        tv = cs.val(t)
        t2 = cs.alloc(tv * tv)
        cs.enforce(t, t, {t2: 1})
        t4 = cs.alloc(cs.z[t2] ** 2)
        cs.enforce({t2: 1}, {t2: 1}, {t4: 1})
        t6 = cs.alloc(cs.z[t4] * cs.z[t2])
        cs.enforce({t4: 1}, {t2: 1}, {t6: 1})
        t7 = cs.alloc(cs.z[t6] * tv)
        cs.enforce({t6: 1}, t, {t7: 1})
        cur = {t7: 1}
    return cur


def _hash2(cs, l_lc, r_lc, out_wire=None):
    x91 = _perm(cs, l_lc, {})
    k1_lc = _lc_add(l_lc, x91)
    k1 = cs.alloc(cs.val(k1_lc))
    cs.enforce(k1_lc, {0: 1}, {k1: 1})
    y91 = _perm(cs, r_lc, {k1: 1})
    out_lc = _lc_add({k1: 2}, r_lc, y91)
    if out_wire is None:
        out_wire = cs.alloc(cs.val(out_lc))
    cs.enforce(out_lc, {0: 1}, {out_wire: 1})
    return out_wire


def pad_value(seed, wire):
    x = (seed + wire) % R
    v = pow(x, 5, R)
    return v & 1 if wire % 5 == 0 else v


def shape(depth, n_pad3=0, n_pad2=0):
    """(n_wires, n_constraints)"""
    hashes = 4 + depth
    wires = 1 + N_PUB + 2 + 2 * depth + 2 + depth + hashes * 730 - 2 + 3 * n_pad3 + 2 * n_pad2
    cons = 2 + 2 * depth + hashes * 730 + n_pad3 + n_pad2
    return wires, cons


def pad_for(depth, n_wires, n_constraints):
    """(n_pad3, n_pad2) hitting exactly n_wires wires and n_constraints constraints."""
    w0, c0 = shape(depth)
    P, W = n_constraints - c0, n_wires - w0
    x = W - 2 * P
    assert 0 <= x <= P, "shape not reachable with 2- and 3-wire padding gates"
    return x, P - x


def leaf_of(nullifier, secret, amount, token):
    """the note commitment the ledger appends at deposit time: H(H(nullifier, secret), H(amount, token))"""
    return mimc7.hash2(mimc7.hash2(nullifier, secret), mimc7.hash2(amount, token))


def build(depth, nullifier, secret, amount, recipient, index, siblings, pad_seed=0, n_pad3=0, n_pad2=0, token=0, chain_id=0):
    """returns (n_wires, n_pub, constraints, witness z)."""
    assert len(siblings) == depth >= 1
    cs = _CS()
    leaf = leaf_of(nullifier, secret, amount, token)
    root = mimc7.merkle_root_from_path(leaf, index, siblings)[-1]
    nh = mimc7.hash2(nullifier, 0)
    w_root, w_nh, w_rec, w_amt = cs.alloc(root), cs.alloc(nh), cs.alloc(recipient), cs.alloc(amount)
    w_tok, w_chain = cs.alloc(token), cs.alloc(chain_id)
    w_null, w_sec = cs.alloc(nullifier), cs.alloc(secret)
    w_sib = [cs.alloc(s) for s in siblings]
    w_bit = [cs.alloc((index >> l) & 1) for l in range(depth)]
    w_rsq = cs.alloc(recipient * recipient)
    w_csq = cs.alloc(chain_id * chain_id)
    cs.enforce({w_rec: 1}, {w_rec: 1}, {w_rsq: 1})
    cs.enforce({w_chain: 1}, {w_chain: 1}, {w_csq: 1})
    inner = _hash2(cs, {w_null: 1}, {w_sec: 1})
    asset = _hash2(cs, {w_amt: 1}, {w_tok: 1})
    cur = _hash2(cs, {inner: 1}, {asset: 1})
    assert cs.z[cur] == leaf
    _hash2(cs, {w_null: 1}, {}, out_wire=w_nh)
    for l in range(depth):
        b, s = w_bit[l], w_sib[l]
        cs.enforce({b: 1}, {b: 1, 0: R - 1}, {})
        left = cs.alloc(cs.z[s] if cs.z[b] else cs.z[cur])
        cs.enforce({b: 1}, _lc_add({s: 1}, {cur: R - 1}), _lc_add({left: 1}, {cur: R - 1}))
        right = _lc_add({s: 1}, {cur: 1}, {left: R - 1})
        cur = _hash2(cs, {left: 1}, right, out_wire=w_root if l == depth - 1 else None)
    prev = None
    for g in range(n_pad3):
        base = len(cs.z)
        p, q = cs.alloc(pad_value(pad_seed, base)), cs.alloc(pad_value(pad_seed, base + 1))
        w = cs.alloc((cs.z[p] + 1) * (cs.z[q] + 2))
        cs.enforce({p: 1, 0: 1}, {q: 1, 0: 2}, {w: 1})
        prev = w
    for g in range(n_pad2):
        base = len(cs.z)
        p = cs.alloc(pad_value(pad_seed, base))
        pw = 0 if g % PAD_SEGMENT == 0 else prev
        w = cs.alloc((cs.z[p] + 1) * (cs.z[pw] + 2))
        cs.enforce({p: 1, 0: 1}, _lc_add({pw: 1}, {0: 2}), {w: 1})
        prev = w
    assert (len(cs.z), len(cs.constraints)) == shape(depth, n_pad3, n_pad2)
    return len(cs.z), N_PUB, cs.constraints, cs.z
