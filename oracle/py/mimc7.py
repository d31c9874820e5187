"""MiMC7 (circomlib convention) + Merkle path/tree.  TEST INFRASTRUCTURE ONLY.

No reference counterpart: the snapshot's only Fr "hash" is a placeholder product
(``/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:202-204``)
and must NOT be used.  Convention fixed here, following the
public circomlib ``mimc7.circom`` / circomlibjs ``mimc7.js``:

* 91 rounds, x -> (x + k + c_i)^7, c_0 = 0, output = x_91 + k;
* c_i (i >= 1) = keccak256^(i+1)("mimc") read big-endian, reduced mod r;
* multi-hash (Miyaguchi-Preneel): r = key; for each input a: r = r + a + E_r(a);
* 2-to-1 Merkle hash H(l, r) = multi_hash([l, r], key = 0).

Weak pin: c_1 equals the first non-zero constant published in circomlib's
mimc7.circom (2088896141...570981, from the surveyor's/author's memory only).
"""
from .fields import R
from .keccak import keccak256

N_ROUNDS = 91
SEED = b"mimc"


def constants(n_rounds=N_ROUNDS, seed=SEED):
    cts = [0] * n_rounds
    c = keccak256(seed)
    for i in range(1, n_rounds):
        c = keccak256(c)
        cts[i] = int.from_bytes(c, "big") % R
    return cts


CONSTANTS = constants()


def permute(x, k):
    """E_k(x): 91 rounds of (x + k + c_i)^7, plus final key addition."""
    r = x % R
    k %= R
    for i in range(N_ROUNDS):
        t = (r + k + CONSTANTS[i]) % R
        t2 = t * t % R
        t4 = t2 * t2 % R
        r = t4 * t2 % R * t % R
    return (r + k) % R


def multi_hash(arr, key=0):
    r = key % R
    for a in arr:
        a %= R
        r = (r + a + permute(a, r)) % R
    return r


def hash2(left, right):
    return multi_hash([left, right], 0)


def merkle_root_from_path(leaf, index, siblings):
    """Walk a path bottom-up; bit i of index = 1 means the node is a right child.

    Returns the list of the depth+1 nodes on the path (leaf first, root last)."""
    nodes = [leaf % R]
    cur = leaf % R
    for lvl, sib in enumerate(siblings):
        if (index >> lvl) & 1:
            cur = hash2(sib, cur)
        else:
            cur = hash2(cur, sib)
        nodes.append(cur)
    return nodes


def tree_build(leaves):
    """Full binary tree; returns list of levels, levels[0] = leaves, last = [root]."""
    n = len(leaves)
    assert n and n & (n - 1) == 0
    levels = [[x % R for x in leaves]]
    while len(levels[-1]) > 1:
        prev = levels[-1]
        levels.append([hash2(prev[2 * i], prev[2 * i + 1]) for i in range(len(prev) // 2)])
    return levels


def zero_hashes(depth, zero_leaf=0):
    """Root of an all-`zero_leaf` subtree at each height (sparse-tree defaults)."""
    z = [zero_leaf % R]
    for _ in range(depth):
        z.append(hash2(z[-1], z[-1]))
    return z


class IncrementalTree:
    """Append-only depth-`depth` Merkle tree kept as a frontier (the "filled subtrees" of deposit contracts):
    frontier[l] is the completed left sibling at level l on the path of the next free position, when that position
    is (or passes through) a right child.  Empty subtrees hash to `zero_hashes(depth, zero_leaf)`.
    The spec for og_mimc7_append_d (the commitment tree `mint_tx` would feed: SURVEY.md 8f-3; no reference code)."""

    def __init__(self, depth, zero_leaf=0):
        self.depth = depth
        self.zeros = zero_hashes(depth, zero_leaf)
        self.frontier = [0] * depth
        self.next_index = 0
        self.root = self.zeros[depth]

    def append(self, leaf):
        idx = self.next_index
        assert idx < 1 << self.depth
        cur = leaf % R
        for l in range(self.depth):
            if (idx >> l) & 1:
                cur = hash2(self.frontier[l], cur)
            else:
                self.frontier[l] = cur
                cur = hash2(cur, self.zeros[l])
        self.root = cur
        self.next_index += 1
        return cur
