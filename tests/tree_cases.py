"""Batched append to the incremental MiMC7 tree (og_mimc7_append_d) against oracle/py/mimc7.IncrementalTree and the
dense tree builder; shared by the CPU-interpreter run and the GPU run."""
import random

import numpy as np

from oracle.py import fields, mimc7


def _tob(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32).copy()


def _toi(arr):
    return [int.from_bytes(r.tobytes(), "little") for r in np.asarray(arr).reshape(-1, 32)]


def case_append_matches_incremental_tree(ctx, depth, batch_sizes, seed):
    rnd = random.Random(seed)
    ref = mimc7.IncrementalTree(depth)
    frontier = ctx.to_device(np.zeros((depth, 32), dtype=np.uint8))
    n = 0
    all_leaves = []
    for k in batch_sizes:
        leaves = [rnd.randrange(fields.R) for _ in range(k)]
        all_leaves += leaves
        for x in leaves:
            ref.append(x)
        frontier, root = ctx.mimc7_append(depth, frontier, n, ctx.to_device(_tob(leaves)))
        n += k
        assert _toi(ctx.to_host(root)) == [ref.root], (depth, n)
        got_f = _toi(ctx.to_host(frontier))
        for l in range(depth):
            if (n >> l) & 1:  # the entries a future append can read
                assert got_f[l] == ref.frontier[l], (depth, n, l)
    if depth <= 10:  # and the root is the dense tree's root over the zero-padded leaves
        full = mimc7.tree_build(all_leaves + [0] * ((1 << depth) - len(all_leaves)))[-1][0]
        assert ref.root == full
