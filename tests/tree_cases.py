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


def case_host_append_equals_the_kernels(ctx, depth, batch_sizes, seed):
    """og_set_host_chains: an append of a handful of leaves walks its hashes on the host -- same frontier bytes (every entry, not only
    the ones a later append reads), same root as the level kernels, step by step, and as the oracle's incremental tree"""
    rnd = random.Random(seed)
    ref = mimc7.IncrementalTree(depth)
    f_gpu = f_host = ctx.to_device(np.zeros((depth, 32), dtype=np.uint8))
    n = 0
    try:
        for k in batch_sizes:
            leaves = [rnd.randrange(fields.R) for _ in range(k)]
            for x in leaves:
                ref.append(x)
            lv = ctx.to_device(_tob(leaves))
            ctx.set_host_chains(0)
            f_gpu, root_gpu = ctx.mimc7_append(depth, f_gpu, n, lv)
            ctx.set_host_chains(max(batch_sizes))
            f_host, root_host = ctx.mimc7_append(depth, f_host, n, lv)
            n += k
            assert bytes(ctx.to_host(root_host)) == bytes(ctx.to_host(root_gpu)) and _toi(ctx.to_host(root_host)) == [ref.root], (depth, n)
            gf, hf = _toi(ctx.to_host(f_gpu)), _toi(ctx.to_host(f_host))
            for l in range(depth):
                if (n >> l) & 1:
                    assert gf[l] == hf[l] == ref.frontier[l], (depth, n, l)
    finally:
        ctx.set_host_chains(0)


def case_one_and_two_lanes_per_hash_agree(ctx, monkeypatch, n_hash, n_paths, depth, n_leaves, witness_depth):
    """OG_MIMC_PAIR = 0 | 1: one lane per hash, or a lane pair per hash (three multiplications deep per round instead of
    four; the form the library picks when a launch cannot fill the chip) -- same bytes from the 2-to-1 hash (ragged, odd
    count: the last pair), the Merkle paths, the tree and the withdraw witness, and the oracle's values"""
    from owshen_amd import circuit
    rnd = random.Random(n_hash)
    l = [0, 1, fields.R - 1] + [rnd.randrange(fields.R) for _ in range(n_hash - 3)]
    r = [0, 2, fields.R - 1] + [rnd.randrange(fields.R) for _ in range(n_hash - 3)]
    leaves = [rnd.randrange(fields.R) for _ in range(n_paths)]
    idx = np.array([rnd.randrange(1 << depth) for _ in range(n_paths)], dtype=np.uint64)
    sib = [[rnd.randrange(fields.R) for _ in range(depth)] for _ in range(n_paths)]
    tree_leaves = [rnd.randrange(fields.R) for _ in range(n_leaves)]
    recs = np.stack([circuit.pack_inputs(rnd.randrange(fields.R), rnd.randrange(fields.R), 5 + k, 6, rnd.randrange(fields.R),
                                         rnd.randrange(1 << witness_depth), [rnd.randrange(fields.R) for _ in range(witness_depth)],
                                         token=808, chain_id=909) for k in range(3)])
    got = {}
    monkeypatch.setenv("OG_WITNESS_W9", "0")   # (the witness in the lane-local kernels this case is about; the wave-wide form has its own cases)
    for form in ("0", "1", "w9", "w9_one_row", "w9_two_rows_29_bit_digit"):
        if form.startswith("w9"):   # a wave per hash (mimc7.hip.h w9_mimc7_hash2): what launches of at most two waves per SIMD take since round 6
            monkeypatch.setenv("OG_MIMC_W9", "1")
            # rounds three products deep over two rows of the wave with the 32-bit Montgomery digit (2, the default), the same with
            # the 29-bit digit (1), or the four-deep single-row form before them (0): mimc7.hip.h w9_mimc7_round<FORM>
            monkeypatch.setenv("OG_W9_ROWS", {"w9": "2", "w9_one_row": "0"}.get(form, "1"))
        else:
            monkeypatch.setenv("OG_MIMC_W9", "0")
            monkeypatch.setenv("OG_MIMC_PAIR", form)
        h = ctx.to_host(ctx.mimc7_hash2(ctx.to_device(_tob(l)), ctx.to_device(_tob(r))))
        p = ctx.to_host(ctx.mimc7_merkle_paths(ctx.to_device(_tob(leaves)), ctx.to_device(idx),
                                               ctx.to_device(np.stack([_tob(s) for s in sib])), depth))
        t = ctx.to_host(ctx.mimc7_tree_build(ctx.to_device(_tob(tree_leaves))))
        w = ctx.to_host(circuit.witness(ctx, witness_depth, ctx.to_device(recs), 2, 3))
        got[form] = tuple(np.asarray(x).tobytes() for x in (h, p, t, w))
    assert got["0"] == got["1"] == got["w9"] == got["w9_one_row"] == got["w9_two_rows_29_bit_digit"]
    assert _toi(np.frombuffer(got["1"][0], dtype=np.uint8)) == [mimc7.hash2(x, y) for x, y in zip(l, r)]
    paths = np.frombuffer(got["1"][1], dtype=np.uint8).reshape(n_paths, depth + 1, 32)
    for i in range(n_paths):
        assert _toi(paths[i]) == mimc7.merkle_root_from_path(leaves[i], int(idx[i]), sib[i])
    assert _toi(np.frombuffer(got["1"][2], dtype=np.uint8)) == [x for lvl in mimc7.tree_build(tree_leaves) for x in lvl]
