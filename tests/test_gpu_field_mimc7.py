"""GPU parity: Montgomery field arithmetic and MiMC7 (8a-N5) through the C ABI,
bit-exact against the Python big-int oracle on seeded inputs."""
import random

import numpy as np
import pytest
import torch

from oracle.py import fields, mimc7

pytestmark = pytest.mark.gpu


def _rand_elems(rnd, mod, n):
    edge = [0, 1, 2, mod - 1, mod - 2, (1 << 253), (1 << 253) - 1, mod >> 1, 0xFFFFFFFF, 1 << 32,
            (1 << 64) - 1, 1 << 64, (1 << 224) - 1]
    v = edge + [rnd.randrange(mod) for _ in range(n - len(edge))]
    return v


@pytest.mark.parametrize("field,mod", [(0, fields.R), (1, fields.P)])
def test_field_ops_bit_exact(ctx, field, mod):
    from owshen_amd import api
    rnd = random.Random(1000 + field)
    n = 1024
    a = _rand_elems(rnd, mod, n)
    b = list(reversed(_rand_elems(rnd, mod, n)))
    ad, bd = ctx.to_device(api.ints_to_bytes(a)), ctx.to_device(api.ints_to_bytes(b))
    for op, fn in (("add", lambda x, y: (x + y) % mod), ("sub", lambda x, y: (x - y) % mod),
                   ("mul", lambda x, y: x * y % mod)):
        got = api.bytes_to_ints(ctx.field_op(field, op, ad, bd).cpu().numpy())
        assert got == [fn(x, y) for x, y in zip(a, b)], op
    got = api.bytes_to_ints(ctx.field_op(field, "inv", ad).cpu().numpy())
    assert got == [pow(x, -1, mod) if x else 0 for x in a]


def test_field_empty_batch(ctx):
    e = ctx.empty(0, 32)
    assert ctx.field_op(0, "mul", e, e).shape[0] == 0


def test_mimc7_constants_match_oracle(ctx):
    assert ctx.mimc7_constants() == mimc7.CONSTANTS


def test_mimc7_hash2_bit_exact(ctx):
    from owshen_amd import api
    rnd = random.Random(7)
    n = 300  # ragged: not a multiple of the 256-lane block
    l = [0, 1, fields.R - 1] + [rnd.randrange(fields.R) for _ in range(n - 3)]
    r = [0, 2, fields.R - 1] + [rnd.randrange(fields.R) for _ in range(n - 3)]
    got = api.bytes_to_ints(ctx.mimc7_hash2(ctx.to_device(api.ints_to_bytes(l)),
                                            ctx.to_device(api.ints_to_bytes(r))).cpu().numpy())
    assert got == [mimc7.hash2(x, y) for x, y in zip(l, r)]
    assert got[1] == mimc7.multi_hash([1, 2], 0)


def test_mimc7_merkle_paths_depth32(ctx):
    from owshen_amd import api
    rnd = random.Random(8)
    n, depth = 5, 32
    leaves = [rnd.randrange(fields.R) for _ in range(n)]
    idx = [0, 0x2A, (1 << 32) - 1, rnd.randrange(1 << 32), rnd.randrange(1 << 32)]
    sib = [[rnd.randrange(fields.R) for _ in range(depth)] for _ in range(n)]
    sib_b = np.stack([api.ints_to_bytes(s) for s in sib])
    out = ctx.mimc7_merkle_paths(ctx.to_device(api.ints_to_bytes(leaves)),
                                 torch.tensor(idx, dtype=torch.int64, device=ctx.device),
                                 ctx.to_device(sib_b), depth).cpu().numpy()
    for i in range(n):
        assert api.bytes_to_ints(out[i]) == mimc7.merkle_root_from_path(leaves[i], idx[i], sib[i])


def test_mimc7_tree_build_and_path_consistency(ctx):
    from owshen_amd import api
    rnd = random.Random(9)
    n = 64
    leaves = [rnd.randrange(fields.R) for _ in range(n)]
    nodes = api.bytes_to_ints(ctx.mimc7_tree_build(ctx.to_device(api.ints_to_bytes(leaves))).cpu().numpy())
    levels = mimc7.tree_build(leaves)
    flat = [x for lvl in levels for x in lvl]
    assert nodes == flat
    # single-leaf tree: root = leaf
    one = api.bytes_to_ints(ctx.mimc7_tree_build(ctx.to_device(api.ints_to_bytes([5]))).cpu().numpy())
    assert one == [5]


def test_mimc7_tree_root_equals_path_root_large(ctx):
    """size-independent property at 2^14 leaves: root from the tree == root recomputed from a
    leaf's path through og_mimc7_merkle_paths_d (8c-v)."""
    from owshen_amd import api
    k = 14
    n = 1 << k
    g = torch.Generator(device="cpu").manual_seed(2)
    raw = torch.randint(0, 256, (n, 32), dtype=torch.uint8, generator=g)
    raw[:, 31] &= 0x1F  # < 2^253 < r : canonical
    leaves = raw.to(ctx.device)
    nodes = ctx.mimc7_tree_build(leaves)
    root = nodes[-1].cpu().numpy().tobytes()
    for leaf_idx in (0, 0x2A, n - 1):
        sib, off, i = [], 0, leaf_idx
        for lvl in range(k):
            sib.append(nodes[off + (i ^ 1)])
            off += n >> lvl
            i >>= 1
        sibs = torch.stack(sib).unsqueeze(0).contiguous()
        out = ctx.mimc7_merkle_paths(leaves[leaf_idx:leaf_idx + 1].contiguous(),
                                     torch.tensor([leaf_idx], dtype=torch.int64, device=ctx.device), sibs, k)
        assert out[0, -1].cpu().numpy().tobytes() == root


def test_mimc7_merkle_paths_depth32_batch_1024_vs_c_oracle(ctx):
    """the witness generator's shape -- 1024 depth-32 paths in one og_mimc7_merkle_paths_d call -- against the C restatement's
    two-to-one hash, level by level"""
    from oracle.c import binding as oc
    n, depth = 1024, 32
    rng = np.random.default_rng(1024)
    leaves = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    leaves[:, 31] &= 0x1F
    sib = rng.integers(0, 256, (n, depth, 32), dtype=np.uint8)
    sib[:, :, 31] &= 0x1F
    idx = rng.integers(0, 1 << 32, n, dtype=np.int64)
    idx[:3] = (0, 0x2A, (1 << 32) - 1)
    out = ctx.mimc7_merkle_paths(ctx.to_device(leaves), torch.from_numpy(idx).to(ctx.device), ctx.to_device(sib), depth).cpu().numpy()
    assert out.shape == (n, depth + 1, 32) and out[:, 0].tobytes() == leaves.tobytes()
    node = leaves
    for lvl in range(depth):
        right = ((idx >> lvl) & 1).astype(bool)[:, None]          # the path node is the RIGHT input where the index bit is set
        l = np.where(right, sib[:, lvl], node)
        r = np.where(right, node, sib[:, lvl])
        node = oc.mimc7_hash2(np.ascontiguousarray(l), np.ascontiguousarray(r))
        assert out[:, lvl + 1].tobytes() == node.tobytes(), lvl


def test_tree_build_root_equals_append_root(ctx):
    """f-3 <-> C5 consistency: the dense builder (og_mimc7_tree_build_d) over 2^12 leaves and the incremental tree
    (og_mimc7_append_d, depth 12, the same leaves in three uneven batches) arrive at the same root"""
    k = 12
    n = 1 << k
    rng = np.random.default_rng(12)
    leaves = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    leaves[:, 31] &= 0x1F
    ld = ctx.to_device(leaves)
    dense_root = ctx.to_host(ctx.mimc7_tree_build(ld)[-1:]).reshape(32).tobytes()
    frontier = ctx.to_device(np.zeros((k, 32), dtype=np.uint8))
    at = 0
    for cnt in (1, 1000, n - 1001):
        frontier, root = ctx.mimc7_append(k, frontier, at, ld[at:at + cnt].contiguous())
        at += cnt
    assert at == n and ctx.to_host(root).reshape(32).tobytes() == dense_root
