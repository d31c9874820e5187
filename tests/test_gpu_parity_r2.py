"""GPU parity, round 2 additions (VERDICT r1 "weak" 1-3): a large G2 MSM with heavy buckets, a 2^22 G1 MSM whose known
answer is computed on the HOST, the 2^20-leaf MiMC7 tree against the C restatement node by node, named EIP-196/197
algebraic known answers run against the PRODUCT's MSM and verifier, and a full-size (2^18-wire) byte-parity case."""
import random

import numpy as np
import pytest

from oracle.py import fields
from oracle.py.curve import G1, G2, G1_GEN, G2_GEN, g1_to_bytes, g2_to_bytes

pytestmark = pytest.mark.gpu


def _rand_fr(rng, *shape, top=0x0F):
    a = rng.integers(0, 256, (*shape, 32), dtype=np.uint8)
    a[..., 31] &= top
    return a


def _dot_mod_r(a, s):
    from bench import host_dot_mod_r
    return host_dot_mod_r(a, s)


@pytest.mark.parametrize("log_n,window,precomp", [(16, 16, True), (17, 16, True), (14, 12, False)])
def test_msm_g2_known_answer_with_heavy_buckets(ctx, log_n, window, precomp):
    """SURVEY 8c(ii) in G2: bases P_i = a_i G2 (generated on the GPU, a sample checked against the C restatement) =>
    MSM(s, P) == (sum a_i s_i mod r) G2 with the dot product AND the expected point computed on the host.  ~10 % of the
    scalars are 0 / 1 (boolean wires): the 1-bucket goes through k_accumulate_heavy<Fq2>."""
    from owshen_amd import api, groth16
    from oracle.c import binding as oc
    n = 1 << log_n
    rng = np.random.default_rng(200 + log_n)
    a, s = _rand_fr(rng, n), _rand_fr(rng, n)
    s[::10] = 0
    s[5::10] = 0
    s[5::10, 0] = 1
    s[7] = np.frombuffer((fields.R - 1).to_bytes(32, "little"), dtype=np.uint8)
    pts = ctx.scalar_mul(2, groth16.G2_GEN_BYTES, ctx.to_device(a))
    idx = rng.choice(n, 64, replace=False)
    gen2 = np.frombuffer(g2_to_bytes(G2_GEN), dtype=np.uint8)
    assert ctx.to_host(pts)[idx].tobytes() == oc.fixed_base_g2(gen2, a[idx]).tobytes()
    want = g2_to_bytes(G2.mul(G2_GEN, _dot_mod_r(a, s)))
    bases = api.Bases(ctx, 2, pts, window, precomp)
    assert bases.msm(ctx.to_device(s))[0].tobytes() == want
    bases.close()


def test_msm_g1_2_22_known_answer_host_side(ctx):
    """2^22-point G1 MSM (the configs[2] shape at 1/16 size): the scalar sum a_i s_i and the expected point come from
    the host (numpy / Python integers / the Python curve oracle), a sample of the GPU-generated bases is checked against
    the C restatement -- no leg of the known answer is produced by the library under test."""
    from owshen_amd import api, groth16
    from oracle.c import binding as oc
    n = 1 << 22
    rng = np.random.default_rng(22)
    a, s = _rand_fr(rng, n), _rand_fr(rng, n)
    s[::16] = 0
    s[1::16] = 0
    s[1::16, 0] = 1
    pts = ctx.scalar_mul(1, groth16.G1_GEN_BYTES, ctx.to_device(a))
    idx = rng.choice(n, 4096, replace=False)
    gen = np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8)
    import torch
    assert ctx.to_host(pts[torch.from_numpy(idx).to(pts.device)]).tobytes() == oc.fixed_base_g1(gen, a[idx]).tobytes()
    want = g1_to_bytes(G1.mul(G1_GEN, _dot_mod_r(a, s)))
    s_d = ctx.to_device(s)
    for precomp in (True, False):
        bases = api.Bases(ctx, 1, pts, 16, precomp)
        assert bases.msm(s_d)[0].tobytes() == want, precomp
        # window-sharded form (every rank played in turn on this GPU): same bytes
        world = 4
        parts = [bases.msm_windows(s_d, r, world) for r in range(world)]
        assert bases.msm_combine(torch.cat(parts), world).tobytes() == want
        bases.close()


def test_mimc7_tree_2_20_vs_c_oracle(ctx):
    """BASELINE.json configs[4] at full size on one GPU: every one of the 2^21 - 1 nodes equals the C restatement's"""
    from oracle.c import binding as oc
    n = 1 << 20
    rng = np.random.default_rng(2)
    leaves = _rand_fr(rng, n, top=0x1F)
    got = ctx.to_host(ctx.mimc7_tree_build(ctx.to_device(leaves)))
    want = oc.mimc7_tree_build(leaves)
    assert got.shape == want.shape == (2 * n - 1, 32)
    assert got.tobytes() == want.tobytes()


# ---- named algebraic known answers (EIP-196 / EIP-197), against the product --------------------------------------

TWO_G1 = (1368015179489954701390400359078579693043519447331113978918064868415326638035,
          9918110051302171585080402603319702774565515993150576347155970296011118125764)   # 2 * (1, 2) on alt_bn128 (EIP-196)


def test_eip196_named_constants_against_msm(ctx):
    from owshen_amd import api
    g = np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8)
    g2 = np.frombuffer(g2_to_bytes(G2_GEN), dtype=np.uint8)
    b1 = api.Bases(ctx, 1, ctx.to_device(np.stack([g, g])), 8, False)
    b2 = api.Bases(ctx, 2, ctx.to_device(np.stack([g2, g2])), 8, False)
    sc = lambda *v: ctx.to_device(api.ints_to_bytes(list(v)))  # noqa: E731
    # 2 G1 = the EIP-196 doubling of the generator, reached as G + G (the in-bucket doubling) and as 2 * G
    assert b1.msm(sc(1, 1))[0].tobytes() == g1_to_bytes(TWO_G1) == b1.msm(sc(2, 0))[0].tobytes()
    # r G = O in both groups: (r - 1) G + G
    assert not b1.msm(sc(fields.R - 1, 1)).any()
    assert not b2.msm(sc(fields.R - 1, 1)).any()
    # (r - 1) G = -G
    assert b1.msm(sc(fields.R - 1, 0))[0].tobytes() == g1_to_bytes((1, fields.P - 2))
    # BN parameter identities (EIP-197): p and r from x = 4965661367192848881
    x = 4965661367192848881
    assert fields.P == 36 * x**4 + 36 * x**3 + 24 * x**2 + 6 * x + 1 and fields.R == 36 * x**4 + 36 * x**3 + 18 * x**2 + 6 * x + 1


# ---- full-size byte parity (BASELINE.json configs[1] shape) inside pytest ---------------------------------------

@pytest.mark.parametrize("dense", [False, True])
def test_full_size_withdraw_proofs_byte_identical_to_c_oracle(ctx, dense):
    """2^18 wires / 2^17 domain, depth-32 Merkle path: GPU proofs (witness generated on the GPU, fused entry point) are
    byte-identical to the C restatement's; the product's verifier accepts them"""
    from owshen_amd import circuit, groth16 as g16
    from oracle.c import binding as oc
    depth = 32
    n_pad3, n_pad2 = circuit.baseline_shape(depth, dense=dense)
    r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2, dense=dense)
    assert (r1.n_wires, r1.domain_size) == (1 << 18, 1 << 17)
    blob, vk = g16.setup(ctx, r1, 0x1234567, 0x2345678, 0x3456789, 0x456789A, 0x56789AB)
    pk = g16.ProvingKey(ctx, blob)
    dn = pk.density()
    if dense:
        assert dn["a"] == dn["b"] == 1 << 18
    rng = np.random.default_rng(18 + dense)
    n = 3
    recs = _rand_fr(rng, n, 8 + depth, top=0x1F)
    recs[:, 5, 8:] = 0
    recs[:, 5, :8] = (recs[:, 5, :8].view(np.uint64) & np.uint64((1 << depth) - 1)).view(np.uint8)
    rs = _rand_fr(rng, n, 2, top=0x1F).reshape(n, 64)
    recs_d = ctx.to_device(recs)
    proofs = circuit.prove_from_inputs(ctx, pk, depth, recs_d, rs, n_pad3, n_pad2)
    wit = ctx.to_host(circuit.witness(ctx, depth, recs_d, n_pad3, n_pad2))
    ck = oc.prepared_key_from_blob(blob)
    vkb = g16.vk_to_bytes(vk)
    for k in range(n if not dense else 1):
        r, s = int.from_bytes(rs[k, :32].tobytes(), "little"), int.from_bytes(rs[k, 32:].tobytes(), "little")
        assert proofs[k].tobytes() == ck.prove(wit[k], r, s)
    for k in range(n):
        assert g16.verify(vkb, wit[k][1:7], proofs[k].tobytes())
    assert not g16.verify(vkb, wit[1][1:7], proofs[0].tobytes())
    pk.close()
