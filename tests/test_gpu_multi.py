"""og_multi_* on real hardware.  A gpurun box has ONE GPU, so this drives the single-process multi-device layer with
one rank -- with OG_MULTI_RCCL=1 the rank still goes through ncclCommInitAll / ncclBroadcast / ncclAllGather (real RCCL
kernels on the ctx stream), which is the part the CPU interpreter cannot cover (tests/test_emu_multi.py covers the
slicing and the N-rank exchange logic)."""
import os

import numpy as np
import pytest

from oracle.py.curve import G1_GEN, g1_to_bytes

pytestmark = pytest.mark.gpu


def _rand_fr(rng, *shape):
    a = rng.integers(0, 256, (*shape, 32), dtype=np.uint8)
    a[..., 31] &= 0x1F
    return a


@pytest.fixture(scope="module")
def multi1():
    """one device, but routed through RCCL all the same (OG_MULTI_RCCL: a switch of the hooks build)"""
    from tests.conftest import hooks_lib
    os.environ["OG_MULTI_RCCL"] = "1"
    from owshen_amd import multi
    m = multi.Multi(1, lib=hooks_lib())
    yield m
    m.close()
    os.environ.pop("OG_MULTI_RCCL", None)


def test_multi_init_refuses_missing_devices():
    from owshen_amd import multi, api
    import torch
    with pytest.raises(api.OwshenGpuError, match="devices"):
        multi.Multi(torch.cuda.device_count() + 1)


@pytest.mark.parametrize("window,precomp", [(16, True), (12, False)])
def test_multi_msm_through_rccl(multi1, window, precomp):
    from oracle.c import binding as oc
    rng = np.random.default_rng(window)
    n = 20000
    bases_np = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), _rand_fr(rng, n))
    sc = _rand_fr(rng, n)
    sc[::7] = 0
    b = multi1.bases(1, bases_np, window, precomp)
    assert multi1.msm(b, sc).tobytes() == oc.msm_g1(bases_np, sc).tobytes()
    multi1.free_bases(b)


def test_multi_withdraw_prove_batch_equals_single_device(ctx, multi1):
    from owshen_amd import circuit, groth16 as g16
    depth, n_pad3, n_pad2 = 4, 50, 700
    r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2)
    blob, _vk = g16.setup(ctx, r1, 31, 32, 33, 34, 35)
    rng = np.random.default_rng(9)
    n = 5
    recs = _rand_fr(rng, n, 8 + depth)
    recs[:, 5, 8:] = 0
    recs[:, 5, :8] = (recs[:, 5, :8].view(np.uint64) & np.uint64((1 << depth) - 1)).view(np.uint8)
    rs = _rand_fr(rng, n, 2).reshape(n, 64)
    pk = g16.ProvingKey(ctx, blob)
    want = circuit.prove_from_inputs(ctx, pk, depth, ctx.to_device(recs), rs, n_pad3, n_pad2)
    pks = multi1.load_key(blob)
    assert multi1.withdraw_prove_batch(pks, depth, recs, rs, n_pad3, n_pad2).tobytes() == want.tobytes()
    wit = ctx.to_host(circuit.witness(ctx, depth, ctx.to_device(recs), n_pad3, n_pad2))
    assert multi1.prove_batch(pks, wit, rs).tobytes() == want.tobytes()
    multi1.free_key(pks)
    pk.close()


# ---- window-sharded PROVING (round 6) ----------------------------------------------------------------------------------------
def _depth4_case(ctx, n, seed=9):
    from owshen_amd import circuit, groth16 as g16
    depth, n_pad3, n_pad2 = 4, 50, 700
    r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2)
    blob, _vk = g16.setup(ctx, r1, 31, 32, 33, 34, 35)
    rng = np.random.default_rng(seed)
    recs = _rand_fr(rng, n, 8 + depth)
    recs[:, 5, 8:] = 0
    recs[:, 5, :8] = (recs[:, 5, :8].view(np.uint64) & np.uint64((1 << depth) - 1)).view(np.uint8)
    rs = _rand_fr(rng, n, 2).reshape(n, 64)
    return depth, n_pad3, n_pad2, blob, recs, rs


@pytest.mark.parametrize("n", [1, 5, 40])
def test_multi_prove_sharded_through_rccl_equals_single_device(ctx, multi1, n):
    """og_multi_withdraw_prove_sharded / og_multi_prove_sharded with ONE rank routed through real RCCL (ncclAllGather on the
    ctx's stream, ordered behind the front half's five streams by events): the bytes of og_withdraw_prove_batch_d, the public
    inputs, for one request (fanned out), a handful and a sub-batched call"""
    from owshen_amd import circuit, groth16 as g16
    depth, n_pad3, n_pad2, blob, recs, rs = _depth4_case(ctx, n)
    pk = g16.ProvingKey(ctx, blob)
    want, want_pub = circuit.prove_from_inputs(ctx, pk, depth, ctx.to_device(recs), rs, n_pad3, n_pad2, return_public=True)
    pks = multi1.load_key(blob)
    got, pub = multi1.withdraw_prove_sharded(pks, depth, recs, rs, n_pad3, n_pad2, return_public=True)
    assert got.tobytes() == want.tobytes() and pub.tobytes() == want_pub.tobytes()
    wit = ctx.to_host(circuit.witness(ctx, depth, ctx.to_device(recs), n_pad3, n_pad2))
    assert multi1.prove_sharded(pks, wit, rs).tobytes() == want.tobytes()
    multi1.free_key(pks)
    pk.close()


@pytest.mark.parametrize("world,n", [(2, 3), (8, 1), (8, 20), (15, 2)])
def test_window_shards_of_one_gpu_add_up(ctx, world, n):
    """The N-rank arithmetic on ONE GPU, no collective: the front half run `world` times on the same context, once per window
    owner (og_withdraw_prove_partials_d with win_rank = 0 .. world - 1: this small key's 22 twelve-bit windows over 2, 8 and 15
    owners; the 2^18-wire key with its 15 seventeen-bit windows: tests/test_gpu_fullsize_pipeline.py), the blocks concatenated the way an all-gather leaves them, og_prove_from_partials_d: the proofs of the unsharded
    call, byte for byte -- on the gfx950 kernels (restricted digit sorts, partial bucket sets, k_sum_ranks, the assembly)"""
    import torch
    from owshen_amd import circuit, groth16 as g16
    depth, n_pad3, n_pad2, blob, recs, rs = _depth4_case(ctx, n, seed=world)
    pk = g16.ProvingKey(ctx, blob)
    inputs_d = ctx.to_device(recs)
    want, want_pub = circuit.prove_from_inputs(ctx, pk, depth, inputs_d, rs, n_pad3, n_pad2, return_public=True)
    parts = []
    for r in range(world):
        part, pub = circuit.partials_from_inputs(ctx, pk, depth, inputs_d, r, world, n_pad3, n_pad2, return_public=True)
        assert pub.tobytes() == want_pub.tobytes()
        parts.append(part.clone())
    got = pk.prove_from_partials(torch.cat(parts), world, rs)
    assert got.tobytes() == want.tobytes()
    # caller-supplied witnesses take the same route
    wit_d = circuit.witness(ctx, depth, inputs_d, n_pad3, n_pad2)
    parts = [pk.prove_partials_device(wit_d, r, world).clone() for r in range(world)]
    assert pk.prove_from_partials(torch.cat(parts), world, rs).tobytes() == want.tobytes()
    pk.close()


def test_window_sharded_front_refuses_what_the_unsharded_call_refuses(ctx):
    from owshen_amd import api, circuit, groth16 as g16
    from oracle.py import fields
    depth, n_pad3, n_pad2, blob, recs, rs = _depth4_case(ctx, 4)
    pk = g16.ProvingKey(ctx, blob)
    bad = recs.copy()
    bad[2, 1] = np.frombuffer(fields.R.to_bytes(32, "little"), dtype=np.uint8)
    with pytest.raises(api.OwshenGpuError, match="input record 2: field 1"):
        circuit.partials_from_inputs(ctx, pk, depth, ctx.to_device(bad), 1, 2, n_pad3, n_pad2)
    with pytest.raises(api.OwshenGpuError, match="window shard"):
        circuit.partials_from_inputs(ctx, pk, depth, ctx.to_device(recs), 2, 2, n_pad3, n_pad2)
    wit = circuit.witness(ctx, depth, ctx.to_device(recs), n_pad3, n_pad2)
    wit[3, -1, 0] ^= 1
    with pytest.raises(api.OwshenGpuError, match="witness 3 does not satisfy"):
        pk.prove_partials_device(wit, 0, 2)
    # the context is usable afterwards: no call slot left occupied
    want = circuit.prove_from_inputs(ctx, pk, depth, ctx.to_device(recs), rs, n_pad3, n_pad2)
    part = circuit.partials_from_inputs(ctx, pk, depth, ctx.to_device(recs), 0, 1, n_pad3, n_pad2)
    assert pk.prove_from_partials(part, 1, rs).tobytes() == want.tobytes()
    pk.close()
