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
