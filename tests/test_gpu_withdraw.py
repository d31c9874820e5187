"""GPU parity: withdraw circuit, witness generator and end-to-end proofs (SURVEY 8a-N5/N6, configs[0]-style
plumbing at depth 32); cases in tests/withdraw_cases.py."""
import pytest

from tests import withdraw_cases as cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth,n_pad3,n_pad2", [(1, 0, 0), (3, 7, 130), (32, 0, 0), (32, 100, 1000)])
def test_r1cs_and_witness_match_spec(ctx, depth, n_pad3, n_pad2):
    cases.case_r1cs_and_witness_match_spec(ctx, depth, n_pad3, n_pad2)


@pytest.mark.parametrize("depth,n_pad3,n_pad2", [(1, 5, 70), (32, 0, 0)])
def test_withdraw_end_to_end(ctx, depth, n_pad3, n_pad2):
    cases.case_withdraw_end_to_end(ctx, depth, n_pad3, n_pad2)


@pytest.mark.parametrize("depth,n_pad3,n_pad2,n", [(3, 7, 130, 3), (32, 0, 0, 5), (32, 100, 1000, 2)])
def test_host_chains_gives_the_kernels_bytes(ctx, depth, n_pad3, n_pad2, n):
    """og_set_host_chains: the chains of a handful of requests on the host CPU -- witnesses, proofs and public inputs are the kernels'"""
    cases.case_host_chains_gives_the_kernels_bytes(ctx, depth, n_pad3, n_pad2, n_proofs=n)


def test_field_mulchain_wave_wide_form(ctx):
    """the wave-wide ("w9") Montgomery product (csrc/field_w9.hip.h) chained on the GPU: the bytes of the lane-local chain, both fields"""
    import torch
    for field in (0, 1):
        g = torch.Generator().manual_seed(90 + field)
        x = torch.randint(0, 256, (70, 32), dtype=torch.uint8, generator=g)
        x[:, 31] &= 0x1F
        y = x.flip(0).contiguous().cuda()
        ref = x.cuda()
        ctx.field_mulchain(field, ref, y, 33)
        for form in (0, 1):
            xd = x.cuda()
            _ms, cyc = ctx.field_mulchain_lat(field, form, xd, y, 33)
            assert torch.equal(xd, ref) and cyc > 0, (field, form)


def test_dense_rows(ctx):
    cases.case_dense_rows(ctx, 2, 9, 200)


def test_withdraw_end_to_end_dense(ctx):
    cases.case_withdraw_end_to_end(ctx, 1, 5, 70, dense=True)


@pytest.mark.parametrize("depth,n_pad3,n_pad2,dense", [(32, 0, 0, False), (4, 100, 1000, True)])
def test_native_builder(ctx, depth, n_pad3, n_pad2, dense):
    cases.case_native_builder_equals_python_builder(ctx, depth, n_pad3, n_pad2, dense)


def test_submitted_batches_overlap_and_equal_blocking_calls(ctx):
    """og_withdraw_prove_batch_submit_d on the full-size (2^18-wire) circuit, where calls really stay enqueued on the streams:
    three batches kept one ahead of their waits give the bytes of the blocking call; a third in-flight call is refused"""
    from owshen_amd import circuit
    n_pad3, n_pad2 = circuit.baseline_shape(32)
    cases.case_submitted_batches_equal_blocking_calls(ctx, 32, n_pad3, n_pad2, [300, 130, 260], third_is_refused=True)


def test_malformed_records_are_rejected(ctx):
    cases.case_malformed_records_are_rejected(ctx, 4, 5, 70)


def test_malformed_records_are_rejected_inside_the_pipeline(ctx):
    """the full-size (2^18-wire) circuit, where witnesses are generated inside the stage pipeline and the records are checked per
    sub-batch beside the accumulations"""
    from owshen_amd import circuit
    n_pad3, n_pad2 = circuit.baseline_shape(32)
    cases.case_malformed_records_are_rejected(ctx, 32, n_pad3, n_pad2, pipeline=True)


def test_jobs_are_consumed_once(ctx):
    from owshen_amd import circuit
    cases.case_jobs_are_consumed_once(ctx, 2, 5, 70)
    n_pad3, n_pad2 = circuit.baseline_shape(32)
    cases.case_jobs_are_consumed_once(ctx, 32, n_pad3, n_pad2, stays_enqueued=True, n_proofs=256)


def test_wave_per_proof_witness_equals_the_two_lane_form(ctx_hooks, monkeypatch):
    """k_withdraw_core_lat (a wave per proof: the independent permutations side by side, up to 16 requests) against
    k_withdraw_core<true> (two lanes per proof) on 16 depth-32 records whose leaf indices cover all-left, all-right, alternating
    and random paths: the same wires byte for byte, and equal to the spec for the first record"""
    import random
    import numpy as np
    from owshen_amd import api, circuit
    from oracle.py import fields, withdraw as spec
    ctx = ctx_hooks  # (OG_WITNESS_LAT forces either form: hooks build)
    rnd = random.Random(16)
    depth = 32
    idx = [0, (1 << depth) - 1, 0x55555555, 0xAAAAAAAA, 1, 1 << 31] + [rnd.randrange(1 << depth) for _ in range(10)]
    ins = [dict(nullifier=rnd.randrange(fields.R), secret=rnd.randrange(fields.R), amount=rnd.randrange(1 << 64), recipient=rnd.randrange(1 << 160),
                pad_seed=rnd.randrange(fields.R), index=i, siblings=[rnd.randrange(fields.R) for _ in range(depth)],
                token=rnd.randrange(1 << 160), chain_id=1387) for i in idx]
    packed = ctx.to_device(np.stack([circuit.pack_inputs(**i) for i in ins]))
    w9 = ctx.to_host(circuit.witness(ctx, depth, packed))      # the default for a call this size: the wave-wide form (k_w9_*, round 6)
    for form in ("0", "1"):   # ... its rounds four products deep in one row of the wave (as first built) / three deep over two rows, 29-bit digit
        monkeypatch.setenv("OG_W9_ROWS", form)
        assert ctx.to_host(circuit.witness(ctx, depth, packed)).tobytes() == w9.tobytes(), form
    monkeypatch.delenv("OG_W9_ROWS")
    monkeypatch.setenv("OG_WITNESS_W9", "0")
    monkeypatch.setenv("OG_WITNESS_LAT", "1")
    lat = ctx.to_host(circuit.witness(ctx, depth, packed))
    monkeypatch.setenv("OG_WITNESS_LAT", "0")
    two = ctx.to_host(circuit.witness(ctx, depth, packed))
    assert lat.tobytes() == two.tobytes() == w9.tobytes()
    for k in (0, 1, 7):
        i = ins[k]
        z = spec.build(depth, i["nullifier"], i["secret"], i["amount"], i["recipient"], i["index"], i["siblings"], i["pad_seed"], 0, 0,
                       token=i["token"], chain_id=i["chain_id"])[3]
        assert api.bytes_to_ints(lat[k]) == z


def test_second_engine_verifies_a_depth_32_gpu_proof(ctx):
    """a depth-32 withdraw proof made on the GPU (inputs -> witness -> proof inside the library), checked by the second,
    independent verifier (oracle/js/bn254_pairing_second.js: V8 BigInt, flat Fp12, untwisted Miller loop) with the public
    inputs the call handed back; another recipient or a replay on another chain is refused"""
    import random
    import numpy as np
    from owshen_amd import circuit
    from oracle.py import fields
    from tests import test_second_engine as se
    if se.NODE is None:
        pytest.skip("node is not installed")
    rnd = random.Random(77)
    depth = 32
    _r1, _blob, vk, pk, close = cases._key(ctx, depth, 0, 0)
    packed = np.stack([cases._pack(circuit, cases._inputs(rnd, depth))])
    proofs, pubs = circuit.prove_from_inputs(ctx, pk, depth, ctx.to_device(packed), [(rnd.randrange(fields.R), rnd.randrange(fields.R))],
                                             0, 0, return_public=True)
    close()
    pub = [str(int.from_bytes(pubs[0][i].tobytes(), "little")) for i in range(6)]
    variants = [pub]
    for slot in (2, 5):                      # someone else's recipient; the same proof on another chain
        v = list(pub)
        v[slot] = str((int(v[slot]) + 1) % fields.R)
        variants.append(v)
    res = se._run_pairing({"vk": se._vk_json(vk), "proofs": [{"public": v, "proof": se._proof_json(proofs[0].tobytes())} for v in variants]})
    assert res["proofs"] == [True, False, False]
