"""The deposit statement on the GPU through the C ABI; cases in tests/deposit_cases.py."""
import numpy as np
import pytest

from tests import deposit_cases as cases

pytestmark = pytest.mark.gpu


def test_deposit_r1cs_and_witness_match_spec(ctx):
    cases.case_r1cs_and_witness_match_spec(ctx, n=70)


def test_deposit_end_to_end(ctx):
    cases.case_deposit_end_to_end(ctx, n=6)


def test_deposit_batch_4096_verifies_and_matches_the_c_restatement(ctx):
    """a throughput-shaped call: 4096 deposits in one og_deposit_prove_batch_d (sub-batches of up to 1024 through the stage
    pipeline), every proof accepted by og_verify with (commitment, depositor), the first / last of every 512 byte-identical to the
    C restatement"""
    from concurrent.futures import ThreadPoolExecutor
    from oracle.c import binding as oc
    from owshen_amd import circuit, groth16 as g16
    n = 4096
    rng = np.random.default_rng(4096)
    recs = rng.integers(0, 256, (n, 3, 32), dtype=np.uint8)
    recs[:, :, 31] &= 0x1F
    recs[:, 2, 20:] = 0
    rs = rng.integers(0, 256, (n, 64), dtype=np.uint8)
    rs[:, 31] &= 0x1F
    rs[:, 63] &= 0x1F
    r1 = circuit.deposit_r1cs_native(ctx)
    blob, vk = g16.setup(ctx, r1, 51, 52, 53, 54, 55)
    pk = g16.ProvingKey(ctx, blob)
    recs_d = ctx.to_device(recs)
    proofs, pub = circuit.deposit_prove(ctx, pk, recs_d, rs, return_public=True)
    assert pub[:, 1].tobytes() == recs[:, 2].tobytes()
    vkb = g16.vk_to_bytes(vk)
    with ThreadPoolExecutor(32) as ex:
        ok = list(ex.map(lambda i: g16.verify(vkb, pub[i], proofs[i].tobytes()), range(n)))
    assert all(ok), f"{ok.count(False)} of {n} deposit proofs refused"
    assert g16.verify(vkb, pub[1], proofs[0].tobytes()) is False
    idx = sorted({i for b in range(0, n, 512) for i in (b, b + 511)})
    wit = ctx.to_host(circuit.deposit_witness(ctx, recs_d[idx]))
    ck = oc.prepared_key_from_blob(blob)
    for j, t in enumerate(idx):
        r_, s_ = int.from_bytes(rs[t][:32].tobytes(), "little"), int.from_bytes(rs[t][32:].tobytes(), "little")
        assert proofs[t].tobytes() == ck.prove(wit[j], r_, s_), t
    pk.close()
