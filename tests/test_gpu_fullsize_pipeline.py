"""Full-size parity of the STAGE PIPELINE (VERDICT r4 "weak" 2): the headline's own call -- 1024 proofs of the 2^18-wire /
2^17-domain circuit, plan [64, 240, 240, 240, 240] over two rotating scratch slots (three until round 6) and four streams -- with every
sub-batch in front of something that is not the pipeline:

  * the strictly serial path (og_set_lanes(1): one stream, one slot) must produce the same 1024 x 256 bytes;
  * the C restatement re-proves the first and the last proof of EVERY sub-batch (sub-batch k runs in slot k mod 2, so the
    sample straddles each slot's reuse -- k and k + 2, released in two steps: groth16.hip pipe_slots) and must produce the same bytes;
  * og_verify accepts all 1024 proofs with the public inputs the call returned, and refuses a proof for its neighbour's.

The reference's convention for the seam: a burn is accepted by `verify` or refused
(/root/reference/src/blockchain/tx/burn_tx.rs:11-32)."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand_fr(rng, *shape, top=0x1F):
    a = rng.integers(0, 256, (*shape, 32), dtype=np.uint8)
    a[..., 31] &= top
    return a


def _records(rng, n, depth):
    recs = _rand_fr(rng, n, 8 + depth)
    recs[:, 3, 20:] = 0
    recs[:, 5, 8:] = 0
    recs[:, 5, :8] = (recs[:, 5, :8].view(np.uint64) & np.uint64((1 << depth) - 1)).view(np.uint8)
    recs[:, 6, 20:] = 0
    recs[:, 7, 8:] = 0
    return recs


@pytest.fixture(scope="module")
def dense_key(ctx):
    from owshen_amd import circuit, groth16 as g16
    depth = 32
    n_pad3, n_pad2 = circuit.baseline_shape(depth, dense=True)
    r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2, dense=True)
    assert (r1.n_wires, r1.domain_size) == (1 << 18, 1 << 17)
    blob, vk = g16.setup(ctx, r1, 0x7654321, 0x2345678, 0x3456789, 0x456789A, 0x56789AB)
    pk = g16.ProvingKey(ctx, blob)
    yield depth, n_pad3, n_pad2, blob, vk, pk
    pk.close()
    ctx.release_scratch()


def test_every_sub_batch_of_the_headline_call_meets_the_serial_path_the_c_oracle_and_the_verifier(ctx, dense_key):
    import torch
    from bench import plan_sample
    from owshen_amd import circuit, groth16 as g16
    from oracle.c import binding as oc
    depth, n_pad3, n_pad2, blob, vk, pk = dense_key
    n = 1024
    rng = np.random.default_rng(1024)
    recs_d = ctx.to_device(_records(rng, n, depth))
    rs = _rand_fr(rng, n, 2).reshape(n, 64)
    mode, sizes = pk.plan(n)
    assert mode == "stage pipeline" and len(sizes) >= 4 and sum(sizes) == n, (mode, sizes)   # >= 4 sub-batches: a slot is reused
    piped, pub = circuit.prove_from_inputs(ctx, pk, depth, recs_d, rs, n_pad3, n_pad2, return_public=True)
    # a second pipelined call starts on another scratch slot (the slot counter runs on across calls): same bytes
    piped2 = circuit.prove_from_inputs(ctx, pk, depth, recs_d, rs, n_pad3, n_pad2)
    assert piped2.tobytes() == piped.tobytes()
    # the same call kept one ahead of a DIFFERENT batch (two calls in flight share the slots through their events)
    rs_b = _rand_fr(rng, n, 2).reshape(n, 64)
    j1 = circuit.submit_from_inputs(ctx, pk, depth, recs_d, rs, n_pad3, n_pad2)
    j2 = circuit.submit_from_inputs(ctx, pk, depth, recs_d, rs_b, n_pad3, n_pad2)
    assert j1.wait().tobytes() == piped.tobytes()
    other = j2.wait()
    assert other.tobytes() != piped.tobytes()
    # strictly serial: one stream, one scratch slot, no events
    ctx.set_lanes(1)
    try:
        assert pk.plan(n)[0] == "serial"
        serial = circuit.prove_from_inputs(ctx, pk, depth, recs_d, rs, n_pad3, n_pad2)
        serial_b = circuit.prove_from_inputs(ctx, pk, depth, recs_d, rs_b, n_pad3, n_pad2)
    finally:
        ctx.set_lanes(2)
    diff = np.nonzero((serial != piped).any(axis=1))[0]
    assert diff.size == 0, f"pipelined and serial proofs differ at {diff[:8].tolist()} (plan {sizes})"
    assert serial_b.tobytes() == other.tobytes()
    # the C restatement on the first and last proof of every sub-batch
    idx, which = plan_sample(sizes, 2 * len(sizes))
    assert sorted(set(which)) == list(range(len(sizes)))
    ck = oc.prepared_key_from_blob(blob)
    wit = ctx.to_host(circuit.witness(ctx, depth, recs_d[torch.as_tensor(idx, device=recs_d.device)].contiguous(), n_pad3, n_pad2))
    ncpu = os.cpu_count() or 1
    groups = max(1, min(len(idx), ncpu // 16))

    def one(j):
        i = idx[j]
        r, s = int.from_bytes(rs[i, :32].tobytes(), "little"), int.from_bytes(rs[i, 32:].tobytes(), "little")
        return ck.prove(wit[j], r, s, threads=max(1, ncpu // groups)) == piped[i].tobytes()

    with ThreadPoolExecutor(groups) as ex:
        same = list(ex.map(one, range(len(idx))))
    assert all(same), f"GPU proofs differ from the C restatement at indices {[i for i, ok in zip(idx, same) if not ok]} (plan {sizes})"
    assert (wit[:, 1:7] == pub[idx]).all()
    # the verifier on ALL of them
    vkb = g16.vk_to_bytes(vk)
    with ThreadPoolExecutor(min(ncpu, 256)) as ex:
        ok = list(ex.map(lambda i: g16.verify(vkb, pub[i], piped[i].tobytes()), range(n)))
        cross = list(ex.map(lambda i: g16.verify(vkb, pub[i + 1], piped[i].tobytes()), range(0, n - 1, 97)))
    assert all(ok), f"og_verify refuses proofs {[i for i, v in enumerate(ok) if not v][:8]}"
    assert not any(cross)


def test_sparse_padding_pipeline_equals_serial_at_768_proofs(ctx):
    """the padding as built (three distinct wire lists: three digit sorts per sub-batch, queries of different window sizes)
    through the pipeline at 768 proofs = 4 sub-batches, against the serial path and -- first / last of each -- the C oracle"""
    import torch
    from bench import plan_sample
    from owshen_amd import circuit, groth16 as g16
    from oracle.c import binding as oc
    depth = 32
    n_pad3, n_pad2 = circuit.baseline_shape(depth, dense=False)
    r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2, dense=False)
    blob, vk = g16.setup(ctx, r1, 0x1111111, 0x2222222, 0x3333333, 0x4444444, 0x5555555)
    pk = g16.ProvingKey(ctx, blob)
    n = 768
    rng = np.random.default_rng(768)
    recs_d = ctx.to_device(_records(rng, n, depth))
    rs = _rand_fr(rng, n, 2).reshape(n, 64)
    mode, sizes = pk.plan(n)
    assert mode == "stage pipeline" and len(sizes) >= 4, (mode, sizes)
    piped = circuit.prove_from_inputs(ctx, pk, depth, recs_d, rs, n_pad3, n_pad2)
    ctx.set_lanes(1)
    try:
        serial = circuit.prove_from_inputs(ctx, pk, depth, recs_d, rs, n_pad3, n_pad2)
    finally:
        ctx.set_lanes(2)
    assert serial.tobytes() == piped.tobytes()
    idx, _ = plan_sample(sizes, 2 * len(sizes))
    ck = oc.prepared_key_from_blob(blob)
    wit = ctx.to_host(circuit.witness(ctx, depth, recs_d[torch.as_tensor(idx, device=recs_d.device)].contiguous(), n_pad3, n_pad2))
    for j, i in enumerate(idx):
        r, s = int.from_bytes(rs[i, :32].tobytes(), "little"), int.from_bytes(rs[i, 32:].tobytes(), "little")
        assert ck.prove(wit[j], r, s) == piped[i].tobytes(), (i, sizes)
    pk.close()
    ctx.release_scratch()


def test_natural_statement_in_sub_batches_of_up_to_1024_proofs(ctx):
    """the natural depth-32 statement (26 385 wires, what `withdraw_handler` would prove: 15-bit windows for its 13 k - 33 k-point
    queries, A sharing L's digit sort, L + H in one bucket set) at 2600 proofs: `choose_sub_batch` lets a small statement's
    sub-batches grow to 1024 proofs (plan 256 + 3 x 782: four sub-batches over the two scratch slots), so the pipelined call must equal
    the strictly serial one byte for byte, the first and last proof of every sub-batch the C restatement's, and og_verify accepts
    every proof with the public inputs the call returned"""
    import torch
    from bench import plan_sample
    from owshen_amd import circuit, groth16 as g16
    from oracle.c import binding as oc
    depth, n_pad3, n_pad2 = 32, 0, 0
    r1 = circuit.withdraw_r1cs_native(ctx, depth, n_pad3, n_pad2)
    blob, vk = g16.setup(ctx, r1, 0x1357, 0x2468, 0x369C, 0x48AC, 0x5BDF)
    pk = g16.ProvingKey(ctx, blob)
    assert set(pk.windows().values()) == {15}, pk.windows()
    n = 2600
    rng = np.random.default_rng(2600)
    recs_d = ctx.to_device(_records(rng, n, depth))
    rs = _rand_fr(rng, n, 2).reshape(n, 64)
    mode, sizes = pk.plan(n)
    assert mode == "stage pipeline" and len(sizes) >= 4 and max(sizes) > 256 and sum(sizes) == n, (mode, sizes)
    piped, pub = circuit.prove_from_inputs(ctx, pk, depth, recs_d, rs, n_pad3, n_pad2, return_public=True)
    ctx.set_lanes(1)
    try:
        serial = circuit.prove_from_inputs(ctx, pk, depth, recs_d, rs, n_pad3, n_pad2)
    finally:
        ctx.set_lanes(2)
    diff = np.nonzero((serial != piped).any(axis=1))[0]
    assert diff.size == 0, f"pipelined and serial proofs differ at {diff[:8].tolist()} (plan {sizes})"
    idx, which = plan_sample(sizes, 2 * len(sizes))
    assert sorted(set(which)) == list(range(len(sizes)))
    ck = oc.prepared_key_from_blob(blob)
    wit = ctx.to_host(circuit.witness(ctx, depth, recs_d[torch.as_tensor(idx, device=recs_d.device)].contiguous(), n_pad3, n_pad2))
    for j, i in enumerate(idx):
        r, s = int.from_bytes(rs[i, :32].tobytes(), "little"), int.from_bytes(rs[i, 32:].tobytes(), "little")
        assert ck.prove(wit[j], r, s) == piped[i].tobytes(), (i, sizes)
    vkb = g16.vk_to_bytes(vk)
    with ThreadPoolExecutor(32) as ex:
        ok = list(ex.map(lambda i: g16.verify(vkb, pub[i], piped[i].tobytes()), range(n)))
    assert all(ok), f"og_verify refuses proofs {[i for i, v in enumerate(ok) if not v][:8]}"
    assert not g16.verify(vkb, pub[1], piped[0].tobytes())
    pk.close()
    ctx.release_scratch()


@pytest.mark.parametrize("world,n", [(8, 1), (8, 16), (4, 96)])
def test_window_sharded_full_size_proofs_equal_the_unsharded_call(ctx, dense_key, world, n):
    """Window-sharded proving at the headline's shape (round 6; BASELINE.json configs[3] as written): the 15 seventeen-bit windows
    of the 2^18-wire key over 8 (4) owners -- uneven: 2 2 2 2 2 2 2 1 -- each owner's front half run in turn on this GPU
    (og_withdraw_prove_partials_d: one request fanned out over the streams, 16 requests, and 96 in the stage pipeline with the
    merged L + H bucket set), the blocks laid out as the all-gather leaves them, og_prove_from_partials_d: byte-identical to
    og_withdraw_prove_batch_d, and the first / last proof to the C restatement"""
    import torch
    from owshen_amd import circuit
    from oracle.c import binding as oc
    depth, n_pad3, n_pad2, blob, vk, pk = dense_key
    rng = np.random.default_rng(100 * world + n)
    recs_d = ctx.to_device(_records(rng, n, depth))
    rs = _rand_fr(rng, n, 2).reshape(n, 64)
    want, want_pub = circuit.prove_from_inputs(ctx, pk, depth, recs_d, rs, n_pad3, n_pad2, return_public=True)
    parts = []
    for r in range(world):
        part, pub = circuit.partials_from_inputs(ctx, pk, depth, recs_d, r, world, n_pad3, n_pad2, return_public=True)
        assert pub.tobytes() == want_pub.tobytes()
        parts.append(part.clone())
    got = pk.prove_from_partials(torch.cat(parts), world, rs)
    assert got.tobytes() == want.tobytes()
    wit = ctx.to_host(circuit.witness(ctx, depth, recs_d[[0, n - 1]], n_pad3, n_pad2))
    ck = oc.prepared_key_from_blob(blob)
    for j, t in enumerate((0, n - 1)):
        r_, s_ = int.from_bytes(rs[t][:32].tobytes(), "little"), int.from_bytes(rs[t][32:].tobytes(), "little")
        assert got[t].tobytes() == ck.prove(wit[j], r_, s_), f"proof {t} differs from the C restatement"


def test_host_chains_at_full_size_in_every_schedule(ctx, dense_key):
    """og_set_host_chains on the 2^18-wire key: 1, 8 and 64 requests (each one sub-batch: its queries fanned out over the streams), blocking and as two submitted
    jobs in flight (they stay enqueued, so the assembly runs in og_job_wait from the results and the (r, s) copy the job carries) -- proofs and public inputs are the GPU-only
    call's, the C restatement re-proves the first and last of the 64, and a call above the bound is untouched"""
    from owshen_amd import circuit
    from oracle.c import binding as oc
    depth, n_pad3, n_pad2, blob, vk, pk = dense_key
    rng = np.random.default_rng(64)
    recs = _records(rng, 70, depth)
    rs = _rand_fr(rng, 70, 2).reshape(70, 64)
    recs_d = ctx.to_device(recs)
    want, want_pub = circuit.prove_from_inputs(ctx, pk, depth, recs_d, rs, n_pad3, n_pad2, return_public=True)
    try:
        ctx.set_host_chains(64)
        for n in (1, 8, 64):
            got, pub = circuit.prove_from_inputs(ctx, pk, depth, recs_d[:n].contiguous(), rs[:n], n_pad3, n_pad2, return_public=True)
            assert got.tobytes() == want[:n].tobytes() and pub.tobytes() == want_pub[:n].tobytes(), n
        assert pk.plan(64)[0] == "query fan-out"   # (jobs of this key stay enqueued whatever the schedule: the assembly runs in og_job_wait)
        j1 = circuit.submit_from_inputs(ctx, pk, depth, recs_d[:64].contiguous(), rs[:64], n_pad3, n_pad2)
        j2 = circuit.submit_from_inputs(ctx, pk, depth, recs_d[6:70].contiguous(), rs[6:70], n_pad3, n_pad2)   # two calls in flight, both host-assembled
        assert j1.wait().tobytes() == want[:64].tobytes() and j2.wait().tobytes() == want[6:70].tobytes()
        assert circuit.prove_from_inputs(ctx, pk, depth, recs_d, rs, n_pad3, n_pad2).tobytes() == want.tobytes()   # 70 > 64: the kernels
    finally:
        ctx.set_host_chains(0)
    wit = ctx.to_host(circuit.witness(ctx, depth, recs_d[:64].contiguous(), n_pad3, n_pad2))
    ck = oc.prepared_key_from_blob(blob)
    for k in (0, 63):
        r, s = int.from_bytes(rs[k, :32].tobytes(), "little"), int.from_bytes(rs[k, 32:].tobytes(), "little")
        assert want[k].tobytes() == ck.prove(wit[k], r, s, threads=os.cpu_count() or 1)
