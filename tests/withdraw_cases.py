"""Withdraw-circuit cases shared by the CPU-interpreter run and the GPU run: the product's R1CS builder
and the HIP witness generator against the plain restatement in oracle/py/withdraw.py, then an end-to-end
proof that the oracle pairing check accepts."""
import os
import random

import numpy as np
import pytest

from oracle.py import fields, mimc7, withdraw as ow, groth16 as og16


def _rows(mat):
    out = []
    for r in range(mat.n_rows):
        lo, hi = int(mat.ptr[r]), int(mat.ptr[r + 1])
        out.append(sorted((int(mat.col[k]), int.from_bytes(mat.val[k].tobytes(), "little")) for k in range(lo, hi)))
    return out


def _oracle_rows(cons, which, extra):
    rows = [sorted((w, c % fields.R) for w, c in x[which].items() if c % fields.R) for x in cons]
    return rows + extra


def _inputs(rnd, depth):
    return dict(nullifier=rnd.randrange(fields.R), secret=rnd.randrange(fields.R), amount=rnd.randrange(1 << 64),
                recipient=rnd.randrange(1 << 160), index=rnd.randrange(1 << depth),
                siblings=[rnd.randrange(fields.R) for _ in range(depth)], pad_seed=rnd.randrange(fields.R),
                token=rnd.randrange(1 << 160), chain_id=rnd.randrange(1 << 32))


def _pack(circuit, i):
    return circuit.pack_inputs(i["nullifier"], i["secret"], i["amount"], i["recipient"], i["pad_seed"], i["index"], i["siblings"],
                               token=i["token"], chain_id=i["chain_id"])


def _spec(i, depth, n_pad3, n_pad2):
    return ow.build(depth, i["nullifier"], i["secret"], i["amount"], i["recipient"], i["index"], i["siblings"], i["pad_seed"],
                    n_pad3, n_pad2, token=i["token"], chain_id=i["chain_id"])


def case_r1cs_and_witness_match_spec(ctx, depth, n_pad3, n_pad2, n_proofs=3):
    from owshen_amd import circuit, api
    rnd = random.Random(depth * 1000 + n_pad3 + n_pad2)
    r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2)
    ins = [_inputs(rnd, depth) for _ in range(n_proofs)]
    ins[0]["index"] = (1 << depth) - 1
    if n_proofs > 1:
        ins[1]["index"] = 0
    packed = np.stack([_pack(circuit, i) for i in ins])
    wit = ctx.to_host(circuit.witness(ctx, depth, ctx.to_device(packed), n_pad3, n_pad2))
    for k, i in enumerate(ins):
        m, l, cons, z = _spec(i, depth, n_pad3, n_pad2)
        assert (m, l) == (r1.n_wires, r1.n_pub) and len(cons) == r1.n_constraints
        assert api.bytes_to_ints(wit[k]) == z, f"witness {k}"
        if k == 0:
            ident = [[(w, 1)] for w in range(l + 1)]
            empty = [[] for _ in range(l + 1)]
            assert _rows(r1.a) == _oracle_rows(cons, 0, ident)
            assert _rows(r1.b) == _oracle_rows(cons, 1, empty)
            assert _rows(r1.c) == _oracle_rows(cons, 2, empty)
            # public wires carry what the statement says
            leaf = mimc7.hash2(mimc7.hash2(i["nullifier"], i["secret"]), mimc7.hash2(i["amount"], i["token"]))
            assert leaf == ow.leaf_of(i["nullifier"], i["secret"], i["amount"], i["token"])
            assert z[1] == mimc7.merkle_root_from_path(leaf, i["index"], i["siblings"])[-1]
            assert z[2] == mimc7.hash2(i["nullifier"], 0)
            assert z[3:7] == [i["recipient"], i["amount"], i["token"], i["chain_id"]] and l == 6


def case_host_chains_gives_the_kernels_bytes(ctx, depth, n_pad3, n_pad2, n_proofs=3, prove=True):
    """og_set_host_chains: calls of a handful of requests walk their MiMC7 chains on the host CPU (witness.hip) -- the witnesses
    must be the kernels' (= the spec's) byte for byte, a call above the bound must not take the path, and the fused
    inputs -> proofs call must return the same proofs and public inputs either way"""
    from owshen_amd import circuit, api
    rnd = random.Random(depth * 77 + n_pad3 + n_pad2)
    ins = [_inputs(rnd, depth) for _ in range(n_proofs)]
    ins[0]["index"] = (1 << depth) - 1
    ins[-1]["index"] = 0
    packed = np.stack([_pack(circuit, i) for i in ins])
    want = bytes(ctx.to_host(circuit.witness(ctx, depth, ctx.to_device(packed), n_pad3, n_pad2)))
    try:
        ctx.set_host_chains(n_proofs)          # at the bound: the host walks
        got = ctx.to_host(circuit.witness(ctx, depth, ctx.to_device(packed), n_pad3, n_pad2))
        assert bytes(got) == want
        for k, i in enumerate(ins):
            assert api.bytes_to_ints(got[k]) == _spec(i, depth, n_pad3, n_pad2)[3], f"witness {k}"
        one = ctx.to_host(circuit.witness(ctx, depth, ctx.to_device(packed[1:2]), n_pad3, n_pad2))   # one request: no helper threads
        assert bytes(one) == want[len(want) // n_proofs:2 * (len(want) // n_proofs)]
        ctx.set_host_chains(n_proofs - 1)      # above the bound: the kernels
        assert bytes(ctx.to_host(circuit.witness(ctx, depth, ctx.to_device(packed), n_pad3, n_pad2))) == want
        if prove:
            _r1, _blob, _vk, pk, close = _key(ctx, depth, n_pad3, n_pad2)
            rs = [(rnd.randrange(fields.R), rnd.randrange(fields.R)) for _ in ins]
            ctx.set_host_chains(0)
            p0, pub0 = circuit.prove_from_inputs(ctx, pk, depth, ctx.to_device(packed), rs, n_pad3, n_pad2, return_public=True)
            ctx.set_host_chains(16)
            p1, pub1 = circuit.prove_from_inputs(ctx, pk, depth, ctx.to_device(packed), rs, n_pad3, n_pad2, return_public=True)
            assert p0.tobytes() == p1.tobytes() and pub0.tobytes() == pub1.tobytes()
            # caller-supplied witnesses (og_prove_batch_d): the same mode assembles their proofs on the host too
            wit_d = ctx.to_device(np.frombuffer(want, dtype=np.uint8).reshape(n_proofs, -1, 32))
            q1 = pk.prove_batch_device(wit_d, rs)
            ctx.set_host_chains(0)
            assert pk.prove_batch_device(wit_d, rs).tobytes() == q1.tobytes() == p0.tobytes()
            close()
        for bad in (-1, 65):
            with pytest.raises(Exception):
                ctx.set_host_chains(bad)
    finally:
        ctx.set_host_chains(0)


_TOXIC = (5, 6, 7, 8, 9)


def _key(ctx, depth, n_pad3, n_pad2, dense=False, toxic=_TOXIC):
    """(r1cs, key blob, vk, loaded key, close) for a withdraw shape.  On the CPU interpreter (tests/emu.Ctx) set-up and the key
    upload take ~25 s per key -- 254 doublings per point for the window tables, one lane at a time -- so there the cases of a
    session share one key per (shape, toxic waste) and `close` does nothing; on the GPU every case loads and frees its own."""
    from owshen_amd import circuit, groth16 as g16
    shared = type(ctx).__module__ == "tests.emu"
    cache = ctx.__dict__.setdefault("_withdraw_keys", {}) if shared else {}
    k = (depth, n_pad3, n_pad2, dense, tuple(toxic))
    if k not in cache:
        r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2, dense=dense)
        blob, vk = g16.setup(ctx, r1, *toxic)
        cache[k] = (r1, blob, vk, g16.ProvingKey(ctx, blob))
    r1, blob, vk, pk = cache[k]
    return r1, blob, vk, pk, (lambda: None) if shared else pk.close


def case_native_builder_equals_python_builder(ctx, depth, n_pad3, n_pad2, dense):
    """og_withdraw_r1cs (the C-ABI builder a Rust host calls) == owshen_amd/circuit.py, row by row"""
    from owshen_amd import circuit
    py = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2, dense=dense)
    nat = circuit.withdraw_r1cs_native(ctx, depth, n_pad3, n_pad2, dense=dense)
    assert (nat.n_wires, nat.n_pub, nat.n_constraints, nat.log_d) == (py.n_wires, py.n_pub, py.n_constraints, py.log_d)
    for name in "abc":
        assert _rows(getattr(nat, name)) == _rows(getattr(py, name)), name


def case_dense_rows(ctx, depth, n_pad3, n_pad2):
    """withdraw_r1cs(dense=True) = the spec's rows + the two density rows; the key then keeps every wire in A and B"""
    from owshen_amd import circuit, groth16 as g16
    base = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2)
    r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2, dense=True)
    m, nc = base.n_wires, base.n_constraints
    assert (r1.n_wires, r1.n_constraints, r1.n_pub) == (m, nc + circuit.N_DENSE_ROWS, base.n_pub)
    allw, extra = [(w, 1) for w in range(m)], base.n_pub + 1
    for name, dense_rows in (("a", [allw, []]), ("b", [[], allw]), ("c", [[], []])):
        rows, brows = _rows(getattr(r1, name)), _rows(getattr(base, name))
        assert rows[:nc] == brows[:nc] and rows[nc:nc + 2] == dense_rows and rows[nc + 2:] == brows[nc:]
        assert len(rows) == nc + 2 + extra
    _r1, _blob, _vk, pk, close = _key(ctx, depth, n_pad3, n_pad2, dense=True)
    dn = pk.density()
    assert (dn["a"], dn["b"], dn["h"]) == (m, m, (1 << pk.log_d) - 1) and dn["l"] <= m - base.n_pub - 1
    # A, B and L keep ONE wire list here (L's few missing bases as points at infinity: one digit sort per sub-batch), so their
    # tables share a window size; og_pk_density still counts real bases only (the assertion above)
    w = pk.windows()
    assert w["a"] == w["b"] == w["l"] and set(w.values()) <= {8, 12, 16, 17}
    close()


def case_withdraw_end_to_end(ctx, depth, n_pad3, n_pad2, dense=False):
    """GPU witness -> GPU proof -> oracle pairing verify; and the C oracle proves the same bytes."""
    from owshen_amd import circuit, groth16 as g16
    from tests.r1cs_util import oracle_c_key_from_blob
    rnd = random.Random(31)
    _r1, blob, vk, pk, close = _key(ctx, depth, n_pad3, n_pad2, dense=dense)
    ins = [_inputs(rnd, depth) for _ in range(2)]
    packed = np.stack([_pack(circuit, i) for i in ins])
    wit_d = circuit.witness(ctx, depth, ctx.to_device(packed), n_pad3, n_pad2)
    rs = [(rnd.randrange(fields.R), rnd.randrange(fields.R)) for _ in ins]
    proofs = pk.prove_batch_device(wit_d, rs)
    # the fused entry point (inputs -> proofs, witnesses generated inside the prover's lanes) gives the same bytes
    assert circuit.prove_from_inputs(ctx, pk, depth, ctx.to_device(packed), rs, n_pad3, n_pad2).tobytes() == proofs.tobytes()
    wit = ctx.to_host(wit_d)
    ck = oracle_c_key_from_blob(blob)
    for k in range(2):
        assert proofs[k].tobytes() == ck.prove(wit[k], *rs[k])
    # verifying key from the product's setup, checked by the oracle pairing (EIP-197 equation)
    from oracle.py.curve import g1_from_bytes, g2_from_bytes
    vk_o = {"alpha_g1": g1_from_bytes(vk["alpha_g1"]), "beta_g2": g2_from_bytes(vk["beta_g2"]),
            "gamma_g2": g2_from_bytes(vk["gamma_g2"]), "delta_g2": g2_from_bytes(vk["delta_g2"]),
            "ic": [g1_from_bytes(vk["ic"][i].tobytes()) for i in range(vk["ic"].shape[0])]}
    pub = [int.from_bytes(wit[0][i].tobytes(), "little") for i in range(1, 7)]
    proof = og16.proof_from_bytes(proofs[0].tobytes())
    assert og16.verify(vk_o, pub, proof)
    pub_bad = list(pub)
    pub_bad[2] = (pub_bad[2] + 1) % fields.R  # someone else's recipient
    assert not og16.verify(vk_o, pub_bad, proof)
    for slot in (4, 5):                       # the same proof presented for another token / on another chain
        replay = list(pub)
        replay[slot] = (replay[slot] + 1) % fields.R
        assert not og16.verify(vk_o, replay, proof)
    # ... and the product's own CPU verifier (og_verify) agrees, on both proofs
    vkb = g16.vk_to_bytes(vk)
    assert g16.verify(vkb, pub, proofs[0].tobytes()) and not g16.verify(vkb, pub_bad, proofs[0].tobytes())
    assert g16.verify(vkb, wit[1][1:7], proofs[1].tobytes())
    assert not g16.verify(vkb, wit[1][1:7], proofs[0].tobytes())
    # the fused call hands back the public inputs it computed (root, nullifier_hash) with the ones it was given
    proofs2, pub2 = circuit.prove_from_inputs(ctx, pk, depth, ctx.to_device(packed), rs, n_pad3, n_pad2, return_public=True)
    assert proofs2.tobytes() == proofs.tobytes() and pub2.tobytes() == np.ascontiguousarray(wit[:, 1:7]).tobytes()
    _gate_model_checks(g16.vk_to_bytes(vk), pub, proofs[0].tobytes())
    _snarkjs_files_check(vkb, pub, pub_bad, proofs[0].tobytes())
    flipped = bytearray(proofs[0].tobytes())
    flipped[200] ^= 1
    try:
        bad = og16.proof_from_bytes(bytes(flipped))
        assert not og16.verify(vk_o, pub, bad)
    except (AssertionError, ValueError):
        pass  # not even a curve point any more
    close()


def _snarkjs_files_check(vk_blob, pub, pub_bad, proof256):
    """the proof as `snarkjs groth16 verify`'s three files (owshen_amd/snarkjs_json.py) in front of the second pairing engine
    (oracle/js/bn254_pairing_second.js --snarkjs, V8 BigInt): accepted; refused for someone else's recipient"""
    import shutil
    import subprocess
    import tempfile
    from owshen_amd import snarkjs_json as sj
    node = shutil.which("node") or shutil.which("nodejs")
    if node is None:
        return
    js = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "js", "bn254_pairing_second.js")
    with tempfile.TemporaryDirectory() as d:
        for sub, inputs, want in (("good", pub, "OK"), ("bad", pub_bad, "INVALID")):
            paths = sj.write(os.path.join(d, sub), vk_blob, proof256, inputs)
            r = subprocess.run([node, js, "--snarkjs", paths["verification_key.json"], paths["public.json"], paths["proof.json"]],
                               capture_output=True, text=True, timeout=600,
                               env={k: v for k, v in os.environ.items() if k != "LD_PRELOAD"})   # (sanitizer runs preload their runtime)
            assert r.stdout.strip() == want, (r.stdout, r.stderr)


class GateModel:
    """contracts/OwshenWithdrawGate.sol line by line, on the word-level model of WithdrawVerifier.verifyProof
    (tests/test_evm_words.py: precompiles 6 / 7 / 8 evaluated by the oracle)"""

    def __init__(self, vk_blob, chain_id):
        from owshen_amd import evm
        self.vk_words, self.chain_id = evm.vk_to_evm_words(vk_blob), chain_id
        self.known_root, self.nullified = set(), set()

    def process_withdraw(self, sender, proof256, root, nullifier_hash, token, amount):
        from owshen_amd import evm
        from tests.test_evm_words import verify_proof_model
        assert 0 <= sender < (1 << 160) and 0 <= token < (1 << 160), "`address` values (msg.sender, tokenAddress) are 160 bits by type"
        if not (root < fields.R and nullifier_hash < fields.R):
            return "ERROR: public input is not a field element."
        if not amount < fields.R:
            return "ERROR: amount is not a field element."
        if root not in self.known_root:
            return "ERROR: unknown commitment root."
        if nullifier_hash in self.nullified:
            return "ERROR: withdraw already executed."
        inp = [root, nullifier_hash, sender, amount, token, self.chain_id]
        if not verify_proof_model(self.vk_words, evm.proof_words(proof256), inp):
            return "ERROR: invalid proof."
        self.nullified.add(nullifier_hash)
        return "ok"


def _gate_model_checks(vk_blob, pub, proof256):
    """a proof made for (recipient, amount, token, chain) passes the gate exactly once and for nothing else"""
    root, nh, recipient, amount, token, chain = pub
    gate = GateModel(vk_blob, chain)
    assert gate.process_withdraw(recipient, proof256, root, nh, token, amount) == "ERROR: unknown commitment root."
    gate.known_root.add(root)
    assert gate.process_withdraw(recipient, proof256, root, nh, token ^ 1, amount) == "ERROR: invalid proof."      # another asset
    assert gate.process_withdraw(recipient ^ 1, proof256, root, nh, token, amount) == "ERROR: invalid proof."      # another caller
    assert gate.process_withdraw(recipient, proof256, root, nh, token, amount + 1) == "ERROR: invalid proof."
    other_chain = GateModel(vk_blob, chain + 1)
    other_chain.known_root.add(root)
    assert other_chain.process_withdraw(recipient, proof256, root, nh, token, amount) == "ERROR: invalid proof."   # cross-chain replay
    # non-canonical encodings of the same field elements are refused before anything else (and could not verify either)
    assert gate.process_withdraw(recipient, proof256, root, nh + fields.R, token, amount) == "ERROR: public input is not a field element."
    assert gate.process_withdraw(recipient, proof256, root + fields.R, nh, token, amount) == "ERROR: public input is not a field element."
    assert gate.process_withdraw(recipient, proof256, root, nh, token, amount + fields.R) == "ERROR: amount is not a field element."
    assert gate.process_withdraw(recipient, proof256, root, nh, token, amount) == "ok"
    assert gate.process_withdraw(recipient, proof256, root, nh, token, amount) == "ERROR: withdraw already executed."
    assert gate.process_withdraw(recipient, proof256, root, nh + fields.R, token, amount) == "ERROR: public input is not a field element."  # no second key for a spent note


def case_submitted_batches_equal_blocking_calls(ctx, depth, n_pad3, n_pad2, sizes, third_is_refused=False):
    """og_withdraw_prove_batch_submit_d / og_job_wait: several batches kept one ahead of the waits -> the bytes of the
    blocking call, public inputs included; a third submit while two are pending is refused, not queued"""
    from owshen_amd import circuit, groth16 as g16, api
    rnd = random.Random(77)
    _r1, blob, _vk, pk, close = _key(ctx, depth, n_pad3, n_pad2)
    batches = []
    for n in sizes:
        ins = [_inputs(rnd, depth) for _ in range(n)]
        packed = ctx.to_device(np.stack([_pack(circuit, i) for i in ins]))
        rs = [(rnd.randrange(fields.R), rnd.randrange(fields.R)) for _ in range(n)]
        batches.append((packed, rs))
    want = [circuit.prove_from_inputs(ctx, pk, depth, p, rs, n_pad3, n_pad2, return_public=True) for p, rs in batches]
    jobs, got = [], []
    for p, rs in batches:
        jobs.append(circuit.submit_from_inputs(ctx, pk, depth, p, rs, n_pad3, n_pad2, return_public=True))
        if len(jobs) == 2:
            got.append(jobs.pop(0).wait())
    while jobs:
        got.append(jobs.pop(0).wait())
    for (wp, wpub), (gp, gpub) in zip(want, got):
        assert gp.tobytes() == wp.tobytes() and gpub.tobytes() == wpub.tobytes()
    if third_is_refused:
        import pytest
        # (both must really stay enqueued: on hardware a batch whose sub-batch x wires is below 2^26 proves inside submit and
        # holds no call slot -- round 3's toggle refused the third submit even then)
        last = len(batches) - 1
        a = circuit.submit_from_inputs(ctx, pk, depth, *batches[0], n_pad3, n_pad2)
        b = circuit.submit_from_inputs(ctx, pk, depth, *batches[last], n_pad3, n_pad2)
        with pytest.raises(api.OwshenGpuError, match="already in flight"):
            circuit.submit_from_inputs(ctx, pk, depth, *batches[0], n_pad3, n_pad2)
        with pytest.raises(api.OwshenGpuError, match="already in flight"):      # the blocking call needs a call slot too
            circuit.prove_from_inputs(ctx, pk, depth, *batches[0], n_pad3, n_pad2)
        assert a.wait().tobytes() == want[0][0].tobytes() and b.wait().tobytes() == want[last][0].tobytes()
        assert circuit.prove_from_inputs(ctx, pk, depth, *batches[0], n_pad3, n_pad2).tobytes() == want[0][0].tobytes()
    close()
    return blob, batches


def case_malformed_records_are_rejected(ctx, depth, n_pad3, n_pad2, pipeline=False, quick=False):
    """boundary check of the withdraw input records (og_withdraw_witness_d, og_withdraw_prove_batch_d, the submit form): a field
    >= r or an index outside the tree is OG_ERR_INVALID naming the record and the field; well-formed batches still prove.
    `pipeline`: the witnesses are generated inside the stage pipeline (the records are checked there, per sub-batch)."""
    import pytest
    from owshen_amd import circuit, api
    rnd = random.Random(5)
    _r1, blob, _vk, pk, close = _key(ctx, depth, n_pad3, n_pad2)
    nrec = 2 if quick else 5
    ins = [_inputs(rnd, depth) for _ in range(nrec)]
    good = np.stack([_pack(circuit, i) for i in ins])
    rs = [(rnd.randrange(fields.R), rnd.randrange(fields.R)) for _ in ins]
    want = circuit.prove_from_inputs(ctx, pk, depth, ctx.to_device(good), rs, n_pad3, n_pad2)

    def le(v):
        return np.frombuffer(int(v).to_bytes(32, "little"), dtype=np.uint8)

    def with_field(rec, field, value):
        x = good.copy()
        x[rec, field] = le(value)
        return x

    last, mid = nrec - 1, nrec // 2
    bad_cases = [
        (with_field(last, 0, ins[last]["nullifier"] + fields.R), last, 0, "nullifier"),   # the same nullifier, second encoding
        (with_field(1, 1, fields.R), 1, 1, "secret"),
        (with_field(last, 2, (1 << 256) - 1), last, 2, "amount"),
        (with_field(0, 5, 1 << depth), 0, 5, "index"),                            # one past the last leaf
        (with_field(mid, 5, ins[mid]["index"] | (1 << 64)), mid, 5, "index"),    # bytes above the u64
        (with_field(mid, 8 + depth - 1, fields.R + 5), mid, 8 + depth - 1, "sibling"),
        (with_field(1, 7, fields.R + 1), 1, 7, "chain_id"),
    ]
    two = with_field(last, 6, fields.R)                                           # two malformed records: the first is named
    two[0, 3] = le(fields.R + 9)
    bad_cases.append((two, 0, 3, "recipient"))
    if quick:   # (the CPU interpreter: a proof costs seconds)
        bad_cases = [bad_cases[0], bad_cases[3], bad_cases[5], bad_cases[7]]
    for k, (packed, rec, field, name) in enumerate(bad_cases):
        calls = [lambda d: circuit.witness(ctx, depth, d, n_pad3, n_pad2),
                 lambda d: circuit.prove_from_inputs(ctx, pk, depth, d, rs, n_pad3, n_pad2),
                 lambda d: circuit.submit_from_inputs(ctx, pk, depth, d, rs, n_pad3, n_pad2).wait()]
        if quick:
            calls = [calls[1 + k % 2]] if pipeline and k < 2 else ([] if pipeline else ([calls[0], calls[1 + k % 2]] if k < 2 else calls[:1]))
        for call in calls:
            with pytest.raises(api.OwshenGpuError) as e:
                call(ctx.to_device(packed))
            msg = str(e.value)
            assert e.value.code == -1 and f"record {rec}:" in msg and f"field {field} " in msg and name in msg, msg
    # the largest well-formed values pass the boundary
    edge = with_field(0, 0, fields.R - 1)
    edge[1, 5] = le((1 << depth) - 1)
    circuit.witness(ctx, depth, ctx.to_device(edge), n_pad3, n_pad2)
    # nothing of the failed calls is left in flight: the good batch still proves the same bytes, blocking and submitted
    if quick and not pipeline:
        close()
        return
    assert circuit.prove_from_inputs(ctx, pk, depth, ctx.to_device(good), rs, n_pad3, n_pad2).tobytes() == want.tobytes()
    if not quick:
        assert circuit.submit_from_inputs(ctx, pk, depth, ctx.to_device(good), rs, n_pad3, n_pad2).wait().tobytes() == want.tobytes()
    close()


def case_jobs_are_consumed_once(ctx, depth, n_pad3, n_pad2, stays_enqueued=False, n_proofs=3):
    """og_job_wait / og_job_abandon: a handle is validated against the context's records before it is touched --
    a second wait, a wait after abandon, a foreign pointer are OG_ERR_INVALID; an abandoned job frees its call slot; a blocking
    call between a submit and its wait does not make the next submit fail (the call slot is the first FREE one, not a toggle)"""
    import ctypes as C
    import pytest
    from owshen_amd import circuit, api
    rnd = random.Random(9)
    _r1, _blob, _vk, pk, close = _key(ctx, depth, n_pad3, n_pad2)
    ins = [_inputs(rnd, depth) for _ in range(n_proofs)]
    packed = ctx.to_device(np.stack([_pack(circuit, i) for i in ins]))
    rs = [(rnd.randrange(fields.R), rnd.randrange(fields.R)) for _ in ins]
    want = circuit.prove_from_inputs(ctx, pk, depth, packed, rs, n_pad3, n_pad2)
    lib = ctx._lib
    # submit -> blocking call -> submit -> wait both: round 3 refused the second submit ("two calls are already in flight")
    a = circuit.submit_from_inputs(ctx, pk, depth, packed, rs, n_pad3, n_pad2)
    assert circuit.prove_from_inputs(ctx, pk, depth, packed, rs, n_pad3, n_pad2).tobytes() == want.tobytes()
    b = circuit.submit_from_inputs(ctx, pk, depth, packed, rs, n_pad3, n_pad2)
    assert a.wait().tobytes() == want.tobytes() and b.wait().tobytes() == want.tobytes()
    # double wait through the raw ABI: the second one is refused, not a use-after-free
    j = circuit.submit_from_inputs(ctx, pk, depth, packed, rs, n_pad3, n_pad2)
    h = j._h
    assert j.wait().tobytes() == want.tobytes()
    assert lib.og_job_wait(ctx._h, h) == -1 and b"not a pending job" in lib.og_last_error()
    assert lib.og_job_abandon(ctx._h, h) == -1
    assert lib.og_job_wait(ctx._h, C.c_void_p(0x1000)) == -1       # a pointer this context never handed out
    # abandon: no results, the slot is free again; two more submits fit
    j = circuit.submit_from_inputs(ctx, pk, depth, packed, rs, n_pad3, n_pad2)
    h = j._h
    j.abandon()
    if stays_enqueued:   # (a small circuit's submit proves synchronously and has filled the buffers already)
        assert not j._out.any(), "an abandoned job must not write the caller's buffers"
    assert lib.og_job_wait(ctx._h, h) == -1
    c1 = circuit.submit_from_inputs(ctx, pk, depth, packed, rs, n_pad3, n_pad2)
    c2 = circuit.submit_from_inputs(ctx, pk, depth, packed, rs, n_pad3, n_pad2)
    del c1                                                           # dropped without wait: ProveJob.__del__ abandons it
    c3 = circuit.submit_from_inputs(ctx, pk, depth, packed, rs, n_pad3, n_pad2)
    assert c2.wait().tobytes() == want.tobytes() and c3.wait().tobytes() == want.tobytes()
    # og_job_poll: non-blocking; the job stays pending until it is waited for; a consumed handle is refused.  And a wait on one
    # host thread does not hold the context's lock: another thread submits the next batch meanwhile (the coalescer's shape).
    import threading
    import time
    p1 = circuit.submit_from_inputs(ctx, pk, depth, packed, rs, n_pad3, n_pad2)
    h = p1._h
    got = {}
    th = threading.Thread(target=lambda: got.setdefault("p1", p1.wait()))
    th.start()
    p2 = circuit.submit_from_inputs(ctx, pk, depth, packed, rs, n_pad3, n_pad2)   # while the other thread sits in og_job_wait
    t0 = time.time()
    while not p2.done():
        assert time.time() - t0 < 120, "og_job_poll never reports completion"
        time.sleep(0.0005)
    th.join()
    assert got["p1"].tobytes() == want.tobytes() and p2.done() and p2.wait().tobytes() == want.tobytes()
    flag = C.c_int(7)
    assert lib.og_job_poll(ctx._h, h, C.byref(flag)) == -1 and flag.value == 0     # consumed: refused, not dereferenced
    ctx.release_scratch()                                            # refuses while a job is pending: none is
    assert circuit.prove_from_inputs(ctx, pk, depth, packed, rs, n_pad3, n_pad2).tobytes() == want.tobytes()
    close()
