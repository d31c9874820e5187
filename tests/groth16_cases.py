"""Groth16 parity cases shared by the CPU-interpreter run (test_emu_groth16.py) and the GPU run
(test_gpu_groth16.py): byte-identical proofs for injected (r, s) against the python oracle, the C
oracle as a second witness, oracle pairing check accepts / rejects."""
import random

import numpy as np
import pytest

from oracle.py import fields, groth16 as og16
from oracle.py.curve import g1_to_bytes, g2_to_bytes
from tests.r1cs_util import random_r1cs


def _wit(z):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in z), dtype=np.uint8).reshape(-1, 32).copy()


def _oracle_key(r1cs_o, toxic):
    return og16.setup(r1cs_o, *toxic)


def case_setup_matches_oracle_setup(ctx, n_constraints, n_pub):
    from owshen_amd import groth16 as g16
    n_wires, cons, z = random_r1cs(n_constraints, n_pub, seed=n_constraints)
    rnd = random.Random(99)
    toxic = tuple(rnd.randrange(1, fields.R) for _ in range(5))
    r1 = g16.R1CS.from_constraints(n_wires, n_pub, cons)
    ro = og16.R1CS(n_wires, n_pub, cons)
    assert r1.domain_size == ro.domain_size
    blob, vk = g16.setup(ctx, r1, *toxic)
    pk_o, vk_o = _oracle_key(ro, toxic)
    # verifying key
    assert vk["alpha_g1"] == g1_to_bytes(vk_o["alpha_g1"])
    assert vk["beta_g2"] == g2_to_bytes(vk_o["beta_g2"])
    assert vk["gamma_g2"] == g2_to_bytes(vk_o["gamma_g2"])
    assert vk["delta_g2"] == g2_to_bytes(vk_o["delta_g2"])
    assert vk["ic"].tobytes() == b"".join(g1_to_bytes(p) for p in vk_o["ic"])
    # proving key queries sit at the tail of the blob in a fixed order
    tail = b"".join(g1_to_bytes(p) for p in pk_o["a_query"])
    tail += b"".join(g1_to_bytes(p) for p in pk_o["b_g1_query"])
    tail += b"".join(g2_to_bytes(p) for p in pk_o["b_g2_query"])
    lq = b"".join(g1_to_bytes(p) for p in pk_o["l_query"])
    hq = b"".join(g1_to_bytes(p) for p in pk_o["h_query"])
    tail += lq + b"\0" * (-len(lq) % 32) + hq + b"\0" * (-len(hq) % 32)
    assert blob[-len(tail):] == tail


def case_prove_batch_matches_oracle_and_verifies(ctx):
    from owshen_amd import groth16 as g16
    n_pub = 2
    n_wires, cons, z0 = random_r1cs(40, n_pub, seed=7, bool_every=4)
    rnd = random.Random(5)
    toxic = tuple(rnd.randrange(1, fields.R) for _ in range(5))
    r1 = g16.R1CS.from_constraints(n_wires, n_pub, cons)
    ro = og16.R1CS(n_wires, n_pub, cons)
    blob, vk = g16.setup(ctx, r1, *toxic)
    pk_o, vk_o = _oracle_key(ro, toxic)
    pk = g16.ProvingKey(ctx, blob)
    assert (pk.n_wires, pk.n_pub, pk.log_d, pk.n_rows) == (n_wires, n_pub, 6, 43)
    # three different witnesses of the same circuit: re-derive the product wires from fresh inputs
    zs = []
    for t in range(3):
        z = list(z0)
        if t:
            r2 = random.Random(100 + t)
            for i in range(1, n_wires - len(cons)):
                z[i] = r2.randrange(fields.R)
            for k, (a, b, c) in enumerate(cons):
                av = sum(v * z[i] for i, v in a.items()) % fields.R
                bv = sum(v * z[i] for i, v in b.items()) % fields.R
                z[n_wires - len(cons) + k] = av * bv % fields.R
        assert ro.is_satisfied(z)
        zs.append(z)
    rs = [(rnd.randrange(fields.R), rnd.randrange(fields.R)) for _ in zs]
    rs[1] = (0, 0)  # unblinded proof: r = s = 0 is legal
    got = pk.prove_batch(np.stack([_wit(z) for z in zs]), rs)
    for z, (r, s), p in zip(zs, rs, got):
        want = og16.proof_to_bytes(og16.prove(pk_o, ro, z, r, s))
        assert p.tobytes() == want
    # oracle pairing check (mirrors the EIP-197 equation) accepts proof 0 and rejects a wrong public input
    proof = og16.proof_from_bytes(got[0].tobytes())
    assert og16.verify(vk_o, zs[0][1:n_pub + 1], proof)
    assert not og16.verify(vk_o, [zs[0][1], (zs[0][2] + 1) % fields.R], proof)
    # single-proof entry point, device-resident witnesses, sub-batching all give the same bytes
    assert pk.prove(_wit(zs[2]), *rs[2]) == got[2].tobytes()
    assert pk.prove_batch_device(ctx.to_device(np.stack([_wit(z) for z in zs])), rs).tobytes() == got.tobytes()


def case_unsatisfied_witness_is_rejected(ctx):
    from owshen_amd import groth16 as g16
    from owshen_amd.api import OwshenGpuError
    n_wires, cons, z = random_r1cs(10, 1, seed=3)
    r1 = g16.R1CS.from_constraints(n_wires, 1, cons)
    blob, _vk = g16.setup(ctx, r1, 11, 12, 13, 14, 15)
    pk = g16.ProvingKey(ctx, blob)
    bad = list(z)
    bad[-1] = (bad[-1] + 1) % fields.R
    with pytest.raises(OwshenGpuError) as e:
        pk.prove_batch(np.stack([_wit(z), _wit(bad)]), [(1, 2), (3, 4)])
    assert e.value.code == -4 and "witness 1" in str(e.value)
    # host witnesses cross to the device sub-batch by sub-batch inside the pipeline (prove_enqueue's z_host): the index
    # reported is the caller's whatever sub-batch holds it, and the failed call leaves nothing in flight
    w9 = [_wit(z)] * 9
    w9[6] = _wit(bad)
    with pytest.raises(OwshenGpuError) as e:
        pk.prove_batch(np.stack(w9), [(k + 1, k + 2) for k in range(9)])
    assert e.value.code == -4 and "witness 6" in str(e.value)
    w9[0] = _wit(bad)                      # ... and of two bad slabs, the earlier one is reported
    with pytest.raises(OwshenGpuError) as e:
        pk.prove_batch(np.stack(w9), [(k + 1, k + 2) for k in range(9)])
    assert e.value.code == -4 and "witness 0" in str(e.value)
    good = pk.prove_batch(np.stack([_wit(z)] * 9), [(k + 1, k + 2) for k in range(9)])
    from tests.r1cs_util import oracle_c_key_from_blob
    ck = oracle_c_key_from_blob(blob)
    for k in (0, 1, 2, 8):
        assert good[k].tobytes() == ck.prove(_wit(z), k + 1, k + 2)


def case_noncanonical_witness_is_rejected(ctx):
    """boundary check (include/owshen_gpu.h, og_prove_batch*): a wire >= r is OG_ERR_INVALID naming witness and wire -- even
    though z + r satisfies every row mod r -- on the host-witness and the device-witness entry points; it comes before
    'does not satisfy'; the failed call leaves nothing behind"""
    from owshen_amd import groth16 as g16
    from owshen_amd.api import OwshenGpuError
    n_wires, cons, z = random_r1cs(10, 1, seed=5)
    r1 = g16.R1CS.from_constraints(n_wires, 1, cons)
    blob, _vk = g16.setup(ctx, r1, 21, 22, 23, 24, 25)
    pk = g16.ProvingKey(ctx, blob)
    good = _wit(z)

    def raw(vals):
        return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(len(vals), 32).copy()

    shifted = list(z)
    shifted[3] = z[3] + fields.R                       # the same residue, another encoding: satisfies the circuit mod r
    w = np.stack([good, raw(shifted), good])
    for call in (lambda a, rs: pk.prove_batch(a, rs), lambda a, rs: pk.prove_batch_device(ctx.to_device(a), rs)):
        with pytest.raises(OwshenGpuError) as e:
            call(w, [(1, 2), (3, 4), (5, 6)])
        assert e.value.code == -1 and "witness 1" in str(e.value) and "wire 3" in str(e.value), str(e.value)
    exactly_r = list(z)
    exactly_r[n_wires - 1] = fields.R                  # r itself (= 0), in the last wire
    top = list(z)
    top[2] = (1 << 256) - 1
    top[5] = fields.R + 7                              # two offenders: the lowest wire is named
    with pytest.raises(OwshenGpuError) as e:
        pk.prove_batch(np.stack([raw(exactly_r)]), [(1, 2)])
    assert e.value.code == -1 and "witness 0" in str(e.value) and f"wire {n_wires - 1}" in str(e.value)
    with pytest.raises(OwshenGpuError) as e:
        pk.prove_batch(np.stack([good, good, raw(top)]), [(1, 2)] * 3)
    assert e.value.code == -1 and "witness 2" in str(e.value) and "wire 2" in str(e.value)
    # malformed AND unsatisfied witnesses in one batch: the malformed one is reported (OG_ERR_INVALID before OG_ERR_UNSATISFIED)
    bad = list(z)
    bad[-1] = (bad[-1] + 1) % fields.R
    with pytest.raises(OwshenGpuError) as e:
        pk.prove_batch(np.stack([_wit(bad), raw(shifted)]), [(1, 2)] * 2)
    assert e.value.code == -1 and "witness 1" in str(e.value)
    # r - 1 is canonical: accepted at the boundary (this witness then simply does not satisfy the circuit)
    edge = list(z)
    edge[4] = fields.R - 1
    with pytest.raises(OwshenGpuError) as e:
        pk.prove_batch(np.stack([raw(edge)]), [(1, 2)])
    assert e.value.code == -4
    from tests.r1cs_util import oracle_c_key_from_blob
    assert pk.prove_batch(np.stack([good]), [(7, 8)])[0].tobytes() == oracle_c_key_from_blob(blob).prove(good, 7, 8)
    pk.close()


def case_pk_load_rejects_malformed_blobs(ctx):
    from owshen_amd import groth16 as g16
    from owshen_amd.api import OwshenGpuError
    n_wires, cons, _z = random_r1cs(5, 1, seed=4)
    blob, _ = g16.setup(ctx, g16.R1CS.from_constraints(n_wires, 1, cons), 2, 3, 4, 5, 6)
    for mutate in (lambda b: b[:-32], lambda b: b"XXXXXXXX" + b[8:], lambda b: b[:100]):
        with pytest.raises(OwshenGpuError) as e:
            g16.ProvingKey(ctx, mutate(blob))
        assert e.value.code == -1


def case_medium_circuit_vs_c_oracle(ctx, n_constraints, n_proofs, sub_batch=None):
    """Key generated by the product's own setup, handed byte-for-byte to the C oracle prover: proofs for
    injected (r, s) must be identical.  Sizes pick the 12- and 16-bit window paths and sub-batching."""
    import os
    from owshen_amd import groth16 as g16
    from tests.r1cs_util import oracle_c_key_from_blob
    n_pub = 4
    n_wires, cons, z0 = random_r1cs(n_constraints, n_pub, seed=n_constraints, n_free=8, bool_every=5)
    r1 = g16.R1CS.from_constraints(n_wires, n_pub, cons)
    blob, _vk = g16.setup(ctx, r1, 1234567, 7654321, 1111111, 2222222, 3333333)
    pk = g16.ProvingKey(ctx, blob)
    ck = oracle_c_key_from_blob(blob)
    zs = []
    for t in range(n_proofs):
        z = list(z0)
        if t:
            r2 = random.Random(500 + t)
            for i in range(1, n_wires - len(cons)):
                z[i] = r2.randrange(2) if i % 5 == 1 else r2.randrange(fields.R)
            for k, (a, b, c) in enumerate(cons):
                av = sum(v * z[i] for i, v in a.items()) % fields.R
                bv = sum(v * z[i] for i, v in b.items()) % fields.R
                z[n_wires - len(cons) + k] = av * bv % fields.R
        zs.append(_wit(z))
    rnd = random.Random(77)
    rs = [(rnd.randrange(fields.R), rnd.randrange(fields.R)) for _ in zs]
    old = os.environ.get("OG_SUB_BATCH")
    if sub_batch:
        os.environ["OG_SUB_BATCH"] = str(sub_batch)
    try:
        got = pk.prove_batch(np.stack(zs), rs)
    finally:
        if sub_batch:
            if old is None:
                del os.environ["OG_SUB_BATCH"]
            else:
                os.environ["OG_SUB_BATCH"] = old
    for w, (r, s), p in zip(zs, rs, got):
        assert p.tobytes() == ck.prove(w, r, s)


def case_degenerate_circuits(ctx):
    """shapes that stress the density maps and the point-at-infinity paths: B uses only the constant wire
    (B density 1 wire), A uses only the constant wire, a witness that makes h identically zero, r = s = 0."""
    from owshen_amd import groth16 as g16
    from tests.r1cs_util import oracle_c_key_from_blob
    R = fields.R
    rnd = random.Random(123)
    # wires: 0 one, 1 public x, 2..5 private;  constraints are "linear": (LC) * 1 = wire  /  1 * (LC) = wire
    z = [1, 7, 0, 0, 0, 0]
    cons = []
    cons.append(({1: 3, 0: 5}, {0: 1}, {2: 1})); z[2] = (3 * z[1] + 5) % R            # B = one only
    cons.append(({0: 1}, {2: 2, 1: 1}, {3: 1})); z[3] = (2 * z[2] + z[1]) % R        # A = one only
    cons.append(({3: 1}, {0: 1}, {4: 1})); z[4] = z[3]
    cons.append(({4: 1, 2: R - 1}, {0: 1}, {5: 1})); z[5] = (z[4] - z[2]) % R
    n_wires = 6
    ro = og16.R1CS(n_wires, 1, cons)
    assert ro.is_satisfied(z)
    blob, _vk = g16.setup(ctx, g16.R1CS.from_constraints(n_wires, 1, cons), 5, 6, 7, 8, 9)
    pk = g16.ProvingKey(ctx, blob)
    ck = oracle_c_key_from_blob(blob)
    zero_w = [1, 0, 5, 10, 10, 5]                                                    # x = 0: still satisfied
    assert ro.is_satisfied(zero_w)
    wits = [_wit(z), _wit(zero_w)]
    rs = [(rnd.randrange(R), rnd.randrange(R)), (0, 0)]
    got = pk.prove_batch(np.stack(wits), rs)
    for w, (r, s), p in zip(wits, rs, got):
        assert p.tobytes() == ck.prove(w, r, s)
    # a circuit whose every constraint is 0 * 0 = 0 on the private wires: A z = B z = C z = 0 on the constraint rows, h = 0
    cons0 = [({2: 1}, {3: 1}, {}), ({3: 1}, {2: 1}, {})]
    z0 = [1, 9, 0, 0]
    blob0, _ = g16.setup(ctx, g16.R1CS.from_constraints(4, 1, cons0), 15, 16, 17, 18, 19)
    pk0 = g16.ProvingKey(ctx, blob0)
    ck0 = oracle_c_key_from_blob(blob0)
    for r, s in ((0, 0), (3, 4)):
        assert pk0.prove(_wit(z0), r, s) == ck0.prove(_wit(z0), r, s)


def case_random_shapes(ctx, seeds):
    """many small random shapes (no public inputs, one constraint, odd batch sizes, forced sub-batching) against the
    C restatement proving with the very same key bytes"""
    import os
    from owshen_amd import groth16 as g16
    from tests.r1cs_util import oracle_c_key_from_blob
    R = fields.R
    for seed in seeds:
        rnd = random.Random(seed)
        n_cons, n_pub = rnd.randrange(1, 70), rnd.randrange(0, 5)
        n_wires, cons, z0 = random_r1cs(n_cons, n_pub, seed=seed, n_free=rnd.randrange(1, 6), bool_every=rnd.choice([0, 2, 3]))
        blob, _vk = g16.setup(ctx, g16.R1CS.from_constraints(n_wires, n_pub, cons), *(rnd.randrange(1, R) for _ in range(5)))
        pk = g16.ProvingKey(ctx, blob)
        ck = oracle_c_key_from_blob(blob)
        n = rnd.randrange(1, 6)
        zs = []
        for t in range(n):
            z = list(z0)
            if t:
                for i in range(1, n_wires - n_cons):
                    z[i] = rnd.choice([0, 1, rnd.randrange(R)])
                for k, (a, b, c) in enumerate(cons):
                    av = sum(v * z[i] for i, v in a.items()) % R
                    bv = sum(v * z[i] for i, v in b.items()) % R
                    z[n_wires - n_cons + k] = av * bv % R
            zs.append(_wit(z))
        rs = [(rnd.randrange(R), rnd.randrange(R)) for _ in zs]
        old = os.environ.get("OG_SUB_BATCH")
        os.environ["OG_SUB_BATCH"] = str(rnd.randrange(1, 4))
        try:
            got = pk.prove_batch(np.stack(zs), rs)
        finally:
            if old is None:
                del os.environ["OG_SUB_BATCH"]
            else:
                os.environ["OG_SUB_BATCH"] = old
        for w, (r, s), p in zip(zs, rs, got):
            assert p.tobytes() == ck.prove(w, r, s), (seed, n_cons, n_pub, n)
        pk.close()
