"""Deposit-statement cases shared by the CPU-interpreter run and the GPU run: the product's two R1CS builders and the HIP witness
kernel against the plain restatement in oracle/py/deposit.py, then proofs -- byte-identical to the C restatement's, accepted by
og_verify for (commitment, depositor) and refused for anything else -- and the boundary check of the records."""
import random

import numpy as np

from oracle.py import deposit as od, fields, mimc7
from tests.withdraw_cases import _rows, _oracle_rows


def _records(circuit, vals):
    return np.stack([circuit.pack_deposit_inputs(*v) for v in vals])


def case_r1cs_and_witness_match_spec(ctx, n=5, seed=1):
    from owshen_amd import api, circuit
    rnd = random.Random(seed)
    vals = [(rnd.randrange(fields.R), rnd.randrange(fields.R), rnd.randrange(1 << 160)) for _ in range(n)]
    vals[0] = (0, 0, 0)
    if n > 1:
        vals[1] = (fields.R - 1, fields.R - 1, fields.R - 1)
    r1 = circuit.deposit_r1cs(ctx.mimc7_constants())
    nat = circuit.deposit_r1cs_native(ctx)
    assert (nat.n_wires, nat.n_pub, nat.n_constraints, nat.log_d) == (r1.n_wires, r1.n_pub, r1.n_constraints, r1.log_d) == (735, 2, 731, 10)
    wit = ctx.to_host(circuit.deposit_witness(ctx, ctx.to_device(_records(circuit, vals))))
    for k, (nu, se, dep) in enumerate(vals):
        m, l, cons, z = od.build(nu, se, dep)
        assert (m, l) == (r1.n_wires, r1.n_pub) and len(cons) == r1.n_constraints
        assert api.bytes_to_ints(wit[k]) == z, f"deposit witness {k}"
        assert z[1] == mimc7.hash2(nu, se) and z[2] == dep
        if k == 0:
            ident, empty = [[(w, 1)] for w in range(l + 1)], [[] for _ in range(l + 1)]
            for name, which, extra in (("a", 0, ident), ("b", 1, empty), ("c", 2, empty)):
                assert _rows(getattr(r1, name)) == _oracle_rows(cons, which, extra), name
                assert _rows(getattr(nat, name)) == _rows(getattr(r1, name)), name


def case_deposit_end_to_end(ctx, n=6, seed=2):
    from oracle.c import binding as oc
    from owshen_amd import api, circuit, groth16 as g16
    rnd = random.Random(seed)
    r1 = circuit.deposit_r1cs(ctx.mimc7_constants())
    blob, vk = g16.setup(ctx, r1, 41, 42, 43, 44, 45)
    pk = g16.ProvingKey(ctx, blob)
    vals = [(rnd.randrange(fields.R), rnd.randrange(fields.R), rnd.randrange(1 << 160)) for _ in range(n)]
    recs = _records(circuit, vals)
    rs = [(rnd.randrange(fields.R), rnd.randrange(fields.R)) for _ in range(n)]
    proofs, pub = circuit.deposit_prove(ctx, pk, ctx.to_device(recs), rs, return_public=True)
    wit = ctx.to_host(circuit.deposit_witness(ctx, ctx.to_device(recs)))
    ck = oc.prepared_key_from_blob(blob)
    vkb = g16.vk_to_bytes(vk)
    lib = ctx._lib
    for t, (nu, se, dep) in enumerate(vals):
        assert proofs[t].tobytes() == ck.prove(wit[t], *rs[t]), f"deposit proof {t} differs from the C restatement"
        c = mimc7.hash2(nu, se)
        assert api.bytes_to_ints(pub[t]) == [c, dep]
        assert g16.verify(vkb, [c, dep], proofs[t].tobytes(), lib=lib) is True
        assert g16.verify(vkb, [c, (dep + 1) % fields.R], proofs[t].tobytes(), lib=lib) is False      # another account
        assert g16.verify(vkb, [(c + 1) % fields.R, dep], proofs[t].tobytes(), lib=lib) is False      # another commitment
    # the same witnesses through the generic entry point: the same bytes
    assert pk.prove_batch_device(ctx.to_device(wit), rs).tobytes() == proofs.tobytes()
    # a malformed record is refused before anything is proved, and names its field
    bad = recs.copy()
    bad[3, 1] = np.frombuffer(fields.R.to_bytes(32, "little"), dtype=np.uint8)   # secret = r: a second encoding of 0
    try:
        circuit.deposit_prove(ctx, pk, ctx.to_device(bad), rs)
        raise AssertionError("a malformed deposit record was proved")
    except api.OwshenGpuError as e:
        assert "input record 3: field 1 (secret)" in str(e), str(e)
    # a key of another statement is refused
    pk.close()
