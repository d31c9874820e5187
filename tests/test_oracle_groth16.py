"""The two oracles agree with each other on Groth16 (python big-int <-> C restatement), and the python
pairing check accepts / rejects as the EIP-197 equation says.  Parity is otherwise UNPINNED: the reference
has no prover and no golden vectors for this path."""
import random

import numpy as np

from oracle.py import fields, groth16 as og16
from tests.r1cs_util import random_r1cs, oracle_c_key


def test_python_and_c_oracle_proofs_are_byte_identical_and_verify():
    n_pub = 2
    n_wires, cons, z = random_r1cs(23, n_pub, seed=1, bool_every=3)
    ro = og16.R1CS(n_wires, n_pub, cons)
    assert ro.is_satisfied(z)
    rnd = random.Random(2)
    toxic = tuple(rnd.randrange(1, fields.R) for _ in range(5))
    pk, vk = og16.setup(ro, *toxic)
    r, s = rnd.randrange(fields.R), rnd.randrange(fields.R)
    proof = og16.prove(pk, ro, z, r, s)
    pb = og16.proof_to_bytes(proof)
    assert og16.proof_from_bytes(pb) == proof
    ck = oracle_c_key(ro, pk)
    w = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in z), dtype=np.uint8).reshape(-1, 32).copy()
    assert ck.prove(w, r, s) == pb
    assert og16.verify(vk, z[1:n_pub + 1], proof)
    assert not og16.verify(vk, [z[1], (z[2] + 1) % fields.R], proof)
    A, B, C = proof
    from oracle.py.curve import G1
    assert not og16.verify(vk, z[1:n_pub + 1], (A, B, G1.add(C, A)))
    # different blinding -> different proof, still valid
    proof2 = og16.prove(pk, ro, z, r + 1, s)
    assert proof2 != proof and og16.verify(vk, z[1:n_pub + 1], proof2)
    cd = og16.proof_to_evm_calldata(proof)
    assert len(cd) == 256 and int.from_bytes(cd[:32], "big") == A[0] and int.from_bytes(cd[64:96], "big") == B[0][1]
