"""Generates tests/golden/golden.json from the PYTHON ORACLE (oracle/py) -- the committed fixtures every
implementation (python oracle, C oracle, CPU-interpreted kernels, GPU kernels) must reproduce.

PARITY UNPINNED caveat: the reference (OwshenNetwork/owshen @ 2024_10_08) has no implementation of this path, so
these vectors pin *this repo's* convention (EIP-196/197 BN254, circomlib MiMC7, Groth16 with arkworks query naming),
not reference outputs.  The only reference-derived constants are the Fr modulus / generator / byte order
(/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11) and BabyJubJub's BASE / ORDER
(mod.rs:177-188), checked in tests/test_oracle_anchor.py.

    python tests/golden/make_golden.py        (rewrites golden.json; deterministic)
"""
import hashlib
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.py import fields, mimc7, ntt, groth16 as og16, withdraw  # noqa: E402
from oracle.py.curve import G1, G2, G1_GEN, G2_GEN, g1_to_bytes, g2_to_bytes  # noqa: E402
from tests.r1cs_util import random_r1cs  # noqa: E402


def main():
    rnd = random.Random(20241008)
    R = fields.R
    g = {}
    g["mimc7_constants_sha256"] = hashlib.sha256(b"".join(c.to_bytes(32, "little") for c in mimc7.CONSTANTS)).hexdigest()
    g["mimc7_c1"] = str(mimc7.CONSTANTS[1])
    g["mimc7_hash2"] = [[str(a), str(b), str(mimc7.hash2(a, b))] for a, b in [(1, 2), (0, 0), (R - 1, 12345)]]
    leaf, idx, sib = 11, 5, [21, 22, 23]
    g["mimc7_path"] = {"leaf": str(leaf), "index": idx, "siblings": [str(s) for s in sib],
                       "nodes": [str(x) for x in mimc7.merkle_root_from_path(leaf, idx, sib)]}
    x = [rnd.randrange(R) for _ in range(8)]
    w = fields.fr_root_of_unity(3)
    g["ntt8"] = {"in": [str(v) for v in x], "out": [str(v) for v in ntt.ntt(x)],
                 "coset_out": [str(v) for v in ntt.coset_ntt(x)]}
    ks = [rnd.randrange(R) for _ in range(6)]
    sc = [rnd.randrange(R) for _ in range(6)]
    g["msm"] = {"base_scalars": [str(k) for k in ks], "scalars": [str(s) for s in sc],
                "g1": g1_to_bytes(G1.msm_naive(sc, [G1.mul(G1_GEN, k) for k in ks])).hex(),
                "g2": g2_to_bytes(G2.msm_naive(sc, [G2.mul(G2_GEN, k) for k in ks])).hex()}
    # a full Groth16 instance: random R1CS (seeded), toxic waste, witness, (r, s) -> 256-byte proof
    n_pub = 2
    n_wires, cons, z = random_r1cs(9, n_pub, seed=424242)
    ro = og16.R1CS(n_wires, n_pub, cons)
    toxic = [rnd.randrange(1, R) for _ in range(5)]
    pk, vk = og16.setup(ro, *toxic)
    r, s = rnd.randrange(R), rnd.randrange(R)
    proof = og16.prove(pk, ro, z, r, s)
    assert og16.verify(vk, z[1:n_pub + 1], proof)
    g["groth16"] = {"r1cs_seed": 424242, "n_constraints": 9, "n_pub": n_pub, "toxic": [str(t) for t in toxic], "r": str(r), "s": str(s),
                    "witness": [str(v) for v in z], "proof": og16.proof_to_bytes(proof).hex(),
                    "vk_alpha_g1": g1_to_bytes(vk["alpha_g1"]).hex(), "vk_delta_g2": g2_to_bytes(vk["delta_g2"]).hex(),
                    "evm_calldata": og16.proof_to_evm_calldata(proof).hex()}
    # the withdraw statement at depth 2 with a little padding: public wires and a witness digest
    m, l, wcons, wz = withdraw.build(2, 101, 202, 303, 404, 2, [505, 606], pad_seed=707, n_pad3=3, n_pad2=70, token=808, chain_id=909)
    g["withdraw_depth2"] = {"n_wires": m, "n_constraints": len(wcons), "root": str(wz[1]), "nullifier_hash": str(wz[2]),
                            "witness_sha256": hashlib.sha256(b"".join(v.to_bytes(32, "little") for v in wz)).hexdigest()}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.json"), "w") as f:
        json.dump(g, f, indent=1, sort_keys=True)
    print("wrote golden.json")


if __name__ == "__main__":
    main()
