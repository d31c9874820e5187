"""TEST INFRASTRUCTURE ONLY -- CPU oracles for the owshen_amd HIP prover path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import,
link or execute it, and only as the checker / reported CPU baseline.

PARITY UNPINNED: the reference snapshot (OwshenNetwork/owshen @ 2024_10_08) has
no Groth16 prover, MSM, NTT, MiMC7 or Merkle tree, hence no
golden vectors for this path.  The only real anchor is the BN254 scalar field
``Fp`` and BabyJubJub code in
``/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-236``,
which ``oracle/py/babyjubjub.py`` restates and whose three tests
(``.../babyjubjub/tests.rs:3-51``) are reproduced in ``tests/test_oracle_anchor.py``.
Everything else follows public standards (EIP-196/197 BN254, Groth16 2016,
circomlib MiMC7) and is pinned by three-way agreement
(python big-int oracle <-> C oracle <-> HIP) plus algebraic known-answer tests.
"""
