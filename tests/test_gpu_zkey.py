"""snarkjs' .zkey / .wtns files on the GPU through the C ABI; cases in tests/zkey_cases.py (the oracle: oracle/py/zkey.py)."""
import json
import os
import struct
import time

import numpy as np
import pytest

from tests import zkey_cases as cases

pytestmark = pytest.mark.gpu


def test_zkey_import_matches_oracle_and_snarkjs_prover(ctx):
    cases.case_import_matches_oracle_and_snarkjs_prover(ctx, 200, 3)


def test_zkey_import_matches_oracle_at_1024(ctx):
    """domain 2^10: the DFT over points against the oracle's 1023 direct sums (C restatement's Pippenger), byte for byte"""
    cases.case_import_matches_oracle_and_snarkjs_prover(ctx, 1000, 2, python_prover=False)


def test_own_key_through_a_zkey(ctx):
    """domain 2^12: og_setup's H query (tau^j Z(tau) / delta, fixed-base multiplications) == the import's (DFT over the odd-coset
    Lagrange points), byte for byte"""
    cases.case_own_key_through_a_zkey(ctx, 4000, 2)


def test_zkey_export_is_what_snarkjs_would_prove_with(ctx):
    cases.case_export_is_what_snarkjs_would_prove_with(ctx, 120, 2)


def test_wtns(ctx):
    cases.case_wtns(ctx._lib)


def test_prove_files(ctx, tmp_path):
    cases.case_prove_files(ctx, tmp_path)


def test_zkey_refusals(ctx):
    cases.case_refusals(ctx)


def test_withdraw_key_round_trip_at_the_benchmark_size(ctx):
    """BASELINE.json configs[1]'s key -- 2^18 wires, domain 2^17 -- out as a .zkey and back in: every group element of the five
    queries returns byte for byte (the H query through two DFTs over 2^17 points), and a batch of withdraw proofs made with the
    re-imported key (no C matrix: C z = A z o B z) is the batch the original key makes.  Timings go to gpurun_out/."""
    from owshen_amd import circuit, groth16 as g16, zkey as zk
    from tests.test_gpu_fullsize_pipeline import _records, _rand_fr
    depth = 32
    n_pad3, n_pad2 = circuit.baseline_shape(depth, dense=True)
    r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2, dense=True)
    blob, vk = g16.setup(ctx, r1, 0x7654321, 0x2345678, 0x3456789, 0x456789A, 0x56789AB)
    vkb = g16.vk_to_bytes(vk)
    m, l, log_d = struct.unpack("<10Q", blob[:80])[1:4]
    assert (m, log_d) == (1 << 18, 17)
    t0 = time.perf_counter()
    data = zk.export_zkey(ctx, blob, vkb)
    t1 = time.perf_counter()
    pk2, vk2 = zk.import_zkey(ctx, data)
    t2 = time.perf_counter()
    assert vk2 == vkb
    d = 1 << log_d
    pad = lambda n: (n + 31) // 32 * 32                                                    # noqa: E731
    tail = pad(64 * m) * 2 + pad(128 * m) + pad(64 * (m - l - 1)) + pad(64 * (d - 1))
    assert pk2[-tail:] == blob[-tail:]
    assert pk2[80:592] == blob[80:592]
    n = 8
    rng = np.random.default_rng(8)
    recs_d = ctx.to_device(_records(rng, n, depth))
    rs = _rand_fr(rng, n, 2).reshape(n, 64)
    a = g16.ProvingKey(ctx, blob)
    pa, pub = circuit.prove_from_inputs(ctx, a, depth, recs_d, rs, n_pad3, n_pad2, return_public=True)
    a.close()
    b = g16.ProvingKey(ctx, pk2)
    pb = circuit.prove_from_inputs(ctx, b, depth, recs_d, rs, n_pad3, n_pad2)
    b.close()
    ctx.release_scratch()
    assert np.asarray(pa).tobytes() == np.asarray(pb).tobytes()
    assert g16.verify(vk2, pub[0], np.asarray(pb)[0].tobytes()) is True
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "zkey_roundtrip_2p18.json"), "w") as f:
        json.dump({"n_wires": m, "domain": d, "zkey_bytes": len(data), "export_s": round(t1 - t0, 3), "import_s": round(t2 - t1, 3),
                   "queries_byte_identical": True, "proofs_byte_identical": n}, f)


def test_r1cs(ctx):
    cases.case_r1cs(ctx._lib)


def test_zkey_import_with_r1cs(ctx):
    cases.case_import_with_r1cs(ctx, *(300, 4))
