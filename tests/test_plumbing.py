"""BASELINE.json configs[0] without a GPU: one depth-32 withdraw proof made by the C restatement from a
HOST-generated key in the product's own key format, accepted by the product's CPU verifier (libowshen_verify.so: og_verify,
the `burn_tx` seam, /root/reference/src/blockchain/tx/burn_tx.rs:11-32) and by the WithdrawVerifier.sol word model (the
replacement of /root/reference/contracts/src/Owshen.sol:66-78) -- in a process that loads no HIP library and never imports
torch.  `python bench.py --workload plumbing` is the same code path with a JSON line around it."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_plumbing_line_runs_without_a_gpu():
    env = dict(os.environ)
    env.pop("OG_ORACLE_NATIVE", None)
    env["HIP_VISIBLE_DEVICES"] = ""                     # even on a GPU box: nothing may touch the device
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "plumbing", "--depth", "32"], capture_output=True, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 0 and d["unit"] == "proofs/s" and d["value"] > 0
    assert d["config"]["n_wires"] == 26385 and d["config"]["merkle_depth"] == 32
    p = d["plumbing"]
    assert p["accepted_by_og_verify"] and p["accepted_by_contract_model"]
    assert p["refused_with_another_nullifier_hash"] and p["refused_with_another_recipient"]
    assert p["gpu_libraries_loaded"] == [] and "libowshen_verify.so" in p["shared_objects"] and "libowshen_gpu.so" not in p["shared_objects"]
    assert d["cpu_baseline"]["kind"] == "port"


def test_host_key_generation_equals_the_python_oracle_on_a_small_statement():
    """oracle/py/keygen.py (scalars from the Python oracle, points from the C restatement, bytes in the OWPK0001 / OWVK0001 layouts)
    against oracle/py/groth16.setup point by point, on the depth-1 withdraw statement; the C restatement proves from the blob
    and og_verify (host-only library) accepts"""
    import random
    from oracle.c import binding as oc
    from oracle.py import fields, groth16 as og16, keygen, withdraw as spec
    from oracle.py.curve import g1_to_bytes, g2_to_bytes
    from owshen_amd import verify_only
    rnd = random.Random(5)
    n_wires, n_pub, cons, z = spec.build(1, rnd.randrange(fields.R), rnd.randrange(fields.R), 7, 9, 1, [rnd.randrange(fields.R)], token=3, chain_id=4)
    toxic = [rnd.randrange(1, fields.R) for _ in range(5)]
    pk_blob, vk_blob = keygen.setup_blobs(n_wires, n_pub, cons, *toxic)
    k = oc.parse_pk_blob(pk_blob)
    pk_o, vk_o = og16.setup(og16.R1CS(n_wires, n_pub, cons), *toxic)
    assert bytes(k["alpha_g1"]) == g1_to_bytes(pk_o["alpha_g1"]) and bytes(k["delta_g2"]) == g2_to_bytes(pk_o["delta_g2"])
    for name, conv in (("a_query", g1_to_bytes), ("b_g1_query", g1_to_bytes), ("b_g2_query", g2_to_bytes), ("l_query", g1_to_bytes), ("h_query", g1_to_bytes)):
        assert k[name].tobytes() == b"".join(conv(p) for p in pk_o[name]), name
    assert vk_blob[16 + 64 + 384:] == b"".join(g1_to_bytes(p) for p in vk_o["ic"])
    wit = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in z), dtype=np.uint8).reshape(n_wires, 32)
    r, s = rnd.randrange(fields.R), rnd.randrange(fields.R)
    proof = oc.prepared_key_from_blob(pk_blob).prove(wit, r, s)
    assert proof == og16.proof_to_bytes(og16.prove(pk_o, og16.R1CS(n_wires, n_pub, cons), z, r, s))
    assert verify_only.verify(vk_blob, z[1:7], proof) and not verify_only.verify(vk_blob, z[1:6] + [z[6] + 1], proof)


def _same_key(blob_a, blob_b):
    """two OWPK0001 blobs hold the same key: every point section and constant byte for byte, the three matrices row for row
    (the order of a row's entries is the builder's business: the product's vectorised builder and the spec's dict rows differ)"""
    from oracle.c import binding as oc
    a, b = oc.parse_pk_blob(blob_a), oc.parse_pk_blob(blob_b)
    for k in ("n_wires", "n_pub", "log_d", "n_rows"):
        assert a[k] == b[k], k
    for k in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2"):
        assert bytes(a[k]) == bytes(b[k]), k
    for k in ("a_query", "b_g1_query", "b_g2_query", "l_query", "h_query"):
        assert a[k].tobytes() == b[k].tobytes(), k
    for m in "abc":
        (pa, ca, va), (pb, cb, vb) = a["csr"][m], b["csr"][m]
        assert np.array_equal(pa, pb), m
        rows = np.repeat(np.arange(len(pa) - 1), np.diff(pa.astype(np.int64)))
        oa, ob = np.lexsort((ca, rows)), np.lexsort((cb, rows))
        assert np.array_equal(ca[oa], cb[ob]) and np.array_equal(va[oa], vb[ob]), m


def test_host_key_generation_equals_og_setup_on_the_interpreter():
    """the same comparison the GPU test makes at depth 32, at depth 2 with the CPU interpreter as og_setup's device"""
    import random
    from tests import emu
    from oracle.py import fields, keygen, withdraw as spec
    from owshen_amd import circuit, groth16 as g16
    ctx = emu.Ctx()
    rnd = random.Random(2)
    depth = 2
    n_wires, n_pub, cons, _z = spec.build(depth, 1, 2, 3, 4, 1, [rnd.randrange(fields.R) for _ in range(depth)], token=6, chain_id=7)
    toxic = [rnd.randrange(1, fields.R) for _ in range(5)]
    pk_host, vk_host = keygen.setup_blobs(n_wires, n_pub, cons, *toxic)
    blob, vk = g16.setup(ctx, circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, 0, 0), *toxic)
    ctx.close()
    assert len(blob) == len(pk_host) and g16.vk_to_bytes(vk) == vk_host
    _same_key(blob, pk_host)


@pytest.mark.gpu
def test_host_key_generation_equals_og_setup_at_depth_32(ctx):
    """two key generators, one key: the host-made key of the natural depth-32 statement is og_setup's for the same toxic waste --
    every group element byte for byte, the matrices row for row -- and the verifying keys are the same bytes"""
    import random
    from oracle.py import fields, keygen, withdraw as spec
    from owshen_amd import circuit, groth16 as g16
    rnd = random.Random(32)
    depth = 32
    n_wires, n_pub, cons, _z = spec.build(depth, 1, 2, 3, 4, 5, [rnd.randrange(fields.R) for _ in range(depth)], token=6, chain_id=7)
    toxic = [rnd.randrange(1, fields.R) for _ in range(5)]
    pk_host, vk_host = keygen.setup_blobs(n_wires, n_pub, cons, *toxic)
    blob, vk = g16.setup(ctx, circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, 0, 0), *toxic)
    assert len(blob) == len(pk_host) and g16.vk_to_bytes(vk) == vk_host
    _same_key(blob, pk_host)
