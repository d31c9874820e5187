"""owshen_amd/snarkjs_json.py: the verifying key, a proof and its public inputs as the three JSON files `snarkjs groth16 verify`
reads -- the route by which a maintainer with network access can put a proof of this library in front of a verifier none of
this repository's authors wrote (parity is unpinned because /root/reference holds no prover and no verifier).
Here: the files' shape, and acceptance / refusal by the second pairing engine reading the FILES (oracle/js/
bn254_pairing_second.js --snarkjs), against og_verify of the host-only library on the same bytes.  No GPU, no torch."""
import json
import os
import random
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node") or shutil.which("nodejs")
JS = os.path.join(ROOT, "oracle", "js", "bn254_pairing_second.js")


@pytest.fixture(scope="module")
def statement():
    """a depth-1 withdraw statement: host-made key in the product's formats, proved by the C restatement"""
    from oracle.c import binding as oc
    from oracle.py import fields, keygen, withdraw as spec
    rnd = random.Random(77)
    n_wires, n_pub, cons, z = spec.build(1, rnd.randrange(fields.R), rnd.randrange(fields.R), 7, 9, 1, [rnd.randrange(fields.R)], token=3, chain_id=4)
    pk_blob, vk_blob = keygen.setup_blobs(n_wires, n_pub, cons, *[rnd.randrange(1, fields.R) for _ in range(5)])
    wit = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in z), dtype=np.uint8).reshape(n_wires, 32)
    proof = oc.prepared_key_from_blob(pk_blob).prove(wit, rnd.randrange(fields.R), rnd.randrange(fields.R))
    return vk_blob, bytes(proof), [int(v) for v in z[1:1 + n_pub]]


def test_files_have_the_snarkjs_shape(statement, tmp_path):
    from oracle.py import fields
    from owshen_amd import snarkjs_json as sj
    vk_blob, proof, pub = statement
    paths = sj.write(str(tmp_path), vk_blob, proof, pub)
    vk = json.load(open(paths["verification_key.json"]))
    pr = json.load(open(paths["proof.json"]))
    pb = json.load(open(paths["public.json"]))
    assert vk["protocol"] == pr["protocol"] == "groth16" and vk["curve"] == pr["curve"] == "bn128"
    assert vk["nPublic"] == len(pub) == len(pb) == 6 and len(vk["IC"]) == 7
    assert pb == [str(x) for x in pub]
    # projective with z = 1, decimal strings, canonical
    for g1 in [vk["vk_alpha_1"], pr["pi_a"], pr["pi_c"]] + vk["IC"]:
        assert len(g1) == 3 and g1[2] == "1" and all(isinstance(c, str) and 0 <= int(c) < fields.P for c in g1)
        x, y = int(g1[0]), int(g1[1])
        assert (y * y - x * x * x - 3) % fields.P == 0
    for g2 in (vk["vk_beta_2"], vk["vk_gamma_2"], vk["vk_delta_2"], pr["pi_b"]):
        assert [len(c) for c in g2] == [2, 2, 2] and g2[2] == ["1", "0"]
    # Fq2 as [c0, c1], the real part first: read that way every G2 point is on the twist y^2 = x^3 + 3 / (9 + u); read the other
    # way round it is not
    from oracle.py.curve import G2_B
    for g2 in (vk["vk_beta_2"], vk["vk_gamma_2"], vk["vk_delta_2"], pr["pi_b"]):
        x, y = (int(g2[0][0]), int(g2[0][1])), (int(g2[1][0]), int(g2[1][1]))
        assert fields.f2_sqr(y) == fields.f2_add(fields.f2_mul(fields.f2_sqr(x), x), G2_B)
        xs, ys = x[::-1], y[::-1]
        assert fields.f2_sqr(ys) != fields.f2_add(fields.f2_mul(fields.f2_sqr(xs), xs), G2_B)
    # against the library's byte layout (x.c0 | x.c1 | y.c0 | y.c1, little-endian) and against the EVM emitter, which swaps the halves
    from owshen_amd import evm
    b = proof[64:192]
    assert [int(pr["pi_b"][0][0]), int(pr["pi_b"][0][1])] == [int.from_bytes(b[0:32], "little"), int.from_bytes(b[32:64], "little")]
    w = evm.proof_words(proof)
    assert [w[2], w[3], w[4], w[5]] == [int(pr["pi_b"][0][1]), int(pr["pi_b"][0][0]), int(pr["pi_b"][1][1]), int(pr["pi_b"][1][0])]
    assert [w[0], w[1], w[6], w[7]] == [int(pr["pi_a"][0]), int(pr["pi_a"][1]), int(pr["pi_c"][0]), int(pr["pi_c"][1])]


def test_bad_blobs_are_refused(statement):
    from owshen_amd import snarkjs_json as sj
    vk_blob, proof, _pub = statement
    with pytest.raises(ValueError):
        sj.verification_key(b"XXXXXXXX" + vk_blob[8:])
    with pytest.raises(ValueError):
        sj.verification_key(vk_blob[:-1])
    with pytest.raises(ValueError):
        sj.proof(proof[:-1])
    assert sj.g1_json(bytes(64)) == ["0", "1", "0"]            # the point at infinity, as snarkjs writes it


@pytest.mark.skipif(NODE is None, reason="node is not installed")
def test_second_engine_accepts_the_files_and_og_verify_agrees(statement, tmp_path):
    from oracle.py import fields
    from owshen_amd import snarkjs_json as sj, verify_only
    vk_blob, proof, pub = statement

    def run(d):
        r = subprocess.run([NODE, JS, "--snarkjs"] + [os.path.join(d, n) for n in ("verification_key.json", "public.json", "proof.json")],
                           capture_output=True, text=True, timeout=600)
        assert r.stdout.strip() in ("OK", "INVALID") and r.returncode == (0 if r.stdout.strip() == "OK" else 1), r.stderr
        return r.stdout.strip() == "OK"

    good = str(tmp_path / "good")
    sj.write(good, vk_blob, proof, pub)
    assert run(good) and verify_only.verify(vk_blob, pub, proof)
    # another recipient (public input 2): both refuse
    other = list(pub)
    other[2] = (other[2] + 1) % fields.R
    d = str(tmp_path / "other")
    sj.write(d, vk_blob, proof, other)
    assert not run(d) and not verify_only.verify(vk_blob, other, proof)
    # A and C swapped: still points of the curve, no longer a proof
    swapped = proof[192:256] + proof[64:192] + proof[0:64]
    d = str(tmp_path / "swapped")
    sj.write(d, vk_blob, swapped, pub)
    assert not run(d) and not verify_only.verify(vk_blob, pub, swapped)
    # a public input at r is not a field element: snarkjs refuses it outright ("Public input is not valid")
    d = str(tmp_path / "noncanonical")
    sj.write(d, vk_blob, proof, [pub[0] + fields.R] + pub[1:])
    assert not run(d)


def test_command_line(statement, tmp_path):
    vk_blob, proof, pub = statement
    for name, data in (("vk.bin", vk_blob), ("proof.bin", proof), ("public.bin", b"".join(x.to_bytes(32, "little") for x in pub))):
        (tmp_path / name).write_bytes(data)
    out = tmp_path / "out"
    r = subprocess.run([sys.executable, "-m", "owshen_amd.snarkjs_json", str(tmp_path / "vk.bin"), str(tmp_path / "proof.bin"),
                        str(tmp_path / "public.bin"), str(out)], capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0, r.stderr
    assert sorted(os.listdir(out)) == ["proof.json", "public.json", "verification_key.json"]
    assert json.load(open(out / "public.json")) == [str(x) for x in pub]
    # no GPU library, no torch in that process
    r = subprocess.run([sys.executable, "-c", "import sys, owshen_amd.snarkjs_json; print(sorted(m for m in sys.modules if m.split('.')[0] in ('torch', 'ctypes')))"],
                       capture_output=True, text=True, cwd=ROOT, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "[]", r.stdout + r.stderr
