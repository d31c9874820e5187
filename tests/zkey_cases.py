"""snarkjs-file cases shared by the CPU-interpreter run (test_emu_zkey.py) and the GPU run (test_gpu_zkey.py): og_zkey_import /
og_zkey_export / og_wtns_read / og_wtns_write against oracle/py/zkey.py -- the formats and snarkjs' prover restated.  The bar:
the blobs an import makes are the oracle's byte for byte, and a proof made with an imported key is the proof snarkjs' own
algorithm (oracle/py/zkey.snarkjs_prove) makes for the same (r, s), byte for byte."""
import random
import struct

import numpy as np
import pytest

from oracle.py import fields, groth16 as og16, zkey as zo
from oracle.py.curve import g1_from_bytes, g1_to_bytes
from tests.r1cs_util import random_r1cs

R = fields.R


def _wit(z):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in z), dtype=np.uint8).reshape(-1, 32).copy()


def _toxic(seed):
    rnd = random.Random(seed)
    return tuple(rnd.randrange(1, R) for _ in range(5))


def _c_msm():
    """the oracle's d direct sums through the C restatement's Pippenger (fast enough for d = 1024)"""
    from oracle.c import binding as oc

    def msm(scalars, points):
        pts = np.frombuffer(b"".join(g1_to_bytes(p) for p in points), dtype=np.uint8).reshape(-1, 64)
        sc = np.frombuffer(b"".join(int(s % R).to_bytes(32, "little") for s in scalars), dtype=np.uint8).reshape(-1, 32)
        return g1_from_bytes(oc.msm_g1(pts.copy(), sc.copy()).tobytes())
    return msm


def _second_witness(n_wires, cons, z0, seed):
    z, rnd = list(z0), random.Random(seed)
    for i in range(1, n_wires - len(cons)):
        z[i] = rnd.randrange(R)
    for k, (a, b, _c) in enumerate(cons):
        av = sum(v * z[i] for i, v in a.items()) % R
        bv = sum(v * z[i] for i, v in b.items()) % R
        z[n_wires - len(cons) + k] = av * bv % R
    return z


def case_import_matches_oracle_and_snarkjs_prover(ctx, n_constraints, n_pub, python_prover=True):
    """a key the way snarkjs lays one out (constraints in order at ffjavascript's roots, C only inside the points, odd-coset H):
    the import's blobs are the oracle's; proofs are snarkjs' (restated) and the C restatement's; og_verify judges them"""
    from oracle.c import binding as oc
    from owshen_amd import groth16 as g16, zkey as zk
    n_wires, cons, z0 = random_r1cs(n_constraints, n_pub, seed=n_constraints + 1000, bool_every=5)
    z = zo.snarkjs_setup(n_wires, n_pub, cons, *_toxic(n_constraints))
    data = zo.write_zkey(z)
    pk_blob, vk_blob = zk.import_zkey(ctx, data)
    want_pk, want_vk = zo.zkey_to_owshen(zo.read_zkey(data), msm=_c_msm())
    assert vk_blob == want_vk
    assert len(pk_blob) == len(want_pk)
    assert pk_blob[:80] == want_pk[:80] and struct.unpack("<10Q", pk_blob[:80])[8] == 1
    assert pk_blob == want_pk
    pk = g16.ProvingKey(ctx, pk_blob)
    ck = oc.prepared_key_from_blob(pk_blob)
    zs = [z0, _second_witness(n_wires, cons, z0, 5)]
    rnd = random.Random(17)
    rs = [(rnd.randrange(R), rnd.randrange(R)), (0, 0)]
    got = pk.prove_batch(np.stack([_wit(w) for w in zs]), rs)
    for w, (r, s), p in zip(zs, rs, got):
        assert p.tobytes() == ck.prove(_wit(w), r, s)
        if python_prover:
            assert p.tobytes() == og16.proof_to_bytes(zo.snarkjs_prove(z, w, r, s))
        assert g16.verify(vk_blob, _wit(w)[1:1 + n_pub], p.tobytes(), lib=ctx._lib) is True
    if n_pub:
        wrong = _wit(zs[0])[1:1 + n_pub].copy()
        wrong[0, 0] ^= 1
        assert g16.verify(vk_blob, wrong, got[0].tobytes(), lib=ctx._lib) is False
    # a witness that violates a constraint: the key has no C matrix to refuse it with -- the proof must simply not verify
    bad = list(zs[0])
    bad[-1] = (bad[-1] + 1) % R
    p_bad = pk.prove(_wit(bad), 3, 4)
    assert g16.verify(vk_blob, _wit(bad)[1:1 + n_pub], bytes(p_bad), lib=ctx._lib) is False
    # ... but wire 0 != 1 still is refused
    from owshen_amd.api import OwshenGpuError
    not_one = _wit(zs[0])
    not_one[0, 0] = 2
    with pytest.raises(OwshenGpuError):
        pk.prove(not_one, 3, 4)
    pk.close()


def case_own_key_through_a_zkey(ctx, n_constraints, n_pub):
    """og_setup's key for a circuit, and the SAME ceremony laid out as a zkey by the oracle (this library's row i at constraint
    i k mod d): the import must give og_setup's group elements back byte for byte -- the H query through the DFT over points --
    and the same proofs"""
    from owshen_amd import groth16 as g16, zkey as zk
    n_wires, cons, z0 = random_r1cs(n_constraints, n_pub, seed=n_constraints + 2000)
    toxic = _toxic(n_constraints + 7)
    blob, vk = g16.setup(ctx, g16.R1CS.from_constraints(n_wires, n_pub, cons), *toxic)
    log_d = struct.unpack("<10Q", blob[:80])[3]
    d, k = 1 << log_d, zo.constraint_of_row(log_d)
    z = zo.snarkjs_setup(n_wires, n_pub, cons, *toxic, row_map=[i * k % d for i in range(d)])
    assert z["domain_size"] == d
    pk2, vk2 = zk.import_zkey(ctx, zo.write_zkey(z))
    assert vk2 == g16.vk_to_bytes(vk)
    q_bytes = 64 * (2 * n_wires + (n_wires - n_pub - 1) + d - 1) + 128 * n_wires
    pad = lambda n: (n + 31) // 32 * 32                                                    # noqa: E731
    tail = pad(64 * n_wires) * 2 + pad(128 * n_wires) + pad(64 * (n_wires - n_pub - 1)) + pad(64 * (d - 1))
    assert tail >= q_bytes
    assert pk2[-tail:] == blob[-tail:]                   # the five queries
    assert pk2[80:80 + 512] == blob[80:80 + 512]         # alpha, beta, delta in both groups
    a, b = g16.ProvingKey(ctx, blob), g16.ProvingKey(ctx, pk2)
    rs = [(11, 12), (0, 5)]
    w = np.stack([_wit(z0), _wit(_second_witness(n_wires, cons, z0, 9))])
    assert a.prove_batch(w, rs).tobytes() == b.prove_batch(w, rs).tobytes()
    a.close()
    b.close()


def case_export_is_what_snarkjs_would_prove_with(ctx, n_constraints, n_pub, python_prover=True):
    """og_zkey_export of og_setup's key: the oracle's reader takes it (every point on its curve), snarkjs' prover restated makes
    THIS library's proofs with it, and importing it again gives the queries back"""
    from owshen_amd import groth16 as g16, zkey as zk
    n_wires, cons, z0 = random_r1cs(n_constraints, n_pub, seed=n_constraints + 3000)
    blob, vk = g16.setup(ctx, g16.R1CS.from_constraints(n_wires, n_pub, cons), *_toxic(n_constraints + 11))
    vkb = g16.vk_to_bytes(vk)
    data = zk.export_zkey(ctx, blob, vkb)
    z = zo.read_zkey(data)
    assert (z["n_vars"], z["n_public"]) == (n_wires, n_pub) and z.get("n_contributions") == 0
    pk = g16.ProvingKey(ctx, blob)
    r, s = 1234567, 7654321
    mine = bytes(pk.prove(_wit(z0), r, s))
    if python_prover:
        assert og16.proof_to_bytes(zo.snarkjs_prove(z, z0, r, s)) == mine
    pk2, vk2 = zk.import_zkey(ctx, data)
    assert vk2 == vkb
    d = z["domain_size"]
    nh = (64 * (d - 1) + 31) // 32 * 32
    assert pk2[-nh:] == blob[-nh:]
    again = g16.ProvingKey(ctx, pk2)
    assert bytes(again.prove(_wit(z0), r, s)) == mine
    assert g16.verify(vkb, _wit(z0)[1:1 + n_pub], mine, lib=ctx._lib) is True
    pk.close()
    again.close()


def case_wtns(lib):
    from owshen_amd import zkey as zk
    from owshen_amd.api import OwshenGpuError
    rnd = random.Random(3)
    w = [1] + [rnd.randrange(R) for _ in range(40)] + [R - 1, 0]
    data = zk.write_wtns(_wit(w), lib=lib)
    assert data == zo.write_wtns(w)
    assert zo.read_wtns(data) == w
    assert zk.read_wtns(zo.write_wtns(w), lib=lib).tobytes() == _wit(w).tobytes()
    assert zk.read_wtns(zo.write_wtns([]), lib=lib).shape == (0, 32)
    over = bytearray(data)
    over[-32:] = R.to_bytes(32, "little")                 # the last value = r: not canonical
    for broken in (data[:-1], b"wtnz" + data[4:], bytes(over), data[:40], b""):
        with pytest.raises(OwshenGpuError):
            zk.read_wtns(broken, lib=lib)
    with pytest.raises(OwshenGpuError):
        zk.write_wtns(np.frombuffer(R.to_bytes(32, "little"), dtype=np.uint8).reshape(1, 32), lib=lib)


def _sections(data):
    off, out = 12, []
    for _ in range(struct.unpack_from("<I", data, 8)[0]):
        sid, size = struct.unpack_from("<IQ", data, off)
        out.append((sid, off + 12, size))
        off += 12 + size
    return out


def case_refusals(ctx):
    """what a file from outside may hold: every malformed zkey is refused with OG_ERR_INVALID and a reason -- never a crash, never
    a key"""
    from owshen_amd import zkey as zk
    from owshen_amd.api import OwshenGpuError
    n_wires, cons, _z0 = random_r1cs(6, 1, seed=77)
    good = zo.write_zkey(zo.snarkjs_setup(n_wires, 1, cons, *_toxic(1)))
    zk.import_zkey(ctx, good)
    sec = {sid: (o, n) for sid, o, n in _sections(good)}

    def patched(off, new):
        b = bytearray(good)
        b[off:off + len(new)] = new
        return bytes(b)
    q = fields.P
    cases = {
        "magic": b"zkex" + good[4:],
        "version": patched(4, struct.pack("<I", 2)),
        "truncated": good[:len(good) - 5],
        "protocol": patched(sec[1][0], struct.pack("<I", 2)),
        "other base field": patched(sec[2][0] + 4, (q + 2).to_bytes(32, "little")),
        "other scalar field": patched(sec[2][0] + 40, (R + 2).to_bytes(32, "little")),
        "domain not a power of two": patched(sec[2][0] + 80, struct.pack("<I", 12)),
        "more public inputs than wires": patched(sec[2][0] + 76, struct.pack("<I", n_wires)),
        "coordinate >= q": patched(sec[5][0], q.to_bytes(32, "little")),
        "point off the curve": patched(sec[5][0] + 64, (1).to_bytes(32, "little")),
        "G2 point off the curve": patched(sec[7][0] + 128, (1).to_bytes(32, "little")),
        "H point off the curve": patched(sec[9][0] + 32, (5).to_bytes(32, "little")),
        "coefficient >= r": patched(sec[4][0] + 4 + 12, R.to_bytes(32, "little")),
        "coefficient matrix 2": patched(sec[4][0] + 4, struct.pack("<I", 2)),
        "coefficient constraint outside the domain": patched(sec[4][0] + 8, struct.pack("<I", 1 << 20)),
        "coefficient signal outside the wires": patched(sec[4][0] + 12, struct.pack("<I", n_wires)),
        "coefficient count": patched(sec[4][0], struct.pack("<I", 3)),
        "section length": patched(sec[6][0] - 8, struct.pack("<Q", sec[6][1] - 64)),
        "empty": b"",
    }
    # a file without its H section: drop section 9 and fix the count
    o9, n9 = sec[9]
    cases["section 9 missing"] = good[:8] + struct.pack("<I", 9) + good[12:o9 - 12] + good[o9 + n9:]
    for name, data in cases.items():
        with pytest.raises(OwshenGpuError) as e:
            zk.import_zkey(ctx, data)
        assert e.value.code == -1, (name, e.value)
        assert "og_zkey_import" in str(e.value), (name, str(e.value))
    # an export wants a verifying key that belongs to the proving key
    from owshen_amd import groth16 as g16
    blob, vk = g16.setup(ctx, g16.R1CS.from_constraints(n_wires, 1, cons), *_toxic(2))
    _b2, vk_other = g16.setup(ctx, g16.R1CS.from_constraints(n_wires, 1, cons), *_toxic(3))
    for pk_b, vk_b in ((blob, g16.vk_to_bytes(vk_other)), (blob[:-32], g16.vk_to_bytes(vk)), (b"OWPK0001", g16.vk_to_bytes(vk))):
        with pytest.raises(OwshenGpuError):
            zk.export_zkey(ctx, pk_b, vk_b)
    # and og_pk_load refuses header flags it does not know, or the C = A o B flag beside a C matrix
    for word, val in ((8, 2), (9, 1), (8, 1)):
        b = bytearray(blob)
        b[8 * word:8 * word + 8] = struct.pack("<Q", val)
        with pytest.raises(OwshenGpuError):
            g16.ProvingKey(ctx, bytes(b))


def case_prove_files(ctx, tmp_path):
    """the `snarkjs groth16 prove` flow on files: .zkey + .wtns in, snarkjs' three JSON files out; the second pairing engine and
    og_verify accept them, and for a given (r, s) the proof is snarkjs' (restated)"""
    import json
    from owshen_amd import groth16 as g16, snarkjs_json, zkey as zk
    n_pub = 2
    n_wires, cons, z0 = random_r1cs(9, n_pub, seed=4242)
    z = zo.snarkjs_setup(n_wires, n_pub, cons, *_toxic(4242))
    zdata, wdata = zo.write_zkey(z), zo.write_wtns(z0)
    proof, pub, vk = zk.prove_files(ctx, zdata, wdata, rs=(5, 6))
    assert proof == og16.proof_to_bytes(zo.snarkjs_prove(z, z0, 5, 6))
    assert pub.tobytes() == _wit(z0)[1:1 + n_pub].tobytes()
    proof2, _pub, _vk = zk.prove_files(ctx, zdata, wdata)          # fresh blinding: another proof of the same statement
    assert proof2 != proof and g16.verify(vk, pub, proof2, lib=ctx._lib) is True
    paths = snarkjs_json.write(str(tmp_path), vk, proof2, [bytes(x) for x in pub])
    vkj = json.load(open(paths["verification_key.json"]))
    assert vkj["nPublic"] == n_pub and [int(x) for x in vkj["vk_alpha_1"][:2]] == list(z["alpha1"])
    assert [int(x) for x in json.load(open(paths["public.json"]))] == z0[1:1 + n_pub]
    with pytest.raises(ValueError):
        zk.prove_files(ctx, zdata, zo.write_wtns(z0 + [1]))
    return paths


def case_r1cs(lib):
    """circom's .r1cs through og_r1cs_read / og_r1cs_write against the oracle's writer / reader"""
    from owshen_amd import zkey as zk
    from owshen_amd.api import OwshenGpuError
    from tests.r1cs_util import csr_from_rows
    n_pub = 3
    n_wires, cons, _z = random_r1cs(17, n_pub, seed=555)
    cons[3] = ({}, {2: 5}, {})                                  # an empty linear combination
    data = zo.write_r1cs(n_wires, n_pub, cons, n_outputs=1)
    r1 = zk.read_r1cs(data, lib=lib)
    assert (r1.n_wires, r1.n_pub, r1.n_constraints) == (n_wires, n_pub, len(cons))
    for k, mat in enumerate((r1.a, r1.b, r1.c)):
        ptr, col, val = csr_from_rows([c[k] for c in cons])
        nc = len(cons)
        assert mat.ptr[:nc + 1].tobytes() == ptr.tobytes() and mat.col[:ptr[-1]].tobytes() == col.tobytes()
        assert mat.val[:ptr[-1]].tobytes() == val.tobytes()
    back = zo.read_r1cs(zk.write_r1cs(r1, lib=lib))
    norm = lambda rows: [tuple({w: v % R for w, v in r.items() if v % R} for r in c) for c in rows]      # noqa: E731
    assert back[0] == n_wires and back[1] == n_pub and norm(back[2]) == norm(cons)
    sec2 = data.index(struct.pack("<IQ", 2, len(data) - 12 - 12 - 64 - 12 - 12 - 8 * n_wires))
    bad_wire = bytearray(data)
    struct.pack_into("<I", bad_wire, sec2 + 12 + 4, n_wires)                                     # the first entry's wire
    bad_val = bytearray(data)
    bad_val[sec2 + 12 + 8:sec2 + 12 + 40] = R.to_bytes(32, "little")                             # ... its coefficient
    for broken in (b"", data[:30], b"r1cx" + data[4:], data[:-8 * n_wires - 20], bytes(bad_wire), bytes(bad_val),
                   data[:16] + struct.pack("<I", 31) + data[20:]):
        with pytest.raises(OwshenGpuError) as e:
            zk.read_r1cs(broken, lib=lib)
        assert e.value.code == -1 and "og_r1cs_read" in str(e.value)


def case_import_with_r1cs(ctx, n_constraints, n_pub):
    """a .zkey beside its .r1cs: the imported key carries the C matrix (flag 0), equals the oracle's, proves what snarkjs proves, and
    REFUSES a witness that violates a constraint; another circuit's .r1cs is refused"""
    from oracle.c import binding as oc
    from owshen_amd import groth16 as g16, zkey as zk
    from owshen_amd.api import OwshenGpuError
    n_wires, cons, z0 = random_r1cs(n_constraints, n_pub, seed=n_constraints + 4000)
    z = zo.snarkjs_setup(n_wires, n_pub, cons, *_toxic(n_constraints + 1))
    zdata, rdata = zo.write_zkey(z), zo.write_r1cs(n_wires, n_pub, cons)
    pk_blob, vk_blob = zk.import_zkey(ctx, zdata, rdata)
    want_pk, want_vk = zo.zkey_to_owshen(zo.read_zkey(zdata), msm=_c_msm(), constraints=zo.read_r1cs(rdata)[2])
    assert struct.unpack("<10Q", pk_blob[:80])[7:9] == (sum(1 for c in cons for v in c[2].values() if v % R), 0)
    assert pk_blob == want_pk and vk_blob == want_vk
    pk = g16.ProvingKey(ctx, pk_blob)
    p = bytes(pk.prove(_wit(z0), 21, 22))
    assert p == og16.proof_to_bytes(zo.snarkjs_prove(z, z0, 21, 22)) == oc.prepared_key_from_blob(pk_blob).prove(_wit(z0), 21, 22)
    assert g16.verify(vk_blob, _wit(z0)[1:1 + n_pub], p, lib=ctx._lib) is True
    bad = list(z0)
    bad[-1] = (bad[-1] + 1) % R
    with pytest.raises(OwshenGpuError) as e:
        pk.prove(_wit(bad), 3, 4)
    assert e.value.code == -4
    pk.close()
    other = [(dict(a), dict(b), dict(c)) for a, b, c in cons]
    w0 = next(iter(other[1][1]))
    other[1][1][w0] = (other[1][1][w0] + 1) % R                     # one coefficient of B differs
    with pytest.raises(OwshenGpuError) as e:
        zk.import_zkey(ctx, zdata, zo.write_r1cs(n_wires, n_pub, other))
    assert e.value.code == -1 and "matrix B" in str(e.value)
    with pytest.raises(OwshenGpuError):
        zk.import_zkey(ctx, zdata, zo.write_r1cs(n_wires + 1, n_pub, cons))
    # the same key without its .r1cs keeps proving the same bytes
    pk1, _vk1 = zk.import_zkey(ctx, zdata)
    k1 = g16.ProvingKey(ctx, pk1)
    assert bytes(k1.prove(_wit(z0), 21, 22)) == p
    k1.close()
