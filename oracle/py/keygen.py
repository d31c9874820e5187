"""Groth16 key generation on the HOST, emitting the product's own key formats.  TEST INFRASTRUCTURE ONLY.

`og_setup` (owshen_amd/csrc/keygen.hip) needs a GPU.  BASELINE.json configs[0] -- "single withdraw proof on the CPU prover,
verified, plumbing, no GPU" -- needs a key without one: the key-generation scalars come from the Python big-integer oracle
(oracle/py/groth16.setup_scalars: Lagrange basis at tau, the QAP polynomials per wire), the ~140 k fixed-base multiplications
from the C restatement (oracle/c: oc_fixed_base_g1 / _g2, threaded), and the bytes are the "OWPK0001" / "OWVK0001" layouts of
include/owshen_gpu.h -- so the SAME blob loads into the C restatement (oracle.c.binding.prepared_key_from_blob), into
`og_pk_load`, and the verifying key into `og_verify` / libowshen_verify.so.  It must be og_setup's key for the same toxic waste --
every group element byte for byte, the matrices row for row (tests/test_plumbing.py: on the interpreter at depth 2, on the GPU
at depth 32): two key generators, one format."""
import struct

import numpy as np

from . import groth16 as og16
from .fields import R
from .curve import G1_GEN, G2_GEN, g1_to_bytes, g2_to_bytes


def _le(v):
    return int(v).to_bytes(32, "little")


def _pad32(b):
    return b + b"\0" * (-len(b) % 32)


def _scalars(vals):
    return np.frombuffer(b"".join(_le(v % R) for v in vals), dtype=np.uint8).reshape(-1, 32).copy() if len(vals) else np.zeros((0, 32), np.uint8)


def _csr(rows, which):
    """QAP rows (constraints + input-consistency rows) -> (ptr u32, col u32, val bytes) with the columns of a row ascending and
    zero coefficients dropped: owshen_amd.groth16.SparseMatrix.from_rows's convention"""
    ptr, col, val = [0], [], []
    for row in rows:
        r = row[which]
        for c in sorted(r):
            v = r[c] % R
            if v:
                col.append(c)
                val.append(_le(v))
        ptr.append(len(col))
    return np.array(ptr, dtype=np.uint32).tobytes(), np.array(col, dtype=np.uint32).tobytes(), b"".join(val), len(col)


def setup_blobs(n_wires, n_pub, constraints, tau, alpha, beta, gamma, delta, threads=None):
    """-> (proving key "OWPK0001" bytes, verifying key "OWVK0001" bytes) for the R1CS `constraints` (list of (a, b, c) row dicts)"""
    from ..c import binding as oc
    for v in (tau, alpha, beta, gamma, delta):
        assert 0 < v < R
    r1 = og16.R1CS(n_wires, n_pub, constraints)
    sc = og16.setup_scalars(r1, tau, alpha, beta, gamma, delta)
    d = r1.domain_size
    n_rows = len(constraints) + n_pub + 1
    log_d = max(1, (n_rows - 1).bit_length())
    assert d == 1 << log_d or (n_rows == 1 and d == 1)
    g1 = np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8)
    g2 = np.frombuffer(g2_to_bytes(G2_GEN), dtype=np.uint8)
    # every G1 multiple in one threaded call, every G2 multiple in another
    k1 = [alpha, beta, delta] + sc["a"] + sc["b"] + sc["l"] + sc["h"][:(1 << log_d) - 1] + sc["ic"]
    p1 = oc.fixed_base_g1(g1, _scalars(k1), threads)
    p2 = oc.fixed_base_g2(g2, _scalars([beta, delta, gamma] + sc["b"]), threads)
    m, nl, nh = n_wires, n_wires - n_pub - 1, (1 << log_d) - 1
    o = 3
    a_q, o = p1[o:o + m], o + m
    b1_q, o = p1[o:o + m], o + m
    l_q, o = p1[o:o + nl], o + nl
    h_q, o = p1[o:o + nh], o + nh
    ic = p1[o:o + n_pub + 1]
    assert o + n_pub + 1 == p1.shape[0]
    rows = r1.rows()
    mats = [_csr(rows, k) for k in range(3)]
    pk = b"OWPK0001" + struct.pack("<9Q", n_wires, n_pub, log_d, n_rows, mats[0][3], mats[1][3], mats[2][3], 0, 0)
    pk += p1[0].tobytes() + p1[1].tobytes() + p1[2].tobytes() + b"\0" * 64 + p2[0].tobytes() + p2[1].tobytes()
    for ptr, col, val, _n in mats:
        pk += _pad32(ptr) + _pad32(col) + _pad32(val)
    pk += _pad32(a_q.tobytes()) + _pad32(b1_q.tobytes()) + _pad32(p2[3:].tobytes()) + _pad32(l_q.tobytes()) + _pad32(h_q.tobytes())
    vk = b"OWVK0001" + struct.pack("<Q", n_pub) + p1[0].tobytes() + p2[0].tobytes() + p2[2].tobytes() + p2[1].tobytes() + ic.tobytes()
    return pk, vk
