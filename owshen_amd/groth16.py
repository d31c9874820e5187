"""Groth16 host side: R1CS container, key generation from explicit toxic waste, key
serialization ("OWPK0001", include/owshen_gpu.h) and the prove() surface over the C ABI.

The reference snapshot has no prover; the surface is shaped for its withdraw
seam -- ``withdraw_handler`` (/root/reference/src/services/api_services/withdraw.rs:27-71)
would call ``prove`` and ``burn_tx`` (/root/reference/src/blockchain/tx/burn_tx.rs:11-32)
would call ``verify`` -- and field elements use the byte format of the reference's
``Fp::to_repr()`` (/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11).

All curve / field arithmetic happens on the GPU through ``api.Context``; this module only
moves bytes, builds sparse-matrix index arrays with numpy, and does a handful of scalar
modular inversions with Python integers.
"""
import ctypes as C
import struct

import numpy as np

from . import api
from .api import FR, FR_MODULUS as R

G1_GEN = (1, 2)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))
PK_MAGIC = b"OWPK0001"
VK_MAGIC = b"OWVK0001"


def _le(v):
    return int(v).to_bytes(32, "little")


G1_GEN_BYTES = _le(G1_GEN[0]) + _le(G1_GEN[1])
G2_GEN_BYTES = _le(G2_GEN[0][0]) + _le(G2_GEN[0][1]) + _le(G2_GEN[1][0]) + _le(G2_GEN[1][1])


def _pad32(b):
    return b + b"\0" * (-len(b) % 32)


class SparseMatrix:
    """CSR over Fr: ptr uint32 [n_rows+1], col uint32 [nnz], val uint8 [nnz, 32] canonical LE."""

    def __init__(self, ptr, col, val, n_cols):
        self.ptr = np.ascontiguousarray(ptr, dtype=np.uint32)
        self.col = np.ascontiguousarray(col, dtype=np.uint32)
        self.val = np.ascontiguousarray(val, dtype=np.uint8).reshape(-1, 32)
        self.n_cols = n_cols
        assert self.ptr[0] == 0 and self.ptr[-1] == self.col.shape[0] == self.val.shape[0]

    @property
    def n_rows(self):
        return self.ptr.shape[0] - 1

    @property
    def nnz(self):
        return self.col.shape[0]

    @classmethod
    def from_rows(cls, rows, n_cols):
        """rows: list of {col: int coefficient}"""
        ptr = np.zeros(len(rows) + 1, dtype=np.uint32)
        cols, vals = [], []
        for i, r in enumerate(rows):
            for c in sorted(r):
                v = r[c] % R
                if v:
                    cols.append(c)
                    vals.append(_le(v))
            ptr[i + 1] = len(cols)
        val = np.frombuffer(b"".join(vals), dtype=np.uint8).reshape(-1, 32) if vals else np.zeros((0, 32), np.uint8)
        return cls(ptr, np.array(cols, dtype=np.uint32), val, n_cols)

    def transpose(self):
        """CSR of the transpose (host index shuffling only)."""
        rows = np.repeat(np.arange(self.n_rows, dtype=np.uint32), np.diff(self.ptr.astype(np.int64)))
        order = np.argsort(self.col, kind="stable")
        ptr = np.zeros(self.n_cols + 1, dtype=np.int64)
        np.add.at(ptr, self.col.astype(np.int64) + 1, 1)
        ptr = np.cumsum(ptr).astype(np.uint32)
        return SparseMatrix(ptr, rows[order], self.val[order], self.n_rows)


class R1CS:
    """A z o B z = C z over Fr.  Wire 0 is the constant 1, wires 1..n_pub are public.

    The QAP rows are the constraints followed by the n_pub + 1 input-consistency rows
    (A = wire i, B = C = 0), the arkworks convention the oracle also follows."""

    def __init__(self, n_wires, n_pub, a, b, c):
        """a, b, c: SparseMatrix over the constraint rows (n_constraints x n_wires)."""
        assert a.n_rows == b.n_rows == c.n_rows
        self.n_wires, self.n_pub, self.n_constraints = n_wires, n_pub, a.n_rows
        self.a, self.b, self.c = self._with_consistency(a, True), self._with_consistency(b, False), self._with_consistency(c, False)
        need = self.n_rows
        self.log_d = max(1, (need - 1).bit_length())

    def _with_consistency(self, mat, identity):
        extra = self.n_pub + 1
        if identity:
            one = np.frombuffer(_le(1), dtype=np.uint8)
            ptr = np.concatenate([mat.ptr, mat.ptr[-1] + np.arange(1, extra + 1, dtype=np.uint32)])
            col = np.concatenate([mat.col, np.arange(extra, dtype=np.uint32)])
            val = np.concatenate([mat.val, np.tile(one, (extra, 1))])
        else:
            ptr = np.concatenate([mat.ptr, np.full(extra, mat.ptr[-1], dtype=np.uint32)])
            col, val = mat.col, mat.val
        return SparseMatrix(ptr, col, val, self.n_wires)

    @property
    def n_rows(self):
        return self.n_constraints + self.n_pub + 1

    @property
    def domain_size(self):
        return 1 << self.log_d

    @classmethod
    def from_constraints(cls, n_wires, n_pub, constraints):
        """constraints: list of (a_row, b_row, c_row) dicts {wire: coeff}."""
        return cls(n_wires, n_pub,
                   SparseMatrix.from_rows([x[0] for x in constraints], n_wires),
                   SparseMatrix.from_rows([x[1] for x in constraints], n_wires),
                   SparseMatrix.from_rows([x[2] for x in constraints], n_wires))


def _r1cs_handle(lib, r1cs):
    """R1CS (constraint rows only) -> og_r1cs handle (og_r1cs_from_csr)"""
    nc = r1cs.n_constraints
    keep = []
    ptrs, cols, vals = (C.c_void_p * 3)(), (C.c_void_p * 3)(), (C.c_void_p * 3)()
    for k, mat in enumerate((r1cs.a, r1cs.b, r1cs.c)):
        ptr = np.ascontiguousarray(mat.ptr[: nc + 1])
        nnz = int(ptr[nc])
        col = np.ascontiguousarray(mat.col[:nnz])
        val = np.ascontiguousarray(mat.val[:nnz])
        keep += [ptr, col, val]
        ptrs[k], cols[k], vals[k] = ptr.ctypes.data, col.ctypes.data, val.ctypes.data
    h = C.c_void_p()
    rc = lib.og_r1cs_from_csr(r1cs.n_wires, r1cs.n_pub, nc, ptrs, cols, vals, C.byref(h))
    return rc, h, keep


def vk_from_bytes(blob):
    """"OWVK0001" blob -> dict (alpha_g1, beta_g2, gamma_g2, delta_g2: bytes; ic: np.uint8 [n_pub + 1, 64])"""
    assert blob[:8] == VK_MAGIC
    n_pub = struct.unpack("<Q", blob[8:16])[0]
    o = 16
    out = {"alpha_g1": blob[o:o + 64], "beta_g2": blob[o + 64:o + 192], "gamma_g2": blob[o + 192:o + 320], "delta_g2": blob[o + 320:o + 448]}
    out["ic"] = np.frombuffer(blob[o + 448:], dtype=np.uint8).reshape(n_pub + 1, 64).copy()
    return out


def setup(ctx, r1cs, tau, alpha, beta, gamma, delta):
    """Key generation from explicit toxic waste (tests / benchmarks; a production key comes from a ceremony), run by the
    library (og_setup: Lagrange evaluations, transposed sparse products, query scalars and the fixed-base multiplications on
    the GPU).  Returns (proving key blob: bytes, verifying key: dict of bytes / np arrays)."""
    for v in (tau, alpha, beta, gamma, delta):
        assert 0 < v < R
    lib = ctx._lib
    rc, h, _keep = _r1cs_handle(lib, r1cs)
    ctx._check(rc)
    try:
        toxic = (C.c_uint8 * 160).from_buffer_copy(b"".join(_le(v) for v in (tau, alpha, beta, gamma, delta)))
        pk_p, vk_p, pk_n, vk_n = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()
        ctx._pre()
        ctx._check(lib.og_setup(ctx._h, h, toxic, C.byref(pk_p), C.byref(pk_n), C.byref(vk_p), C.byref(vk_n)))
        try:
            pk = C.string_at(pk_p, pk_n.value)
            vk = C.string_at(vk_p, vk_n.value)
        finally:
            lib.og_blob_free(pk_p)
            lib.og_blob_free(vk_p)
    finally:
        lib.og_r1cs_free(h)
    return pk, vk_from_bytes(vk)


class ProvingKey:
    """Device-resident proving key (og_pk)."""

    def __init__(self, ctx, blob):
        self.ctx = ctx
        h = C.c_void_p()
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        ctx._check(ctx._lib.og_pk_load(ctx._h, buf, len(blob), C.byref(h)))
        self._h = h
        info = (C.c_uint64 * 4)()
        ctx._check(ctx._lib.og_pk_info(h, info))
        self.n_wires, self.n_pub, self.log_d, self.n_rows = (int(x) for x in info)

    def density(self):
        """points actually accumulated per proof by the A, B (once in G1, once in G2), L and H queries (og_pk_density)"""
        d = (C.c_uint64 * 4)()
        self.ctx._check(self.ctx._lib.og_pk_density(self._h, d))
        return dict(zip(("a", "b", "l", "h"), (int(x) for x in d)))

    def hbm_bytes(self):
        """bytes of HBM the resident key occupies (og_pk_bytes)"""
        out = C.c_uint64()
        self.ctx._check(self.ctx._lib.og_pk_bytes(self._h, C.byref(out)))
        return int(out.value)

    def plan(self, n):
        """(mode, [sub-batch sizes]) a call of n proofs would run as right now (og_prove_plan); mode is one of 'serial',
        'query fan-out', 'symmetric lanes', 'stage pipeline'."""
        sizes, cnt, mode = (C.c_uint32 * 4096)(), C.c_size_t(), C.c_int()
        self.ctx._check(self.ctx._lib.og_prove_plan(self.ctx._h, self._h, n, sizes, 4096, C.byref(cnt), C.byref(mode)))
        return ("serial", "query fan-out", "symmetric lanes", "stage pipeline")[mode.value], [int(sizes[k]) for k in range(min(cnt.value, 4096))]

    def windows(self):
        """window bits of the A, B, L and H queries' tables (og_pk_windows): n x ceil(255 / bits) additions per proof each"""
        d = (C.c_uint64 * 4)()
        self.ctx._check(self.ctx._lib.og_pk_windows(self._h, d))
        return dict(zip(("a", "b", "l", "h"), (int(x) for x in d)))

    @staticmethod
    def _rs_bytes(rs):
        """rs: list of (r, s) int pairs or uint8 array [n, 64]."""
        if isinstance(rs, np.ndarray):
            return np.ascontiguousarray(rs, dtype=np.uint8).reshape(-1, 64)
        return np.frombuffer(b"".join(_le(r) + _le(s) for r, s in rs), dtype=np.uint8).reshape(-1, 64).copy()

    def prove(self, witness, r, s):
        """witness: np.uint8 [n_wires, 32] (host) -> 256-byte proof."""
        return self.prove_batch(np.ascontiguousarray(witness, dtype=np.uint8)[None], [(r, s)])[0].tobytes()

    def prove_batch(self, witnesses, rs):
        """witnesses: np.uint8 [n, n_wires, 32] on the HOST -> np.uint8 [n, 256]."""
        w = np.ascontiguousarray(witnesses, dtype=np.uint8)
        n = w.shape[0]
        assert w.shape[1:] == (self.n_wires, 32)
        rsb = self._rs_bytes(rs)
        assert rsb.shape[0] == n
        out = np.zeros((n, 256), dtype=np.uint8)
        ctx = self.ctx
        ctx._check(ctx._lib.og_prove_batch(ctx._h, self._h, w.ctypes.data_as(C.c_void_p), n,
                                           rsb.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p)))
        return out

    def prove_batch_device(self, witnesses_d, rs):
        """witnesses_d: DEVICE uint8 [n, n_wires, 32] -> np.uint8 [n, 256]."""
        n = witnesses_d.shape[0]
        assert tuple(witnesses_d.shape[1:]) == (self.n_wires, 32)
        rsb = self._rs_bytes(rs)
        assert rsb.shape[0] == n
        out = np.zeros((n, 256), dtype=np.uint8)
        ctx = self.ctx
        ctx._pre()
        ctx._check(ctx._lib.og_prove_batch_d(ctx._h, self._h, ctx.ptr(witnesses_d), n, rsb.ctypes.data_as(C.c_void_p),
                                             out.ctypes.data_as(C.c_void_p)))
        return out

    # -- window-sharded proving (include/owshen_gpu.h: og_prove_partials_d / og_prove_from_partials_d; owshen_amd/shard.py) --
    PARTIAL_BYTES = 768   # OG_PARTIAL_BYTES: per proof and rank, the five queries' partial points

    def prove_partials_device(self, witnesses_d, win_rank, win_world):
        """the front half on this rank: witnesses_d DEVICE uint8 [n, n_wires, 32] -> DEVICE uint8 [n * 768], this rank's
        partial points of the five queries (windows k = win_rank mod win_world)"""
        n = witnesses_d.shape[0]
        assert tuple(witnesses_d.shape[1:]) == (self.n_wires, 32)
        ctx = self.ctx
        part = ctx.empty(n * self.PARTIAL_BYTES)
        ctx._pre()
        ctx._check(ctx._lib.og_prove_partials_d(ctx._h, self._h, ctx.ptr(witnesses_d), n, win_rank, win_world, ctx.ptr(part)))
        return part

    def prove_from_partials(self, gathered_d, world, rs):
        """the back half: gathered_d DEVICE uint8 [world * n * 768] (the ranks' partials, rank-major) + blinding -> np.uint8 [n, 256]"""
        rsb = self._rs_bytes(rs)
        n = rsb.shape[0]
        assert int(np.prod(gathered_d.shape)) == world * n * self.PARTIAL_BYTES
        out = np.zeros((n, 256), dtype=np.uint8)
        ctx = self.ctx
        ctx._pre()
        ctx._check(ctx._lib.og_prove_from_partials_d(ctx._h, self._h, ctx.ptr(gathered_d), world, n, rsb.ctypes.data_as(C.c_void_p),
                                                     out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._lib.og_pk_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass



def vk_to_bytes(vk):
    """verifying key dict (as returned by `setup`) -> "OWVK0001" blob (include/owshen_gpu.h, og_verify)"""
    ic = np.ascontiguousarray(vk["ic"], dtype=np.uint8).reshape(-1, 64)
    return (VK_MAGIC + struct.pack("<Q", ic.shape[0] - 1) + bytes(vk["alpha_g1"]) + bytes(vk["beta_g2"]) + bytes(vk["gamma_g2"]) +
            bytes(vk["delta_g2"]) + ic.tobytes())


def verify(vk_blob, public_inputs, proof, lib=None):
    """Groth16 verification on the CPU (og_verify; needs no GPU and no Context).
    public_inputs: ints or a uint8 array [n_pub, 32]; proof: 256 bytes.  Returns True / False."""
    if lib is None:
        from ._lib import lib
    if isinstance(public_inputs, np.ndarray):
        pub = np.ascontiguousarray(public_inputs, dtype=np.uint8).reshape(-1, 32)
    else:
        pub = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in public_inputs), dtype=np.uint8).reshape(-1, 32).copy()
    ok = C.c_int(0)
    vkb = (C.c_uint8 * len(vk_blob)).from_buffer_copy(bytes(vk_blob))
    pf = (C.c_uint8 * 256).from_buffer_copy(bytes(proof))
    rc = lib.og_verify(vkb, len(vk_blob), pub.ctypes.data_as(C.c_void_p), pub.shape[0], pf, C.byref(ok))
    if rc != 0:
        raise api.OwshenGpuError(rc, lib.og_last_error().decode("utf-8", "replace"))
    return bool(ok.value)


from .evm import proof_to_evm_calldata  # noqa: E402,F401  (256-byte proof -> the verifier contract's 8 words; owshen_amd/evm.py)
