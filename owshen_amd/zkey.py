"""Host-side mirror of the snarkjs-file entry points of include/owshen_gpu.h (og_zkey_import / og_zkey_export / og_wtns_read /
og_wtns_write, owshen_amd/csrc/zkey.hip): a `.zkey` becomes this library's proving / verifying key blobs, a `.wtns` a witness.
No reference counterpart; the lineage BASELINE.json's north_star names proved with circom / snarkjs.  The
formats are restated, with their upstream sources, in oracle/py/zkey.py -- which this module does not import: every
conversion happens inside the library.

    pk_blob, vk_blob = zkey.import_zkey(ctx, open("circuit_final.zkey", "rb").read())
    witness = zkey.read_wtns(open("witness.wtns", "rb").read(), lib=ctx._lib)            # np.uint8 [n_wires, 32]
    pk = groth16.ProvingKey(ctx, pk_blob);  proof = pk.prove(witness, r, s);  groth16.verify(vk_blob, witness[1:1 + n_pub], proof)
"""
import ctypes as C

import numpy as np


def _take(lib, ptr, n):
    try:
        return C.string_at(ptr, n.value)
    finally:
        lib.og_blob_free(ptr)


def _r1cs_handle(lib, data):
    data = bytes(data)
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    h = C.c_void_p()
    rc = lib.og_r1cs_read(buf, len(data), C.byref(h))
    if rc:
        _raise(lib, rc)
    return h


def import_zkey(ctx, data, r1cs=None):
    """.zkey bytes -> (OWPK0001 bytes, OWVK0001 bytes).  r1cs: the bytes of the circuit's .r1cs -- the key then carries its C
    matrix and refuses an unsatisfying witness; without it the key has the "C z = A z o B z" flag, as snarkjs proves."""
    lib = ctx._lib
    data = bytes(data)
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    pk_p, vk_p, pk_n, vk_n = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()
    rh = _r1cs_handle(lib, r1cs) if r1cs is not None else None
    try:
        ctx._pre()
        ctx._check(lib.og_zkey_import(ctx._h, buf, len(data), rh, C.byref(pk_p), C.byref(pk_n), C.byref(vk_p), C.byref(vk_n)))
    finally:
        if rh is not None:
            lib.og_r1cs_free(rh)
    pk = _take(lib, pk_p, pk_n)
    return pk, _take(lib, vk_p, vk_n)


def read_r1cs(data, lib=None):
    """circom's .r1cs bytes -> owshen_amd.groth16.R1CS (constraint rows; og_setup makes a test key for it)"""
    from . import groth16
    lib = _lib_or_default(lib)
    h = _r1cs_handle(lib, data)
    try:
        info = (C.c_uint64 * 6)()
        rc = lib.og_r1cs_info(h, info)
        if rc:
            _raise(lib, rc)
        n_wires, n_pub, nc = int(info[0]), int(info[1]), int(info[2])
        mats = []
        for k in range(3):
            nnz = int(info[3 + k])
            ptr, col, val = np.zeros(nc + 1, np.uint32), np.zeros(nnz, np.uint32), np.zeros((nnz, 32), np.uint8)
            rc = lib.og_r1cs_export(h, k, C.c_void_p(ptr.ctypes.data), C.c_void_p(col.ctypes.data), C.c_void_p(val.ctypes.data))
            if rc:
                _raise(lib, rc)
            mats.append(groth16.SparseMatrix(ptr, col, val, n_wires))
        return groth16.R1CS(n_wires, n_pub, *mats)
    finally:
        lib.og_r1cs_free(h)


def write_r1cs(r1cs, lib=None):
    """owshen_amd.groth16.R1CS -> .r1cs bytes (`snarkjs r1cs info`, `snarkjs groth16 setup`)"""
    from . import groth16
    lib = _lib_or_default(lib)
    rc, h, _keep = groth16._r1cs_handle(lib, r1cs)
    if rc:
        _raise(lib, rc)
    try:
        p, n = C.c_void_p(), C.c_size_t()
        rc = lib.og_r1cs_write(h, C.byref(p), C.byref(n))
        if rc:
            _raise(lib, rc)
        return _take(lib, p, n)
    finally:
        lib.og_r1cs_free(h)


def export_zkey(ctx, pk_blob, vk_blob):
    """(OWPK0001, OWVK0001) -> .zkey bytes `snarkjs groth16 prove` accepts"""
    lib = ctx._lib
    pk = (C.c_uint8 * len(pk_blob)).from_buffer_copy(pk_blob)
    vk = (C.c_uint8 * len(vk_blob)).from_buffer_copy(vk_blob)
    z_p, z_n = C.c_void_p(), C.c_size_t()
    ctx._pre()
    ctx._check(lib.og_zkey_export(ctx._h, pk, len(pk_blob), vk, len(vk_blob), C.byref(z_p), C.byref(z_n)))
    return _take(lib, z_p, z_n)


def _lib_or_default(lib):
    if lib is not None:
        return lib
    from ._lib import lib as default
    return default


def _raise(lib, code):
    from ._lib import OwshenGpuError
    raise OwshenGpuError(code, lib.og_last_error().decode("utf-8", "replace"))


def read_wtns(data, lib=None):
    """.wtns bytes -> np.uint8 [n, 32] canonical little-endian values (host only)"""
    lib = _lib_or_default(lib)
    data = bytes(data)
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    n = C.c_uint64()
    rc = lib.og_wtns_read(buf, len(data), None, 0, C.byref(n))
    if rc:
        _raise(lib, rc)
    out = np.zeros((n.value, 32), dtype=np.uint8)
    rc = lib.og_wtns_read(buf, len(data), C.c_void_p(out.ctypes.data), n.value, C.byref(n))
    if rc:
        _raise(lib, rc)
    return out


def write_wtns(values, lib=None):
    """np.uint8 [n, 32] canonical values -> .wtns bytes"""
    lib = _lib_or_default(lib)
    v = np.ascontiguousarray(values, dtype=np.uint8).reshape(-1, 32)
    p, n = C.c_void_p(), C.c_size_t()
    rc = lib.og_wtns_write(C.c_void_p(v.ctypes.data), v.shape[0], C.byref(p), C.byref(n))
    if rc:
        _raise(lib, rc)
    return _take(lib, p, n)


def prove_files(ctx, zkey_bytes, wtns_bytes, rs=None, r1cs_bytes=None):
    """what `snarkjs groth16 prove circuit.zkey witness.wtns proof.json public.json` computes, on the GPU: -> (proof 256 B,
    public inputs np.uint8 [n_pub, 32], OWVK0001 bytes).  (r, s): two ints below the group order, or None = fresh randomness."""
    import secrets
    from . import groth16
    from .api import FR_MODULUS
    pk_blob, vk_blob = import_zkey(ctx, zkey_bytes, r1cs_bytes)
    w = read_wtns(wtns_bytes, lib=ctx._lib)
    pk = groth16.ProvingKey(ctx, pk_blob)
    try:
        if w.shape[0] != pk.n_wires:
            raise ValueError(f"the witness holds {w.shape[0]} values, the key {pk.n_wires} wires")
        r, s = rs if rs is not None else (secrets.randbelow(FR_MODULUS), secrets.randbelow(FR_MODULUS))
        proof = bytes(pk.prove(w, r, s))
        return proof, w[1:1 + pk.n_pub].copy(), vk_blob
    finally:
        pk.close()


def main(argv=None):
    import argparse
    import os
    ap = argparse.ArgumentParser(description="snarkjs' .zkey / .wtns files and this library's keys (include/owshen_gpu.h: og_zkey_*, og_wtns_*)")
    sub = ap.add_subparsers(dest="cmd", required=True)
    p = sub.add_parser("import", help=".zkey -> OWPK0001 / OWVK0001 blobs")
    p.add_argument("zkey"), p.add_argument("pk_out"), p.add_argument("vk_out")
    p.add_argument("--r1cs", help="the circuit's .r1cs: the key then carries the C matrix and refuses an unsatisfying witness")
    p = sub.add_parser("export", help="OWPK0001 + OWVK0001 -> a .zkey `snarkjs groth16 prove` accepts")
    p.add_argument("pk"), p.add_argument("vk"), p.add_argument("zkey_out")
    p = sub.add_parser("prove", help="`snarkjs groth16 prove` on the GPU: .zkey + .wtns -> proof.json, public.json, verification_key.json")
    p.add_argument("zkey"), p.add_argument("wtns"), p.add_argument("outdir")
    p.add_argument("--r1cs", help="the circuit's .r1cs (a witness that violates a constraint is then refused instead of proved)")
    p = sub.add_parser("wtns2bin", help=".wtns -> n x 32 B little-endian values")
    p.add_argument("wtns"), p.add_argument("out")
    p = sub.add_parser("bin2wtns", help="n x 32 B little-endian values -> .wtns")
    p.add_argument("values"), p.add_argument("out")
    a = ap.parse_args(argv)

    def rd(path):
        with open(path, "rb") as f:
            return f.read()

    def wr(path, data):
        with open(path, "wb") as f:
            f.write(data)
    if a.cmd == "wtns2bin":
        return wr(a.out, read_wtns(rd(a.wtns)).tobytes())
    if a.cmd == "bin2wtns":
        return wr(a.out, write_wtns(np.frombuffer(rd(a.values), dtype=np.uint8).reshape(-1, 32)))
    from .api import Context
    ctx = Context(0)
    try:
        if a.cmd == "import":
            pk, vk = import_zkey(ctx, rd(a.zkey), rd(a.r1cs) if a.r1cs else None)
            wr(a.pk_out, pk), wr(a.vk_out, vk)
        elif a.cmd == "export":
            wr(a.zkey_out, export_zkey(ctx, rd(a.pk), rd(a.vk)))
        else:
            from . import snarkjs_json
            proof, pub, vk = prove_files(ctx, rd(a.zkey), rd(a.wtns), r1cs_bytes=rd(a.r1cs) if a.r1cs else None)
            for path in snarkjs_json.write(a.outdir, vk, proof, [bytes(x) for x in pub]).values():
                print(os.path.abspath(path))
    finally:
        ctx.close()


if __name__ == "__main__":
    main()
