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


def import_zkey(ctx, data):
    """.zkey bytes -> (OWPK0001 bytes with the "C z = A z o B z" flag, OWVK0001 bytes)"""
    lib = ctx._lib
    data = bytes(data)
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data or b"\0")
    pk_p, vk_p, pk_n, vk_n = C.c_void_p(), C.c_void_p(), C.c_size_t(), C.c_size_t()
    ctx._pre()
    ctx._check(lib.og_zkey_import(ctx._h, buf, len(data), C.byref(pk_p), C.byref(pk_n), C.byref(vk_p), C.byref(vk_n)))
    pk = _take(lib, pk_p, pk_n)
    return pk, _take(lib, vk_p, vk_n)


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
