"""og_verify without the GPU stack: binds libowshen_verify.so, the host-only build of the verifier (owshen_amd/csrc/
verify_only.cpp).  For the `burn_tx` seam (/root/reference/src/blockchain/tx/burn_tx.rs:11-32) on a node that only replays
blocks: no torch, no libamdhip64, no librccl -- ctypes and this one library."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libowshen_verify.so")
if not os.path.exists(_SO):
    raise ImportError(f"{_SO} is missing: run `make -C owshen_amd/csrc` (or __graft_entry__.build())")
lib = C.CDLL(_SO)
lib.og_verify.restype = C.c_int
lib.og_verify.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_int)]
lib.og_last_error.restype = C.c_char_p
lib.og_last_error.argtypes = []


def verify(vk_blob, public_inputs, proof256):
    """vk_blob: "OWVK0001" bytes; public_inputs: ints or 32-byte little-endian values; proof256: 256 bytes -> bool.
    Raises ValueError for a malformed verifying key (the library's OG_ERR_INVALID)."""
    pub = b"".join(int(x).to_bytes(32, "little") if isinstance(x, int) else bytes(x) for x in public_inputs)
    vk, proof = bytes(vk_blob), bytes(proof256)
    if len(proof) != 256 or len(pub) != 32 * len(public_inputs):
        raise ValueError("proof must be 256 bytes, every public input 32 bytes")
    ok = C.c_int(0)
    rc = lib.og_verify(vk, len(vk), pub if pub else None, len(public_inputs), proof, C.byref(ok))
    if rc != 0:
        raise ValueError(lib.og_last_error().decode("utf-8", "replace"))
    return ok.value == 1
