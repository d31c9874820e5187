"""The C-ABI library loads on a CPU-only host and exports every symbol the header declares
(no compute calls without a GPU)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "owshen_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(og_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from owshen_amd import _lib
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(_lib.lib, s), f"{s} declared in include/owshen_gpu.h but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature in owshen_amd/_lib.py"
    for s in _lib.SIGNATURES:
        assert s in syms, f"{s} bound in _lib.py but not declared in the header"


def test_no_device_is_a_loud_error():
    import ctypes as C
    from owshen_amd import _lib
    if _lib.lib.og_device_count() > 0:
        return  # on the GPU box this path is not reachable
    h = C.c_void_p()
    rc = _lib.lib.og_init(0, C.byref(h))
    assert rc == -3 and not h.value
    assert b"no HIP device" in _lib.lib.og_last_error()


def test_product_never_imports_oracle():
    """The product path must not route through the oracle or the CPU interpreter (task rule 3): no import,
    include, link or dlopen of anything under oracle/ or tests/ from owshen_amd/ (comments may cite the spec)."""
    pkg = os.path.join(ROOT, "owshen_amd")
    bad = [r"^\s*(from|import)\s+(oracle|tests)\b", r"#\s*include\s*[\"<][^\">]*(oracle|hipemu)", r"liboracle", r"libowshen_emu",
           r"CDLL\([^)]*(oracle|tests)"]
    for dp, _dn, fns in os.walk(pkg):
        if os.path.basename(dp) == "build":
            continue
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".h")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn), errors="replace").read()
                for pat in bad:
                    assert not re.search(pat, txt, flags=re.M), (os.path.join(dp, fn), pat)


def test_null_and_invalid_arguments_are_errors_not_crashes():
    """every entry point validates its handles before touching the device (runs without a GPU)"""
    import ctypes as C
    from owshen_amd import _lib
    lib = _lib.lib
    buf = (C.c_uint8 * 256)()
    null = C.c_void_p()
    h = C.c_void_p()
    assert lib.og_sync(null) == -1 and b"null og_ctx" in lib.og_last_error()
    assert lib.og_pk_load(null, buf, 256, C.byref(h)) == -1
    assert lib.og_prove(null, null, buf, buf, buf) == -1
    assert lib.og_prove_batch_d(null, null, buf, 1, buf, buf) == -1
    assert lib.og_withdraw_prove_batch_d(null, null, 32, 0, 0, buf, 1, buf, buf, None) == -1
    assert lib.og_release_scratch(null) == -1
    assert lib.og_msm_d(null, null, buf, 1, 1, 32, buf) == -1
    assert lib.og_ntt_fr_d(null, buf, buf, 3, 1, 0, 0) == -1
    assert lib.og_mimc7_hash2_d(null, buf, buf, buf, 1) == -1
    assert lib.og_profile(null, 1) == -1 and lib.og_set_lanes(null, 2) == -1
    shp = (C.c_uint64 * 3)()
    assert lib.og_withdraw_shape(0, 0, 0, shp) == -1 and lib.og_withdraw_shape(65, 0, 0, shp) == -1
    assert lib.og_withdraw_shape(32, 0, 0, shp) == 0 and list(shp) == [26385, 26346, 6]
    ok = C.c_int(7)
    assert lib.og_verify(None, 0, None, 0, buf, C.byref(ok)) == -1
    info = (C.c_uint64 * 4)()
    assert lib.og_pk_info(null, info) == -1
    lib.og_shutdown(null)   # no-ops on null handles
    lib.og_pk_free(null)
    lib.og_bases_free(null)
