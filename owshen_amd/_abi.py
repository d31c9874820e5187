"""ctypes signatures of the C ABI, one entry per declaration in include/owshen_gpu.h (pure data)."""
import ctypes as C

_vp, _sz, _i, _u8p = C.c_void_p, C.c_size_t, C.c_int, C.c_void_p
_fp = C.POINTER(C.c_float)

# name -> (restype, argtypes); mirrors include/owshen_gpu.h one to one
SIGNATURES = {
    "og_init": (_i, [_i, C.POINTER(_vp)]),
    "og_shutdown": (None, [_vp]),
    "og_last_error": (C.c_char_p, []),
    "og_device_count": (_i, []),
    "og_sync": (_i, [_vp]),
    "og_stream": (_vp, [_vp]),
    "og_malloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "og_free": (_i, [_vp, _vp]),
    "og_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "og_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "og_field_op_d": (_i, [_vp, _i, _i, _u8p, _u8p, _u8p, _sz]),
    "og_field_mulchain_d": (_i, [_vp, _i, _u8p, _u8p, _sz, _i, _fp]),
    "og_ubench": (_i, [_vp, _i, _i, _i, _fp]),
    "og_ubench_cycles": (_i, [_vp, _i, _i, _i, _fp, C.POINTER(C.c_uint64)]),
    "og_ubench_coresidency": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fp]),
    "og_mimc7_constants": (_i, [_vp, _vp]),
    "og_mimc7_hash2_d": (_i, [_vp, _u8p, _u8p, _u8p, _sz]),
    "og_mimc7_merkle_paths_d": (_i, [_vp, _u8p, _vp, _u8p, _i, _u8p, _sz]),
    "og_mimc7_tree_build_d": (_i, [_vp, _u8p, _sz, _u8p]),
    "og_mimc7_append_d": (_i, [_vp, _i, _u8p, C.c_uint64, _u8p, _sz, _u8p, _u8p]),
    "og_eddsa_verify_batch_d": (_i, [_vp, _u8p, _sz, _vp]),
    "og_ntt_fr_d": (_i, [_vp, _u8p, _u8p, _i, _i, _i, _i]),
    "og_h_poly_d": (_i, [_vp, _u8p, _u8p, _u8p, _i, _i, _u8p]),
    "og_bases_create_d": (_i, [_vp, _i, _u8p, _sz, _i, _i, C.POINTER(_vp)]),
    "og_bases_free": (None, [_vp]),
    "og_msm_d": (_i, [_vp, _vp, _u8p, _sz, _i, _sz, _vp]),
    "og_msm_partial_slots": (_i, [_vp]),
    "og_msm_windows_d": (_i, [_vp, _vp, _u8p, _sz, _i, _i, _u8p]),
    "og_msm_combine_d": (_i, [_vp, _vp, _u8p, _i, _vp]),
    "og_multi_init": (_i, [_i, C.POINTER(_vp)]),
    "og_multi_shutdown": (None, [_vp]),
    "og_multi_size": (_i, [_vp]),
    "og_multi_slice": (_i, [_vp, _sz, _i, C.POINTER(_sz)]),
    "og_multi_ctx": (_vp, [_vp, _i]),
    "og_multi_pk_load": (_i, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "og_multi_pk_free": (None, [_vp, C.POINTER(_vp)]),
    "og_multi_prove_batch": (_i, [_vp, C.POINTER(_vp), _vp, _sz, _vp, _vp]),
    "og_multi_withdraw_prove_batch": (_i, [_vp, C.POINTER(_vp), _i, C.c_uint64, C.c_uint64, _vp, _sz, _vp, _vp, _vp]),
    "og_multi_bases_create": (_i, [_vp, _i, _vp, _sz, _i, _i, C.POINTER(_vp)]),
    "og_multi_bases_free": (None, [_vp, C.POINTER(_vp)]),
    "og_multi_msm": (_i, [_vp, C.POINTER(_vp), _vp, _sz, _vp]),
    "og_withdraw_r1cs": (_i, [_vp, _i, C.c_uint64, C.c_uint64, _i, C.POINTER(_vp)]),
    "og_r1cs_from_csr": (_i, [C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "og_r1cs_free": (None, [_vp]),
    "og_r1cs_info": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "og_r1cs_export": (_i, [_vp, _i, _vp, _vp, _vp]),
    "og_setup": (_i, [_vp, _vp, _vp, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_vp), C.POINTER(_sz)]),
    "og_blob_free": (None, [_vp]),
    "og_pk_load": (_i, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "og_pk_free": (None, [_vp]),
    "og_pk_info": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "og_pk_density": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "og_pk_windows": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "og_prove": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "og_prove_batch": (_i, [_vp, _vp, _vp, _sz, _vp, _vp]),
    "og_prove_batch_d": (_i, [_vp, _vp, _u8p, _sz, _vp, _vp]),
    "og_verify": (_i, [_vp, _sz, _vp, _sz, _vp, C.POINTER(_i)]),
    "og_withdraw_shape": (_i, [_i, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    "og_withdraw_witness_d": (_i, [_vp, _i, C.c_uint64, C.c_uint64, _u8p, _sz, _u8p]),
    "og_profile": (_i, [_vp, _i]),
    "og_set_lanes": (_i, [_vp, _i]),
    "og_profile_read": (_i, [_vp, _i, C.POINTER(C.c_double)]),
    "og_release_scratch": (_i, [_vp]),
    "og_withdraw_prove_batch_d": (_i, [_vp, _vp, _i, C.c_uint64, C.c_uint64, _u8p, _sz, _vp, _vp, _vp]),
    "og_withdraw_prove_batch_submit_d": (_i, [_vp, _vp, _i, C.c_uint64, C.c_uint64, _u8p, _sz, _vp, _vp, _vp, C.POINTER(_vp)]),
    "og_job_wait": (_i, [_vp, _vp]),
    "og_job_abandon": (_i, [_vp, _vp]),
    "og_job_poll": (_i, [_vp, _vp, _vp]),
    "og_multi_device_info": (_i, [_vp, _i, _vp, _vp]),
    "og_mem_info": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "og_pk_bytes": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "og_glv_decompose": (_i, [_vp, _vp]),
    "og_prove_plan": (_i, [_vp, _vp, _sz, C.POINTER(C.c_uint32), _sz, C.POINTER(_sz), C.POINTER(_i)]),
    "og_scalar_mul_d": (_i, [_vp, _i, _vp, _u8p, _sz, _u8p]),
    "og_lagrange_evals_d": (_i, [_vp, _i, _vp, _u8p]),
    "og_spmv_fr_d": (_i, [_vp, _vp, _vp, _u8p, _sz, _u8p, _u8p]),
}


def bind(lib):
    """attach restype/argtypes to every exported symbol; AttributeError = header/library drift."""
    for name, (res, args) in SIGNATURES.items():
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = args
    return lib
