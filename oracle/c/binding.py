"""ctypes binding to oracle/_build/liboracle.so.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

All arrays are numpy uint8, 32-byte little-endian canonical field elements."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(os.path.dirname(_HERE), "_build", "liboracle.so")


def _load():
    """OG_ORACLE_NATIVE=1 (set by bench.py's CPU-baseline leg): build and load the -march=native variant ON this machine
    (it never travels); anything going wrong there falls back to the portable build every test uses."""
    if os.environ.get("OG_ORACLE_NATIVE"):
        try:
            subprocess.check_call(["make", "-s", "-C", _HERE, "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            return C.CDLL(os.path.join(os.path.dirname(_HERE), "_build", "native", "liboracle.so"))
        except (OSError, subprocess.CalledProcessError):
            pass
    if not os.path.exists(_SO):
        subprocess.check_call(["make", "-C", _HERE])
    lib = C.CDLL(_SO)
    return lib


NATIVE = bool(os.environ.get("OG_ORACLE_NATIVE"))


lib = _load()
_vp, _sz, _i = C.c_void_p, C.c_size_t, C.c_int
for name, res, args in [
    ("oc_field_op", None, [_i, _i, _vp, _vp, _vp, _sz]),
    ("oc_mimc7_set_constants", None, [_vp]),
    ("oc_mimc7_hash2", None, [_vp, _vp, _vp, _sz]),
    ("oc_mimc7_tree_build", None, [_vp, _sz, _vp, _i]),
    ("oc_ntt", None, [_vp, _i, _i, _i]),
    ("oc_msm_g1", None, [_vp, _vp, _sz, _vp, _i]),
    ("oc_msm_g2", None, [_vp, _vp, _sz, _vp, _i]),
    ("oc_bases_g1_new", _vp, [_vp, _sz, _i]),
    ("oc_bases_g2_new", _vp, [_vp, _sz, _i]),
    ("oc_bases_free", None, [_vp]),
    ("oc_msm_g1_prepared", None, [_vp, _vp, _sz, _vp, _i]),
    ("oc_msm_g2_prepared", None, [_vp, _vp, _sz, _vp, _i]),
    ("oc_fixed_base_g1", None, [_vp, _vp, _sz, _vp, _i]),
    ("oc_fixed_base_g2", None, [_vp, _vp, _sz, _vp, _i]),
    ("oc_pk_prepare", _vp, [_vp, _i]),
    ("oc_pk_free", None, [_vp]),
    ("oc_groth16_prove", _i, [_vp, _vp, _vp, _vp, _vp, _i]),
    ("oc_h_poly", None, [_vp, _vp, _vp, _i, _vp]),
    ("oc_prove_threads", _i, [C.c_uint64, _i]),
]:
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = args

def _usable_cpus():
    """hardware threads this process can keep busy: the scheduler affinity capped by the container's CPU quota (cgroup v2
    cpu.max / v1 cpu.cfs_quota_us) -- os.cpu_count() reports the machine, not the container (256 against 16 on the GPU boxes)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
            if q != "max":
                quota = int(q) / int(p)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0 and p > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    return max(1, min(n, int(quota + 0.5))) if quota else n


THREADS = _usable_cpus()


def prove_threads(n_wires=1 << 18, threads=None):
    """threads the C prover keeps busy in its MSM phase (5 concurrent MSMs x window parallelism)"""
    return int(lib.oc_prove_threads(n_wires, threads or THREADS))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


_consts_set = False


def _ensure_mimc():
    global _consts_set
    if not _consts_set:
        from ..py import mimc7
        c = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in mimc7.CONSTANTS), dtype=np.uint8).copy()
        lib.oc_mimc7_set_constants(_p(c))
        _consts_set = True


def field_op(field, op, a, b=None):
    a = _u8(a)
    b = a if b is None else _u8(b)
    out = np.empty_like(a)
    lib.oc_field_op(field, {"add": 0, "sub": 1, "mul": 2, "inv": 3}[op], _p(a), _p(b), _p(out), a.shape[0])
    return out


def mimc7_hash2(l, r):
    _ensure_mimc()
    l, r = _u8(l), _u8(r)
    out = np.empty_like(l)
    lib.oc_mimc7_hash2(_p(l), _p(r), _p(out), l.shape[0])
    return out


def mimc7_tree_build(leaves, threads=None):
    _ensure_mimc()
    leaves = _u8(leaves)
    n = leaves.shape[0]
    out = np.empty((2 * n - 1, 32), dtype=np.uint8)
    lib.oc_mimc7_tree_build(_p(leaves), n, _p(out), threads or THREADS)
    return out


def ntt(data, inverse=False, coset=False):
    a = _u8(data).copy()
    n = a.shape[0]
    lib.oc_ntt(_p(a), n.bit_length() - 1, int(inverse), int(coset))
    return a


def h_poly(a, b, c):
    a, b, c = _u8(a), _u8(b), _u8(c)
    out = np.empty_like(a)
    lib.oc_h_poly(_p(a), _p(b), _p(c), a.shape[0].bit_length() - 1, _p(out))
    return out


def msm_g1(bases, scalars, threads=None):
    bases, scalars = _u8(bases), _u8(scalars)
    out = np.zeros(64, dtype=np.uint8)
    lib.oc_msm_g1(_p(bases), _p(scalars), scalars.shape[0], _p(out), threads or THREADS)
    return out


def msm_g2(bases, scalars, threads=None):
    bases, scalars = _u8(bases), _u8(scalars)
    out = np.zeros(128, dtype=np.uint8)
    lib.oc_msm_g2(_p(bases), _p(scalars), scalars.shape[0], _p(out), threads or THREADS)
    return out


def fixed_base_g1(base64, scalars, threads=None):
    base64, scalars = _u8(base64), _u8(scalars)
    out = np.empty((scalars.shape[0], 64), dtype=np.uint8)
    lib.oc_fixed_base_g1(_p(base64), _p(scalars), scalars.shape[0], _p(out), threads or THREADS)
    return out


def fixed_base_g2(base128, scalars, threads=None):
    base128, scalars = _u8(base128), _u8(scalars)
    out = np.empty((scalars.shape[0], 128), dtype=np.uint8)
    lib.oc_fixed_base_g2(_p(base128), _p(scalars), scalars.shape[0], _p(out), threads or THREADS)
    return out


class OcPk(C.Structure):
    _fields_ = [("n_wires", C.c_uint64), ("n_pub", C.c_uint64), ("domain_log", C.c_uint64), ("n_rows", C.c_uint64)] + \
        [(f"{m}_{k}", C.c_void_p) for m in "abc" for k in ("ptr", "col", "val")] + \
        [(k, C.c_void_p) for k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2",
                                   "a_query", "b_g1_query", "b_g2_query", "l_query", "h_query")] + [("flags", C.c_uint64)]


class PreparedKey:
    """Holds numpy arrays alive + the C-side prepared (Montgomery) copy."""

    def __init__(self, n_wires, n_pub, domain_log, n_rows, csr, points, threads=None, flags=0):
        """csr: {'a': (ptr u32, col u32, val u8[nnz,32]), 'b':..., 'c':...}; points: dict of u8 arrays; flags: header word 8."""
        self.keep = []
        s = OcPk()
        s.n_wires, s.n_pub, s.domain_log, s.n_rows = n_wires, n_pub, domain_log, n_rows
        s.flags = flags
        for m in "abc":
            ptr, col, val = csr[m]
            ptr = np.ascontiguousarray(ptr, dtype=np.uint32)
            col = np.ascontiguousarray(col, dtype=np.uint32)
            val = _u8(val)
            self.keep += [ptr, col, val]
            setattr(s, f"{m}_ptr", ptr.ctypes.data)
            setattr(s, f"{m}_col", col.ctypes.data)
            setattr(s, f"{m}_val", val.ctypes.data)
        for k in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2", "a_query", "b_g1_query", "b_g2_query",
                  "l_query", "h_query"):
            arr = _u8(points[k])
            self.keep.append(arr)
            setattr(s, k, arr.ctypes.data)
        self.struct = s
        self.handle = lib.oc_pk_prepare(C.byref(s), threads or THREADS)

    def prove(self, witness, r, s, threads=None):
        witness = _u8(witness)
        rb = np.frombuffer(int(r).to_bytes(32, "little"), dtype=np.uint8).copy()
        sb = np.frombuffer(int(s).to_bytes(32, "little"), dtype=np.uint8).copy()
        out = np.zeros(256, dtype=np.uint8)
        rc = lib.oc_groth16_prove(self.handle, _p(witness), _p(rb), _p(sb), _p(out), threads or THREADS)
        if rc != 0:
            raise ValueError(f"oc_groth16_prove rc={rc} (witness does not satisfy the circuit)")
        return out.tobytes()

    def __del__(self):
        try:
            lib.oc_pk_free(self.handle)
        except Exception:
            pass


def parse_pk_blob(blob):
    """OWPK0001 (include/owshen_gpu.h) -> dict of header fields, CSR arrays and point arrays."""
    import struct
    assert blob[:8] == b"OWPK0001"
    m, l, log_d, n_rows, na, nb, nc, flags, _ = struct.unpack("<9Q", blob[8:80])
    off = 80
    out = {"n_wires": m, "n_pub": l, "log_d": log_d, "n_rows": n_rows, "flags": flags}

    def take(nbytes):
        nonlocal off
        b = blob[off:off + nbytes]
        off += (nbytes + 31) // 32 * 32
        return b
    c1 = take(256)
    c2 = take(256)
    out["alpha_g1"], out["beta_g1"], out["delta_g1"] = c1[0:64], c1[64:128], c1[128:192]
    out["beta_g2"], out["delta_g2"] = c2[0:128], c2[128:256]
    out["csr"] = {}
    for name, nnz in (("a", na), ("b", nb), ("c", nc)):
        ptr = np.frombuffer(take((n_rows + 1) * 4), dtype=np.uint32).copy()
        col = np.frombuffer(take(nnz * 4), dtype=np.uint32).copy()
        val = np.frombuffer(take(nnz * 32), dtype=np.uint8).reshape(-1, 32).copy()
        out["csr"][name] = (ptr, col, val)
    d = 1 << log_d
    for name, n, w in (("a_query", m, 64), ("b_g1_query", m, 64), ("b_g2_query", m, 128), ("l_query", m - l - 1, 64),
                       ("h_query", d - 1, 64)):
        out[name] = np.frombuffer(take(n * w), dtype=np.uint8).reshape(-1, w).copy()
    assert off == len(blob)
    return out


def prepared_key_from_blob(blob):
    """the product's serialized proving key, handed byte-for-byte to the C restatement"""
    k = parse_pk_blob(blob)
    points = {n: np.frombuffer(k[n], dtype=np.uint8).copy() for n in ("alpha_g1", "beta_g1", "beta_g2", "delta_g1", "delta_g2")}
    for n in ("a_query", "b_g1_query", "b_g2_query", "l_query", "h_query"):
        points[n] = k[n]
    return PreparedKey(k["n_wires"], k["n_pub"], k["log_d"], k["n_rows"], k["csr"], points, flags=k["flags"])
