"""Host-side mirror of the C ABI (include/owshen_gpu.h) for tests, bench and Python callers.

Device buffers are torch uint8 CUDA tensors (PyTorch is only the HBM allocator here); every
compute call goes through ctypes into libowshen_gpu.so.  Field elements are 32-byte
little-endian canonical, the format of the reference's ``Fp::to_repr()``
(/root/reference/src/blockchain/tx/owshen_airdrop/babyjubjub/mod.rs:7-11).
"""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check, OwshenGpuError  # noqa: F401

FR = 0
FQ = 1
FR_MODULUS = 21888242871839275222246405745257275088548364400416034343698204186575808495617
FQ_MODULUS = 21888242871839275222246405745257275088696311157297823662689037894645226208583


# ---- int <-> bytes helpers (host logic, no arithmetic) ------------------------

def ints_to_bytes(vals, width=32):
    """list of python ints -> np.uint8 array [n, width], little-endian."""
    out = np.empty((len(vals), width), dtype=np.uint8)
    for i, v in enumerate(vals):
        out[i] = np.frombuffer(int(v).to_bytes(width, "little"), dtype=np.uint8)
    return out


def bytes_to_ints(arr, width=32):
    a = np.ascontiguousarray(arr, dtype=np.uint8).reshape(-1, width)
    return [int.from_bytes(row.tobytes(), "little") for row in a]


class Context:
    """One og_ctx (one GPU).  Mirrors og_init / og_shutdown.

    Device buffers are torch uint8 CUDA tensors.  The buffer plumbing (`to_device`, `to_host`,
    `empty`, `ptr`, `_pre`) is the only torch-dependent part; every other method is written
    against it."""

    _lib = lib

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise OwshenGpuError(-3, "no GPU visible to torch; owshen_amd has no CPU fallback")
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        self._check(self._lib.og_init(device, C.byref(h)))
        self._h = h

    def _check(self, code):
        if code != 0:
            raise OwshenGpuError(code, self._lib.og_last_error().decode("utf-8", "replace"))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.og_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing --
    def to_device(self, arr):
        """np array -> CUDA uint8 tensor holding the same bytes (non-u8 dtypes are viewed as bytes)."""
        arr = np.ascontiguousarray(arr)
        if arr.dtype != np.uint8:
            arr = arr.view(np.uint8)
        return torch.from_numpy(arr).to(self.device)

    def to_host(self, buf):
        return buf.cpu().numpy()

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.uint8, device=self.device)

    def ptr(self, t):
        if t is None:
            return None
        assert t.is_cuda and t.is_contiguous(), "device buffers must be contiguous CUDA tensors"
        return C.c_void_p(t.data_ptr())

    def _pre(self):
        """torch fills buffers on its own stream; the library runs on the ctx stream."""
        torch.cuda.synchronize()

    def sync(self):
        self._check(self._lib.og_sync(self._h))

    @property
    def stream_ptr(self):
        return self._lib.og_stream(self._h)

    PROFILE_KINDS = ("accumulate_g1", "accumulate_g2", "h_poly", "digit_sort", "reduce_g1", "reduce_g2", "witness",
                     "spmv", "assemble", "heavy_g1", "heavy_g2")

    def profile(self, enable):
        self._check(self._lib.og_profile(self._h, int(enable)))

    def release_scratch(self):
        """og_release_scratch: hand the sub-batch scratch arena back to the allocator (it regrows on demand)."""
        self._check(self._lib.og_release_scratch(self._h))

    def set_scratch_budget(self, n_bytes):
        """bound the HBM the prover's sub-batch slots may reserve together (og_set_scratch_budget; 0 = default)"""
        self._check(self._lib.og_set_scratch_budget(self._h, int(n_bytes)))

    def set_host_chains(self, max_requests):
        """withdraw calls of at most `max_requests` requests walk their MiMC7 chains on the host CPU (og_set_host_chains; 0 = never,
        the default): the latency form for a handler that proves one request per call"""
        self._check(self._lib.og_set_host_chains(self._h, int(max_requests)))

    def mem_info(self):
        """{"scratch_bytes", "scratch_buffers", "device_free_bytes", "device_total_bytes"} (og_mem_info)"""
        out = (C.c_uint64 * 4)()
        self._check(self._lib.og_mem_info(self._h, out))
        return dict(zip(("scratch_bytes", "scratch_buffers", "device_free_bytes", "device_total_bytes"), (int(x) for x in out)))

    def set_lanes(self, n):
        self._check(self._lib.og_set_lanes(self._h, int(n)))

    def profile_read(self):
        """{region: (total_ms, launches, units)} since profile(True)."""
        out = {}
        buf = (C.c_double * 3)()
        for k, name in enumerate(self.PROFILE_KINDS):
            self._check(self._lib.og_profile_read(self._h, k, buf))
            out[name] = (buf[0], int(buf[1]), buf[2])
        return out

    # -- N1 field --
    def field_op(self, field, op, a, b=None):
        """a, b: device uint8 [n,32] canonical.  op in {'add','sub','mul','inv'}."""
        opc = {"add": 0, "sub": 1, "mul": 2, "inv": 3}[op]
        self._pre()
        out = self.empty(*a.shape)
        self._check(self._lib.og_field_op_d(self._h, field, opc, self.ptr(a), self.ptr(b if b is not None else a),
                                            self.ptr(out), a.shape[0]))
        return out

    def field_mulchain(self, field, x, y, iters):
        self._pre()
        ms = C.c_float()
        self._check(self._lib.og_field_mulchain_d(self._h, field, self.ptr(x), self.ptr(y), x.shape[0], iters, C.byref(ms)))
        return ms.value

    def field_mulchain_lat(self, field, form, x, y, iters):
        """x <- x y, iters times, as a latency probe (form 0: lane-local product; 1: the wave-wide w9 product): (ms, wave cycles)"""
        self._pre()
        ms, cyc = C.c_float(), C.c_uint64()
        self._check(self._lib.og_field_mulchain_lat_d(self._h, field, form, self.ptr(x), self.ptr(y), x.shape[0], iters,
                                                      C.byref(ms), C.byref(cyc)))
        return ms.value, int(cyc.value)

    def ubench(self, kind, iters, blocks):
        ms = C.c_float()
        self._check(self._lib.og_ubench(self._h, kind, iters, blocks, C.byref(ms)))
        return ms.value

    def ubench_cycles(self, kind, iters, blocks):
        """(kernel ms, longest wave's loop time in shader cycles)"""
        ms, cyc = C.c_float(), C.c_uint64()
        self._check(self._lib.og_ubench_cycles(self._h, kind, iters, blocks, C.byref(ms), C.byref(cyc)))
        return ms.value, int(cyc.value)

    # -- N5 MiMC7 --
    def mimc7_constants(self):
        buf = (C.c_uint8 * (91 * 32))()
        self._check(self._lib.og_mimc7_constants(self._h, buf))
        return bytes_to_ints(np.frombuffer(bytes(buf), dtype=np.uint8))

    def mimc7_hash2(self, left, right):
        self._pre()
        out = self.empty(*left.shape)
        self._check(self._lib.og_mimc7_hash2_d(self._h, self.ptr(left), self.ptr(right), self.ptr(out), left.shape[0]))
        return out

    def mimc7_merkle_paths(self, leaves, indices, siblings, depth):
        """leaves [n,32] u8, indices [n] int64/uint64 buffer, siblings [n,depth,32] u8 -> [n,depth+1,32]."""
        self._pre()
        n = leaves.shape[0]
        out = self.empty(n, depth + 1, 32)
        self._check(self._lib.og_mimc7_merkle_paths_d(self._h, self.ptr(leaves), self.ptr(indices), self.ptr(siblings),
                                                      depth, self.ptr(out), n))
        return out

    def mimc7_tree_build(self, leaves):
        self._pre()
        n = leaves.shape[0]
        out = self.empty(2 * n - 1, 32)
        self._check(self._lib.og_mimc7_tree_build_d(self._h, self.ptr(leaves), n, self.ptr(out)))
        return out

    def mimc7_append(self, depth, frontier, next_index, leaves):
        """batched append to an incremental tree: frontier [depth,32], leaves [k,32] -> (new frontier [depth,32], root [32])"""
        self._pre()
        out_f, root = self.empty(depth, 32), self.empty(32)
        self._check(self._lib.og_mimc7_append_d(self._h, depth, self.ptr(frontier), next_index, self.ptr(leaves), leaves.shape[0],
                                                self.ptr(out_f), self.ptr(root)))
        return out_f, root

    def eddsa_verify(self, records):
        """records: device uint8 [n, 6, 32] (pk.x | pk.y | R.x | R.y | s | message) -> np.uint32 [n] (1 = accept)"""
        self._pre()
        n = records.shape[0]
        out = np.zeros(n, dtype=np.uint32)
        self._check(self._lib.og_eddsa_verify_batch_d(self._h, self.ptr(records), n, out.ctypes.data_as(C.c_void_p)))
        return out

    # -- N4 NTT --
    def ntt(self, data, inverse=False, coset=False):
        """data: device uint8 [n,32] or [batch,n,32] canonical -> same shape."""
        self._pre()
        shp = tuple(data.shape)
        d3 = data if len(shp) == 3 else data[None]
        batch, n = d3.shape[0], d3.shape[1]
        out = self.empty(batch, n, 32)
        self._check(self._lib.og_ntt_fr_d(self._h, self.ptr(d3), self.ptr(out), n.bit_length() - 1, batch, int(inverse),
                                          int(coset)))
        return out.reshape(shp)

    def h_poly(self, a, b, c):
        self._pre()
        shp = tuple(a.shape)
        batch, n = (shp[0], shp[1]) if len(shp) == 3 else (1, shp[0])
        out = self.empty(batch, n, 32)
        self._check(self._lib.og_h_poly_d(self._h, self.ptr(a), self.ptr(b), self.ptr(c), n.bit_length() - 1, batch,
                                          self.ptr(out)))
        return out.reshape(shp)

    # -- key-generation helpers --
    def scalar_mul(self, group, base_bytes, scalars):
        """out[i] = k_i * base.  base_bytes: 64/128 B canonical affine (host); scalars: device [n,32]."""
        self._pre()
        n = scalars.shape[0]
        pb = 64 if group == 1 else 128
        assert len(base_bytes) == pb
        out = self.empty(n, pb)
        base = (C.c_uint8 * pb).from_buffer_copy(bytes(base_bytes))
        self._check(self._lib.og_scalar_mul_d(self._h, group, base, self.ptr(scalars), n, self.ptr(out)))
        return out

    def lagrange_evals(self, log_d, tau):
        self._pre()
        out = self.empty(1 << log_d, 32)
        t = (C.c_uint8 * 32).from_buffer_copy(int(tau).to_bytes(32, "little"))
        self._check(self._lib.og_lagrange_evals_d(self._h, log_d, t, self.ptr(out)))
        return out

    def spmv(self, row_ptr, col, val, x, n_rows):
        """CSR (device byte buffers: row_ptr u32 [n_rows+1], col u32 [nnz], val [nnz,32]) times x [n,32]."""
        self._pre()
        out = self.empty(n_rows, 32)
        self._check(self._lib.og_spmv_fr_d(self._h, self.ptr(row_ptr), self.ptr(col), self.ptr(val), n_rows, self.ptr(x),
                                           self.ptr(out)))
        return out


class Bases:
    """Device-resident MSM bases (og_bases).  group: 1 = G1, 2 = G2."""

    def __init__(self, ctx, group, points, window_bits=0, precompute=False):
        """points: device uint8 [n, 64|128] canonical affine."""
        ctx._pre()
        self.ctx, self.group, self.n = ctx, group, points.shape[0]
        h = C.c_void_p()
        ctx._check(ctx._lib.og_bases_create_d(ctx._h, group, ctx.ptr(points), self.n, window_bits, int(precompute),
                                              C.byref(h)))
        self._h = h

    def msm(self, scalars, n=None):
        """scalars: device uint8 [n,32] or [batch,n,32] -> np.uint8 [batch, 64|128] affine canonical."""
        ctx = self.ctx
        ctx._pre()
        if len(scalars.shape) == 2:
            scalars = scalars[None]
        batch, nn = scalars.shape[0], scalars.shape[1]
        n = nn if n is None else n
        pb = 64 if self.group == 1 else 128
        out = np.zeros((batch, pb), dtype=np.uint8)
        ctx._check(ctx._lib.og_msm_d(ctx._h, self._h, ctx.ptr(scalars), n, batch, nn * 32, out.ctypes.data_as(C.c_void_p)))
        return out

    def partial_bytes(self):
        """size of one rank's partial array for the window-sharded MSM (og_msm_partial_slots x 128 | 256 B)"""
        return int(self.ctx._lib.og_msm_partial_slots(self._h)) * (128 if self.group == 1 else 256)

    def msm_windows(self, scalars, win_rank, win_world):
        """this rank's share of a window-sharded MSM: scalars device uint8 [n,32] -> device uint8 [partial_bytes()]"""
        ctx = self.ctx
        ctx._pre()
        out = ctx.empty(self.partial_bytes())
        ctx._check(ctx._lib.og_msm_windows_d(ctx._h, self._h, ctx.ptr(scalars), scalars.shape[0], win_rank, win_world, ctx.ptr(out)))
        return out

    def msm_combine(self, gathered, world):
        """gathered: device uint8 [world * partial_bytes()] (rank-major) -> np.uint8 [64 | 128] canonical affine"""
        ctx = self.ctx
        ctx._pre()
        out = np.zeros(64 if self.group == 1 else 128, dtype=np.uint8)
        ctx._check(ctx._lib.og_msm_combine_d(ctx._h, self._h, ctx.ptr(gathered), world, out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._lib.og_bases_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
