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


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device buffers must be contiguous CUDA tensors"
    return C.c_void_p(t.data_ptr())


class Context:
    """One og_ctx (one GPU).  Mirrors og_init / og_shutdown."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise OwshenGpuError(-3, "no GPU visible to torch; owshen_amd has no CPU fallback")
        self.device = torch.device("cuda", device)
        h = C.c_void_p()
        check(lib.og_init(device, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            lib.og_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing --
    def to_device(self, arr):
        """np.uint8 array -> CUDA tensor (same shape)."""
        return torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)

    def empty(self, *shape):
        return torch.empty(*shape, dtype=torch.uint8, device=self.device)

    def sync(self):
        check(lib.og_sync(self._h))

    @property
    def stream_ptr(self):
        return lib.og_stream(self._h)

    # -- N1 field --
    def field_op(self, field, op, a, b=None):
        """a, b: CUDA uint8 [n,32] canonical.  op in {'add','sub','mul','inv'}."""
        opc = {"add": 0, "sub": 1, "mul": 2, "inv": 3}[op]
        torch.cuda.synchronize()
        out = torch.empty_like(a)
        check(lib.og_field_op_d(self._h, field, opc, _ptr(a), _ptr(b if b is not None else a), _ptr(out),
                                a.shape[0]))
        return out

    def field_mulchain(self, field, x, y, iters):
        torch.cuda.synchronize()
        ms = C.c_float()
        check(lib.og_field_mulchain_d(self._h, field, _ptr(x), _ptr(y), x.shape[0], iters, C.byref(ms)))
        return ms.value

    def ubench(self, kind, iters, blocks):
        ms = C.c_float()
        check(lib.og_ubench(self._h, kind, iters, blocks, C.byref(ms)))
        return ms.value

    # -- N5 MiMC7 --
    def mimc7_constants(self):
        buf = (C.c_uint8 * (91 * 32))()
        check(lib.og_mimc7_constants(self._h, buf))
        return bytes_to_ints(np.frombuffer(bytes(buf), dtype=np.uint8))

    def mimc7_hash2(self, left, right):
        torch.cuda.synchronize()
        out = torch.empty_like(left)
        check(lib.og_mimc7_hash2_d(self._h, _ptr(left), _ptr(right), _ptr(out), left.shape[0]))
        return out

    def mimc7_merkle_paths(self, leaves, indices, siblings, depth):
        """leaves [n,32] u8, indices [n] int64/uint64 tensor, siblings [n,depth,32] u8 -> [n,depth+1,32]."""
        torch.cuda.synchronize()
        n = leaves.shape[0]
        out = self.empty(n, depth + 1, 32)
        check(lib.og_mimc7_merkle_paths_d(self._h, _ptr(leaves), C.c_void_p(indices.data_ptr()), _ptr(siblings),
                                          depth, _ptr(out), n))
        return out

    def mimc7_tree_build(self, leaves):
        torch.cuda.synchronize()
        n = leaves.shape[0]
        out = self.empty(2 * n - 1, 32)
        check(lib.og_mimc7_tree_build_d(self._h, _ptr(leaves), n, _ptr(out)))
        return out

    # -- N4 NTT --
    def ntt(self, data, inverse=False, coset=False):
        """data: CUDA uint8 [n,32] or [batch,n,32] canonical -> same shape."""
        torch.cuda.synchronize()
        shp = data.shape
        d3 = data if data.dim() == 3 else data.unsqueeze(0)
        batch, n = d3.shape[0], d3.shape[1]
        out = torch.empty_like(d3)
        check(lib.og_ntt_fr_d(self._h, _ptr(d3), _ptr(out), n.bit_length() - 1, batch, int(inverse), int(coset)))
        return out.reshape(shp)

    def h_poly(self, a, b, c):
        torch.cuda.synchronize()
        shp = a.shape
        a3 = a if a.dim() == 3 else a.unsqueeze(0)
        batch, n = a3.shape[0], a3.shape[1]
        out = torch.empty_like(a3)
        check(lib.og_h_poly_d(self._h, _ptr(a), _ptr(b), _ptr(c), n.bit_length() - 1, batch, _ptr(out)))
        return out.reshape(shp)


class Bases:
    """Device-resident MSM bases (og_bases).  group: 1 = G1, 2 = G2."""

    def __init__(self, ctx, group, points, window_bits=0, precompute=False):
        """points: CUDA uint8 [n, 64|128] canonical affine."""
        torch.cuda.synchronize()
        self.ctx, self.group, self.n = ctx, group, points.shape[0]
        h = C.c_void_p()
        check(lib.og_bases_create_d(ctx._h, group, _ptr(points), self.n, window_bits, int(precompute), C.byref(h)))
        self._h = h

    def msm(self, scalars, n=None):
        """scalars: CUDA uint8 [n,32] or [batch,n,32] -> np.uint8 [batch, 64|128] affine canonical."""
        torch.cuda.synchronize()
        if scalars.dim() == 2:
            scalars = scalars.unsqueeze(0)
        batch, nn = scalars.shape[0], scalars.shape[1]
        n = nn if n is None else n
        pb = 64 if self.group == 1 else 128
        out = np.zeros((batch, pb), dtype=np.uint8)
        check(lib.og_msm_d(self.ctx._h, self._h, _ptr(scalars), n, batch, nn * 32, out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            lib.og_bases_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
