"""Host-side mirror of the og_multi_* entry points (include/owshen_gpu.h): ONE process driving every GPU of the
node -- the shape of the reference's node, a single process holding one Context
(/root/reference/src/cli/node.rs:71-76).  Proofs are sharded across devices with a replicated key and no
data-path collective; a lone MSM is window-sharded with an RCCL broadcast of the scalars and an all-gather of
the per-window points inside the library.  All buffers here are host numpy arrays.
"""
import ctypes as C

import numpy as np

from ._lib import lib as _default_lib, OwshenGpuError


class Multi:
    def __init__(self, n_devices=0, lib=None):
        """n_devices = 0: every visible device.  Raises if fewer are visible than asked for."""
        self._lib = lib or _default_lib
        h = C.c_void_p()
        self._check(self._lib.og_multi_init(int(n_devices), C.byref(h)))
        self._h = h
        self.size = int(self._lib.og_multi_size(h))

    def _check(self, code):
        if code != 0:
            raise OwshenGpuError(code, self._lib.og_last_error().decode("utf-8", "replace"))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.og_multi_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def device_info(self, rank):
        """{"device", "pci", "comm_nranks", "comm_rank", "comm_device"} of rank `rank` (og_multi_device_info): the HIP ordinal and
        PCI address it is bound to, and what RCCL says about its communicator (zeros with one device)"""
        out, pci = (C.c_uint64 * 4)(), C.create_string_buffer(32)
        self._check(self._lib.og_multi_device_info(self._h, rank, out, pci))
        return {"device": int(out[0]), "pci": pci.value.decode("ascii", "replace"), "comm_nranks": int(out[1]), "comm_rank": int(out[2]),
                "comm_device": int(out[3])}

    def slice(self, n, rank):
        """[lo, hi) of a batch of n proofs that device `rank` proves (og_multi_slice)"""
        out = (C.c_size_t * 2)()
        self._check(self._lib.og_multi_slice(self._h, n, rank, out))
        return int(out[0]), int(out[1])

    @staticmethod
    def _p(arr):
        return arr.ctypes.data_as(C.c_void_p)

    def load_key(self, blob):
        """OWPK0001 blob -> per-device key handles (replicated)"""
        pks = (C.c_void_p * self.size)()
        buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
        self._check(self._lib.og_multi_pk_load(self._h, buf, len(blob), pks))
        return pks

    def free_key(self, pks):
        self._lib.og_multi_pk_free(self._h, pks)

    def prove_batch(self, pks, witnesses, rs):
        """witnesses np.uint8 [n, n_wires, 32], rs np.uint8 [n, 64] -> np.uint8 [n, 256]"""
        w = np.ascontiguousarray(witnesses, dtype=np.uint8)
        rsb = np.ascontiguousarray(rs, dtype=np.uint8).reshape(-1, 64)
        n = w.shape[0]
        assert rsb.shape[0] == n
        out = np.zeros((n, 256), dtype=np.uint8)
        self._check(self._lib.og_multi_prove_batch(self._h, pks, self._p(w), n, self._p(rsb), self._p(out)))
        return out

    def withdraw_prove_batch(self, pks, depth, inputs, rs, n_pad3=0, n_pad2=0, return_public=False):
        """inputs np.uint8 [n, 8 + depth, 32] (host records), rs np.uint8 [n, 64] -> np.uint8 [n, 256]
        (return_public: also the six public inputs of every proof, np.uint8 [n, 6, 32])"""
        x = np.ascontiguousarray(inputs, dtype=np.uint8)
        rsb = np.ascontiguousarray(rs, dtype=np.uint8).reshape(-1, 64)
        n = x.shape[0]
        assert x.shape[1:] == (8 + depth, 32) and rsb.shape[0] == n
        out = np.zeros((n, 256), dtype=np.uint8)
        pub = np.zeros((n, 6, 32), dtype=np.uint8) if return_public else None
        self._check(self._lib.og_multi_withdraw_prove_batch(self._h, pks, depth, n_pad3, n_pad2, self._p(x), n, self._p(rsb),
                                                            self._p(out), self._p(pub) if return_public else None))
        return (out, pub) if return_public else out

    def withdraw_prove_sharded(self, pks, depth, inputs, rs, n_pad3=0, n_pad2=0, return_public=False):
        """window-sharded PROVING (og_multi_withdraw_prove_sharded): ONE batch on all devices together -- every device walks the
        witnesses and the quotient itself and accumulates the windows k = rank (mod N) of the five queries; one RCCL all-gather
        of 768 B per proof and rank; device 0 assembles.  Same arguments and bytes as withdraw_prove_batch."""
        x = np.ascontiguousarray(inputs, dtype=np.uint8)
        rsb = np.ascontiguousarray(rs, dtype=np.uint8).reshape(-1, 64)
        n = x.shape[0]
        assert x.shape[1:] == (8 + depth, 32) and rsb.shape[0] == n
        out = np.zeros((n, 256), dtype=np.uint8)
        pub = np.zeros((n, 6, 32), dtype=np.uint8) if return_public else None
        self._check(self._lib.og_multi_withdraw_prove_sharded(self._h, pks, depth, n_pad3, n_pad2, self._p(x), n, self._p(rsb),
                                                              self._p(out), self._p(pub) if return_public else None))
        return (out, pub) if return_public else out

    def prove_sharded(self, pks, witnesses, rs):
        """the same for caller-supplied witnesses (og_multi_prove_sharded): np.uint8 [n, n_wires, 32], [n, 64] -> [n, 256]"""
        w = np.ascontiguousarray(witnesses, dtype=np.uint8)
        rsb = np.ascontiguousarray(rs, dtype=np.uint8).reshape(-1, 64)
        n = w.shape[0]
        assert rsb.shape[0] == n
        out = np.zeros((n, 256), dtype=np.uint8)
        self._check(self._lib.og_multi_prove_sharded(self._h, pks, self._p(w), n, self._p(rsb), self._p(out)))
        return out

    def bases(self, group, points, window_bits=0, precompute=False):
        """points np.uint8 [n, 64 | 128] canonical affine -> per-device bases handles (replicated)"""
        pts = np.ascontiguousarray(points, dtype=np.uint8)
        hs = (C.c_void_p * self.size)()
        self._check(self._lib.og_multi_bases_create(self._h, group, self._p(pts), pts.shape[0], window_bits, int(precompute), hs))
        return hs, (64 if group == 1 else 128)

    def free_bases(self, bases):
        self._lib.og_multi_bases_free(self._h, bases[0])

    def msm(self, bases, scalars):
        """window-sharded MSM: scalars np.uint8 [n, 32] -> np.uint8 [64 | 128] canonical affine"""
        hs, pb = bases
        sc = np.ascontiguousarray(scalars, dtype=np.uint8).reshape(-1, 32)
        out = np.zeros(pb, dtype=np.uint8)
        self._check(self._lib.og_multi_msm(self._h, hs, self._p(sc), sc.shape[0], self._p(out)))
        return out
