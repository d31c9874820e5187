"""og_multi_* (one process, several devices) on the CPU interpreter: OG_EMU_DEVICES pretends to have 3 devices, the
ranks run in turn and the RCCL calls are the in-process stand-in of tests/hipemu/rccl/rccl.h.  Checks the host logic of
owshen_amd/csrc/multi.hip -- slicing, key / bases replication, the broadcast + all-gather + Horner combine of the
window-sharded MSM -- against the C oracle.  The real RCCL path runs in tests/test_gpu_multi.py."""
import random

import numpy as np
import pytest

from oracle.py import fields
from oracle.py.curve import G1_GEN, g1_to_bytes


@pytest.fixture(scope="module")
def emu3():
    import os
    os.environ["OG_EMU_DEVICES"] = "3"
    os.environ["OG_MULTI_SEQUENTIAL"] = "1"   # the interpreter is single-threaded: ranks run in turn
    from tests import emu
    from owshen_amd import multi
    m = multi.Multi(3, lib=emu.lib)
    yield emu, m
    m.close()
    os.environ.pop("OG_EMU_DEVICES", None)
    os.environ.pop("OG_MULTI_SEQUENTIAL", None)


def _rand_fr(rng, *shape):
    a = rng.integers(0, 256, (*shape, 32), dtype=np.uint8)
    a[..., 31] &= 0x1F
    return a


def test_emu_multi_init_refuses_missing_devices():
    from tests import emu
    from owshen_amd import multi, api
    with pytest.raises(api.OwshenGpuError, match="asked for 64 devices"):
        multi.Multi(64, lib=emu.lib)


@pytest.mark.parametrize("window,precomp", [(8, False), (12, True)])
def test_emu_multi_msm_window_sharded(emu3, window, precomp):
    from oracle.c import binding as oc
    emu, m = emu3
    assert m.size == 3
    rng = np.random.default_rng(window)
    n = 200
    bases_np = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), _rand_fr(rng, n))
    sc = _rand_fr(rng, n)
    sc[:3] = 0
    sc[1, 0] = 1
    b = m.bases(1, bases_np, window, precomp)
    assert m.msm(b, sc).tobytes() == oc.msm_g1(bases_np, sc).tobytes()
    m.free_bases(b)


def test_emu_multi_prove_batch_shards_proofs(emu3):
    """7 proofs over 3 devices (3 + 2 + 2): byte-identical to the C restatement, in the caller's order"""
    from oracle.c import binding as oc
    from owshen_amd import groth16 as g16
    from tests.r1cs_util import random_r1cs
    emu, m = emu3
    ctx = emu.Ctx()
    n_wires, cons, z0 = random_r1cs(10, 1, seed=5)
    blob, _vk = g16.setup(ctx, g16.R1CS.from_constraints(n_wires, 1, cons), 3, 5, 7, 11, 13)
    rnd = random.Random(8)
    zs, rs = [], []
    for t in range(7):
        z = list(z0)
        r2 = random.Random(70 + t)
        for i in range(1, n_wires - len(cons)):
            z[i] = r2.randrange(fields.R)
        for k, (a, b, c) in enumerate(cons):
            av = sum(v * z[i] for i, v in a.items()) % fields.R
            bv = sum(v * z[i] for i, v in b.items()) % fields.R
            z[n_wires - len(cons) + k] = av * bv % fields.R
        zs.append(np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in z), dtype=np.uint8).reshape(-1, 32))
        rs.append(np.frombuffer(rnd.randrange(fields.R).to_bytes(32, "little") + rnd.randrange(fields.R).to_bytes(32, "little"),
                                dtype=np.uint8))
    pks = m.load_key(blob)
    proofs = m.prove_batch(pks, np.stack(zs), np.stack(rs))
    ck = oc.prepared_key_from_blob(blob)
    for t in range(7):
        r, s = int.from_bytes(rs[t][:32].tobytes(), "little"), int.from_bytes(rs[t][32:].tobytes(), "little")
        assert proofs[t].tobytes() == ck.prove(zs[t], r, s)
    # an unsatisfied witness on the last device is reported with its device
    bad = np.stack(zs).copy()
    bad[6, n_wires - 1, 0] ^= 1
    from owshen_amd import api
    with pytest.raises(api.OwshenGpuError, match="device 2"):
        m.prove_batch(pks, bad, np.stack(rs))
    m.free_key(pks)
    ctx.close()


def test_emu_multi_withdraw_prove_batch(emu3):
    from oracle.c import binding as oc
    from owshen_amd import circuit, groth16 as g16
    emu, m = emu3
    ctx = emu.Ctx()
    depth, n_pad3, n_pad2 = 1, 2, 3
    r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2)
    blob, _vk = g16.setup(ctx, r1, 21, 22, 23, 24, 25)
    rnd = random.Random(2)
    recs = np.stack([circuit.pack_inputs(rnd.randrange(fields.R), rnd.randrange(fields.R), 5, 6, rnd.randrange(fields.R), rnd.randrange(2),
                                         [rnd.randrange(fields.R)], token=rnd.randrange(1 << 160), chain_id=1387) for _ in range(4)])
    rs = _rand_fr(np.random.default_rng(1), 4, 2).reshape(4, 64)
    pks = m.load_key(blob)
    got = m.withdraw_prove_batch(pks, depth, recs, rs, n_pad3, n_pad2)
    wit = circuit.witness(ctx, depth, ctx.to_device(recs), n_pad3, n_pad2)
    ck = oc.prepared_key_from_blob(blob)
    for t in range(4):
        r, s = int.from_bytes(rs[t][:32].tobytes(), "little"), int.from_bytes(rs[t][32:].tobytes(), "little")
        assert got[t].tobytes() == ck.prove(wit[t], r, s)
    m.free_key(pks)
    ctx.close()


def test_emu_single_ctx_call_binds_its_own_device(emu3):
    """og_multi_init leaves the calling thread on the LAST device; a direct og_* call on og_multi_ctx(m, 1) must
    bind device 1 before it allocates scratch or launches (the interpreter records the current device and where the last
    hipMalloc landed), and og_multi_* calls must hand the caller's device back."""
    import ctypes as C
    from oracle.c import binding as oc
    emu, m = emu3
    lib = emu.lib
    lib.emu_current_device.restype = C.c_int
    lib.emu_last_malloc_device.restype = C.c_int
    lib.og_multi_ctx.restype = C.c_void_p
    lib.og_multi_ctx.argtypes = [C.c_void_p, C.c_int]
    rng = np.random.default_rng(5)
    n = 40
    pts = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), _rand_fr(rng, n))
    sc = _rand_fr(rng, n)
    want = oc.msm_g1(pts, sc).tobytes()
    for rank in (1, 2, 0):
        lib.emu_set_device(2 - rank if rank != 1 else 0)       # the caller sits on some OTHER device
        h = lib.og_multi_ctx(m._h, rank)
        assert h
        bases = C.c_void_p()
        assert lib.og_bases_create_d(h, 1, pts.ctypes.data_as(C.c_void_p), n, 8, 0, C.byref(bases)) == 0
        out = np.zeros(64, dtype=np.uint8)
        assert lib.og_msm_d(h, bases, sc.ctypes.data_as(C.c_void_p), n, 1, n * 32, out.ctypes.data_as(C.c_void_p)) == 0
        assert out.tobytes() == want
        assert lib.emu_current_device() == rank and lib.emu_last_malloc_device() == rank
        lib.og_bases_free(bases)
    # og_multi_* restores the caller's device
    lib.emu_set_device(1)
    b = m.bases(1, pts, 8, False)
    assert m.msm(b, sc).tobytes() == want and lib.emu_current_device() == 1
    m.free_bases(b)
