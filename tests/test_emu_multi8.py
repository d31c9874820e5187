"""og_multi_* with EIGHT pretend devices on the CPU interpreter -- the shape of BASELINE.json configs[3] (one 8 x MI355X node;
the driver owns the real one).  Slicing of a 4096-proof batch (and of batches 8 does not divide), proofs over 8 devices with
devices that get nothing, the window-sharded MSM with uneven window ownership (22 / 32 / 15 windows over 8 owners), and failure
injection: one device's part of a call fails => the error names the device, an open RCCL group is closed, the next call works."""
import os
import random

import numpy as np
import pytest

from oracle.py import fields
from oracle.py.curve import G1_GEN, g1_to_bytes


@pytest.fixture(scope="module")
def emu8():
    os.environ["OG_EMU_DEVICES"] = "8"
    os.environ["OG_MULTI_SEQUENTIAL"] = "1"   # the interpreter is single-threaded: ranks run in turn
    from tests import emu
    from owshen_amd import multi
    m = multi.Multi(8, lib=emu.lib)
    yield emu, m
    m.close()
    os.environ.pop("OG_EMU_DEVICES", None)
    os.environ.pop("OG_MULTI_SEQUENTIAL", None)
    os.environ.pop("OG_MULTI_FAIL", None)


def _rand_fr(rng, *shape):
    a = rng.integers(0, 256, (*shape, 32), dtype=np.uint8)
    a[..., 31] &= 0x1F
    return a


@pytest.mark.parametrize("n", [4096, 4099, 4103, 8, 5, 1, 0])
def test_emu_multi8_slices_partition_the_batch(emu8, n):
    """configs[3]: 4096 proofs over 8 devices = 512 each; batches 8 does not divide differ by at most one; contiguous, in order"""
    _emu, m = emu8
    assert m.size == 8
    sl = [m.slice(n, r) for r in range(8)]
    assert sl[0][0] == 0 and sl[-1][1] == n
    assert all(sl[r][1] == sl[r + 1][0] for r in range(7))
    sizes = [hi - lo for lo, hi in sl]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    if n == 4096:
        assert sizes == [512] * 8


def _small_proof_case(emu, n_proofs):
    from owshen_amd import groth16 as g16
    from tests.r1cs_util import random_r1cs
    ctx = emu.Ctx()
    n_wires, cons, z0 = random_r1cs(6, 1, seed=8)
    blob, _vk = g16.setup(ctx, g16.R1CS.from_constraints(n_wires, 1, cons), 3, 5, 7, 11, 13)
    ctx.close()
    rnd = random.Random(80)
    zs, rs = [], []
    for t in range(n_proofs):
        z = list(z0)
        r2 = random.Random(700 + t)
        for i in range(1, n_wires - len(cons)):
            z[i] = r2.randrange(fields.R)
        for k, (a, b, c) in enumerate(cons):
            av = sum(v * z[i] for i, v in a.items()) % fields.R
            bv = sum(v * z[i] for i, v in b.items()) % fields.R
            z[n_wires - len(cons) + k] = av * bv % fields.R
        zs.append(np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in z), dtype=np.uint8).reshape(-1, 32))
        rs.append(np.frombuffer(rnd.randrange(fields.R).to_bytes(32, "little") + rnd.randrange(fields.R).to_bytes(32, "little"),
                                dtype=np.uint8))
    return blob, np.stack(zs), np.stack(rs)


def test_emu_multi8_prove_batch_uneven_and_short_batches(emu8):
    """19 proofs over 8 devices (3,3,3,2,2,2,2,2), then 5 proofs (three devices sit the call out): the caller's order, the C
    restatement's bytes; an unsatisfied witness is reported with its device"""
    from oracle.c import binding as oc
    from owshen_amd import api
    emu, m = emu8
    blob, zs, rs = _small_proof_case(emu, 19)
    pks = m.load_key(blob)
    ck = oc.prepared_key_from_blob(blob)
    want = [ck.prove(zs[t], int.from_bytes(rs[t][:32].tobytes(), "little"), int.from_bytes(rs[t][32:].tobytes(), "little")) for t in range(19)]
    got = m.prove_batch(pks, zs, rs)
    assert [got[t].tobytes() for t in range(19)] == want
    got5 = m.prove_batch(pks, zs[:5], rs[:5])
    assert [got5[t].tobytes() for t in range(5)] == want[:5]
    bad = zs.copy()
    bad[18, -1, 0] ^= 1                       # proof 18 belongs to device 7
    with pytest.raises(api.OwshenGpuError, match="device 7"):
        m.prove_batch(pks, bad, rs)
    m.free_key(pks)


@pytest.mark.parametrize("window,precomp", [(12, False), (8, True), (16, True)])
def test_emu_multi8_msm_uneven_window_ownership(emu8, window, precomp):
    """22 windows (12 bits) / 32 (8 bits) / 16 (16 bits) over 8 owners, round robin: 3,3,3,3,3,3,2,2 / 4 each / 2 each"""
    from oracle.c import binding as oc
    _emu, m = emu8
    rng = np.random.default_rng(window)
    n = 120
    bases_np = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), _rand_fr(rng, n))
    sc = _rand_fr(rng, n)
    sc[:3] = 0
    sc[1, 0] = 1
    sc[2] = np.frombuffer((fields.R - 1).to_bytes(32, "little"), dtype=np.uint8)
    b = m.bases(1, bases_np, window, precomp)
    assert m.msm(b, sc).tobytes() == oc.msm_g1(bases_np, sc).tobytes()
    m.free_bases(b)


@pytest.mark.parametrize("site,rank", [("msm.scratch", 2), ("msm.broadcast", 5), ("msm.accumulate", 7), ("msm.allgather", 3)])
def test_emu_multi8_failure_injection_msm(emu8, site, rank, monkeypatch):
    """one device's part of og_multi_msm fails (before, inside and between the two grouped RCCL exchanges): the error names the
    device, nothing is left open -- the very next call on the same og_multi gives the right answer"""
    from oracle.c import binding as oc
    from owshen_amd import api
    _emu, m = emu8
    rng = np.random.default_rng(3)
    n = 60
    bases_np = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), _rand_fr(rng, n))
    sc = _rand_fr(rng, n)
    b = m.bases(1, bases_np, 8, False)
    want = oc.msm_g1(bases_np, sc).tobytes()
    assert m.msm(b, sc).tobytes() == want
    monkeypatch.setenv("OG_MULTI_FAIL", f"{site}:{rank}")
    with pytest.raises(api.OwshenGpuError, match=f"device {rank}: .*injected failure at {site}"):
        m.msm(b, sc)
    monkeypatch.delenv("OG_MULTI_FAIL")
    assert m.msm(b, sc).tobytes() == want
    m.free_bases(b)


def test_emu_multi8_failure_injection_prove_and_key_load(emu8, monkeypatch):
    from oracle.c import binding as oc
    from owshen_amd import api
    emu, m = emu8
    blob, zs, rs = _small_proof_case(emu, 9)
    monkeypatch.setenv("OG_MULTI_FAIL", "pk_load:4")
    with pytest.raises(api.OwshenGpuError, match="device 4: .*injected failure at pk_load"):
        m.load_key(blob)                                   # (the replicas already loaded are released)
    monkeypatch.delenv("OG_MULTI_FAIL")
    pks = m.load_key(blob)
    monkeypatch.setenv("OG_MULTI_FAIL", "prove:6")
    with pytest.raises(api.OwshenGpuError, match="device 6: .*injected failure at prove"):
        m.prove_batch(pks, zs, rs)
    monkeypatch.delenv("OG_MULTI_FAIL")
    ck = oc.prepared_key_from_blob(blob)
    got = m.prove_batch(pks, zs, rs)
    for t in range(9):
        assert got[t].tobytes() == ck.prove(zs[t], int.from_bytes(rs[t][:32].tobytes(), "little"), int.from_bytes(rs[t][32:].tobytes(), "little"))
    m.free_key(pks)


def test_emu_multi8_device_info_is_the_runtimes_account(emu8):
    """og_multi_device_info: rank g is bound to device g, and the RCCL stand-in's communicator has 8 ranks with g as rank g --
    what bench.py --in-process quotes so that "8 ranks on 8 GPUs" is a reading"""
    _emu, m = emu8
    infos = [m.device_info(r) for r in range(8)]
    assert [i["device"] for i in infos] == list(range(8))
    assert all(i["comm_nranks"] == 8 for i in infos) and [i["comm_rank"] for i in infos] == list(range(8))
    assert len({i["pci"] for i in infos}) == 8
    with pytest.raises(Exception):
        m.device_info(8)


def test_bench_in_process_path_over_8_pretend_devices(monkeypatch):
    """bench.py's own --in-process code path (og_multi_withdraw_prove_batch from one process) with the interpreter as the eight
    devices: a batch-total that 8 does not divide (9 = 2 1 1 1 1 1 1 1), two alternating input sets, every proof verified,
    the line's per-rank batches / devices / communicator size filled from the library; and the arithmetic of BASELINE.json
    configs[3] (--gpus 8 --batch-total 4096 = 512 per rank) through the same two rules the modes use"""
    import argparse
    import importlib.util
    monkeypatch.setenv("OG_EMU_DEVICES", "8")
    monkeypatch.setenv("OG_MULTI_SEQUENTIAL", "1")
    monkeypatch.setenv("OG_WITNESS_W9", "0")   # (18 witnesses over 8 pretend devices: the lane-local walk is the interpreter's fast one)
    from tests import emu
    from tests.test_bench_contract import BENCH, _FakeDist
    from owshen_amd import multi
    spec = importlib.util.spec_from_file_location("bench_in_process", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert [bench.rank_batch(4096, 8, r) for r in range(8)] == [512] * 8
    assert [bench.rank_batch(4099, 8, r) for r in range(8)] == [513, 513, 513, 512, 512, 512, 512, 512]
    m_probe = multi.Multi(8, lib=emu.lib)
    assert [m_probe.slice(4096, r) for r in range(8)] == [(512 * r, 512 * (r + 1)) for r in range(8)]
    m_probe.close()
    args = argparse.Namespace(gpus=8, batch=2, batch_total=9, depth=1, natural=True, sparse=False, dense=False, steps=2, warmup=0, no_verify=False,
                              in_process=True)
    ctx = emu.Ctx()
    line = bench.run_prove_in_process(args, _FakeDist(), ctx, make_multi=lambda n: multi.Multi(n, lib=emu.lib))
    ctx.close()
    assert line["n_gpus"] == 8 and line["ranks"]["per_rank_batch"] == [2, 1, 1, 1, 1, 1, 1, 1] and line["config"]["batch_total"] == 9
    assert line["ranks"]["distinct_devices"] == 8 and all(d["comm_nranks"] == 8 for d in line["ranks"]["devices"])
    assert line["repeatability"]["verified"].startswith("9 / 9") and line["value"] > 0


# ---- window-sharded PROVING (round 6: og_multi_prove_sharded / og_multi_withdraw_prove_sharded) -----------------------------------
def _want_proofs(blob, zs, rs):
    from oracle.c import binding as oc
    ck = oc.prepared_key_from_blob(blob)
    return [ck.prove(zs[t], int.from_bytes(rs[t][:32].tobytes(), "little"), int.from_bytes(rs[t][32:].tobytes(), "little")) for t in range(len(zs))]


@pytest.mark.parametrize("mode", ["split", "sym", "pipe", "pipe_heavy"])
def test_emu_multi8_prove_sharded_equals_the_c_restatement(emu8, mode, monkeypatch):
    """ONE batch window-sharded over 8 devices (32 eight-bit windows, 4 per owner): every device walks all the witnesses and
    accumulates only its windows, the partial points meet in the all-gather, device 0 assembles -- the C restatement's bytes, in
    every schedule the front half can take: one request fanned out over the streams (<= 16 proofs: `split`), whole sub-batches
    side by side (`sym`), the stage pipeline with a ramped plan and slot reuse (OG_PIPE_MIN=1 reaches it at toy size), and the
    pipeline with heavy buckets forced in both halves of the merged L + H pair"""
    emu, m = emu8
    n = {"split": 3, "sym": 19, "pipe": 11, "pipe_heavy": 7}[mode]
    blob, zs, rs = _small_proof_case(emu, n)
    if mode == "sym":
        monkeypatch.setenv("OG_SUB_BATCH", "5")
    if mode.startswith("pipe"):
        monkeypatch.setenv("OG_PIPE_MIN", "1")
        monkeypatch.setenv("OG_SUB_BATCH", "3")
    if mode == "pipe_heavy":
        monkeypatch.setenv("OG_HEAVY", "1")
    pks = m.load_key(blob)
    got = m.prove_sharded(pks, zs, rs)
    assert [got[t].tobytes() for t in range(n)] == _want_proofs(blob, zs, rs)
    # and the proof-sharded form on the same og_multi right after it: same bytes (no scratch / call slot left behind)
    assert m.prove_batch(pks, zs, rs).tobytes() == got.tobytes()
    m.free_key(pks)


def test_emu_multi8_withdraw_prove_sharded(emu8, monkeypatch):
    """input records -> proofs, window-sharded: witnesses generated on every device (whole-slab and inside the pipeline), the
    public inputs come back from device 0, a malformed record is refused with its index whichever device looks"""
    from owshen_amd import api, circuit, groth16 as g16
    emu, m = emu8
    # (the witness in the lane-local kernels: on the interpreter every cross-lane read of the wave-wide walk is a rendezvous of the
    # whole workgroup, and this case walks 8 devices x 4 calls -- 130 s with it, and the walk's forms have their own cases)
    monkeypatch.setenv("OG_WITNESS_W9", "0")
    ctx = emu.Ctx()
    depth, n_pad3, n_pad2 = 1, 2, 3
    r1 = circuit.withdraw_r1cs(ctx.mimc7_constants(), depth, n_pad3, n_pad2)
    blob, _vk = g16.setup(ctx, r1, 21, 22, 23, 24, 25)
    rnd = random.Random(2)
    recs = np.stack([circuit.pack_inputs(rnd.randrange(fields.R), rnd.randrange(fields.R), 5, 6, rnd.randrange(fields.R), rnd.randrange(2),
                                         [rnd.randrange(fields.R)], token=rnd.randrange(1 << 160), chain_id=1387) for _ in range(3)])
    rs = _rand_fr(np.random.default_rng(1), 3, 2).reshape(3, 64)
    wit = circuit.witness(ctx, depth, ctx.to_device(recs), n_pad3, n_pad2)
    want = _want_proofs(blob, wit, rs)
    pks = m.load_key(blob)
    got, pub = m.withdraw_prove_sharded(pks, depth, recs, rs, n_pad3, n_pad2, return_public=True)
    assert [got[t].tobytes() for t in range(3)] == want
    assert pub.tobytes() == np.ascontiguousarray(wit[:, 1:7]).tobytes()
    monkeypatch.setenv("OG_GEN_MIN", "1")     # witnesses inside the pipeline
    monkeypatch.setenv("OG_PIPE_MIN", "1")
    monkeypatch.setenv("OG_SUB_BATCH", "2")
    got2 = m.withdraw_prove_sharded(pks, depth, recs, rs, n_pad3, n_pad2)
    assert got2.tobytes() == got.tobytes()
    bad = recs.copy()
    bad[2, 1] = np.frombuffer(fields.R.to_bytes(32, "little"), dtype=np.uint8)   # secret = r: a second encoding of 0
    with pytest.raises(api.OwshenGpuError, match="input record 2: field 1"):
        m.withdraw_prove_sharded(pks, depth, bad, rs, n_pad3, n_pad2)
    assert m.withdraw_prove_sharded(pks, depth, recs, rs, n_pad3, n_pad2).tobytes() == got.tobytes()   # nothing left pending
    m.free_key(pks)
    ctx.close()


@pytest.mark.parametrize("site,rank", [("sharded.front", 5), ("sharded.allgather", 2), ("sharded.finish", 0), ("sharded.finish", 6)])
def test_emu_multi8_failure_injection_prove_sharded(emu8, site, rank, monkeypatch):
    """one device's part of a window-sharded call fails -- in the front half, inside the grouped all-gather, at the end: the
    error names the device, no job stays pending on ANY device, the RCCL group is closed; the next call gives the right bytes"""
    from owshen_amd import api
    emu, m = emu8
    blob, zs, rs = _small_proof_case(emu, 4)
    want = _want_proofs(blob, zs, rs)
    pks = m.load_key(blob)
    monkeypatch.setenv("OG_MULTI_FAIL", f"{site}:{rank}")
    with pytest.raises(api.OwshenGpuError, match=f"device {rank}: .*injected failure at {site}"):
        m.prove_sharded(pks, zs, rs)
    monkeypatch.delenv("OG_MULTI_FAIL")
    got = m.prove_sharded(pks, zs, rs)
    assert [got[t].tobytes() for t in range(4)] == want
    bad = zs.copy()
    bad[2, -1, 0] ^= 1
    with pytest.raises(api.OwshenGpuError, match="witness 2 does not satisfy"):
        m.prove_sharded(pks, bad, rs)
    assert m.prove_sharded(pks, zs, rs).tobytes() == got.tobytes()
    m.free_key(pks)
