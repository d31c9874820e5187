"""GPU parity: Pippenger MSM over G1 / G2 through the C ABI, bit-exact against
the oracles; plus the size-independent known-answer construction of SURVEY 8c(ii)."""
import random

import numpy as np
import pytest
import torch

from oracle.py import fields
from oracle.py.curve import G1, G2, G1_GEN, G2_GEN, g1_to_bytes, g2_to_bytes, g1_from_bytes, g2_from_bytes

pytestmark = pytest.mark.gpu


def _tob(vals):
    from owshen_amd import api
    return api.ints_to_bytes(vals)


def _g1_bases(ks):
    return np.frombuffer(b"".join(g1_to_bytes(G1.mul(G1_GEN, k) if k else None) for k in ks), dtype=np.uint8).reshape(-1, 64).copy()


def _g2_bases(ks):
    return np.frombuffer(b"".join(g2_to_bytes(G2.mul(G2_GEN, k) if k else None) for k in ks), dtype=np.uint8).reshape(-1, 128).copy()


@pytest.mark.parametrize("window,precomp", [(8, False), (8, True), (12, False), (15, True), (16, False), (16, True), (17, True)])
def test_msm_g1_small_vs_python_oracle(ctx, window, precomp):
    from owshen_amd import api
    rnd = random.Random(11 + window)
    n = 37
    ks = [rnd.randrange(1, fields.R) for _ in range(n)]
    ks[4] = 0  # a base at infinity
    ks[9] = ks[8]  # repeated base
    sc = [rnd.randrange(fields.R) for _ in range(n)]
    sc[0], sc[1], sc[2], sc[3] = 0, 1, fields.R - 1, 1 << 253
    sc[8] = sc[9] = 5  # same base, same digit -> doubling inside a bucket
    sc[10], ks[10] = fields.R - 5, ks[8]  # P and -P meeting in a bucket
    bases = api.Bases(ctx, 1, ctx.to_device(_g1_bases(ks)), window, precomp)
    got = bases.msm(ctx.to_device(_tob(sc)))
    want = G1.msm_naive(sc, [G1.mul(G1_GEN, k) if k else None for k in ks])
    assert g1_from_bytes(got[0].tobytes()) == want


@pytest.mark.parametrize("window,precomp", [(8, False), (15, True), (16, True), (17, True)])
def test_msm_g2_small_vs_python_oracle(ctx, window, precomp):
    from owshen_amd import api
    rnd = random.Random(21 + window)
    n = 19
    ks = [rnd.randrange(1, fields.R) for _ in range(n)]
    ks[3] = 0
    sc = [rnd.randrange(fields.R) for _ in range(n)]
    sc[0], sc[1], sc[2] = 0, 1, fields.R - 1
    bases = api.Bases(ctx, 2, ctx.to_device(_g2_bases(ks)), window, precomp)
    got = bases.msm(ctx.to_device(_tob(sc)))
    want = G2.msm_naive(sc, [G2.mul(G2_GEN, k) if k else None for k in ks])
    assert g2_from_bytes(got[0].tobytes()) == want


def test_msm_edge_cases(ctx):
    from owshen_amd import api
    ks = [1, 2, 3, 4]
    bases = api.Bases(ctx, 1, ctx.to_device(_g1_bases(ks)), 8, False)
    # all-zero scalars -> infinity (zero bytes)
    assert not bases.msm(ctx.to_device(_tob([0, 0, 0, 0]))).any()
    # empty MSM
    assert not bases.msm(ctx.empty(0, 32)).any()
    # prefix MSM (n < number of bases)
    got = bases.msm(ctx.to_device(_tob([7, 9])))
    assert g1_from_bytes(got[0].tobytes()) == G1.mul(G1_GEN, 7 * 1 + 9 * 2)
    # result = infinity by cancellation: 2*G*3 + 3*G*(r-2)
    got = bases.msm(ctx.to_device(_tob([0, 3, fields.R - 2, 0])))
    assert not got.any()


def test_msm_batch_and_heavy_bucket(ctx):
    """batch of 3 scalar vectors; one is all ones (a single > HEAVY bucket: the boolean-wire shape)."""
    from owshen_amd import api
    from oracle.c import binding as oc
    n = 5000
    rng = np.random.default_rng(5)
    ks = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x1F
    gen = np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8)
    bases_np = oc.fixed_base_g1(gen, ks)
    sc = rng.integers(0, 256, (3, n, 32), dtype=np.uint8)
    sc[:, :, 31] &= 0x1F
    sc[1] = 0
    sc[1, :, 0] = 1  # all scalars = 1
    sc[2, ::2] = 0   # half zeros
    for window, precomp in ((12, False), (16, True)):
        bases = api.Bases(ctx, 1, ctx.to_device(bases_np), window, precomp)
        got = bases.msm(ctx.to_device(sc))
        for g in range(3):
            assert got[g].tobytes() == oc.msm_g1(bases_np, sc[g]).tobytes(), (window, precomp, g)


def test_msm_g1_known_answer_2_18(ctx):
    """8c(ii): bases P_i = a_i G with known a_i  =>  MSM(s, P) == (sum a_i s_i mod r) G.
    O(N) check, valid at any size; here N = 2^18 with the C oracle as a second witness."""
    from owshen_amd import api
    from oracle.c import binding as oc
    n = 1 << 18
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    a[:, 31] &= 0x0F
    s = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    s[:, 31] &= 0x0F
    s[::10] = 0
    s[5::10] = 0
    s[5::10, 0] = 1  # ~10% boolean wires, as in a real witness
    gen = np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8)
    bases_np = oc.fixed_base_g1(gen, a)
    ai = api.bytes_to_ints(a)
    si = api.bytes_to_ints(s)
    k = sum(x * y for x, y in zip(ai, si)) % fields.R
    want = g1_to_bytes(G1.mul(G1_GEN, k))
    for precomp in (False, True):
        bases = api.Bases(ctx, 1, ctx.to_device(bases_np), 16, precomp)
        got = bases.msm(ctx.to_device(s))
        assert got[0].tobytes() == want, precomp
        bases.close()


def test_msm_g1_2_26_known_answer_microbench(ctx):
    """BASELINE.json configs[2]: one G1 MSM over 2^26 points on one GPU.  Bases P_i = a_i G, so MSM(s, P) must equal
    (sum a_i s_i mod r) G (size-independent known answer) -- and no leg of that answer comes from the library
    under test (VERDICT r4 "weak" 3): the dot product is taken on the HOST (numpy half-limb products, exact; bench.host_dot_mod_r),
    k G is computed by the C restatement, and 4096 randomly chosen bases of the GPU-generated table are compared with the C
    restatement's a_i G.  The GPU's own dot product (element-wise product, output 0 of a size-2^26 NTT) must agree with the
    host's: a 2^26-point check of the field layer and the NTT on the side.
    Timing goes to gpurun_out/msm_2_26.json (96 B/point algorithmic bytes)."""
    import json
    import os
    import time
    from bench import host_dot_mod_r
    from oracle.c import binding as oc
    from owshen_amd import api, groth16
    log_n = 26
    n = 1 << log_n
    g = torch.Generator(device="cuda").manual_seed(26)
    a = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    a[:, 31] &= 0x0F
    s = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
    s[:, 31] &= 0x0F
    s[::16] = 0          # some zero scalars
    s[1::16] = 0
    s[1::16, 0] = 1      # and a heavy bucket of ones
    t0 = time.time()
    pts = ctx.scalar_mul(1, groth16.G1_GEN_BYTES, a)
    t_gen = time.time() - t0
    a_h, s_h = a.cpu().numpy(), s.cpu().numpy()
    k = host_dot_mod_r(a_h, s_h)                                    # oracle side: the host
    gen = np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8)
    want = oc.fixed_base_g1(gen, api.ints_to_bytes([k]))[0]         # oracle side: the C restatement
    idx = np.random.default_rng(26).choice(n, 4096, replace=False)
    idx[:4] = (0, 1, n - 2, n - 1)
    assert pts[torch.from_numpy(idx).to(pts.device)].cpu().numpy().tobytes() == oc.fixed_base_g1(gen, a_h[idx]).tobytes(), \
        "GPU-generated bases differ from the C restatement's a_i G"
    del a_h, s_h
    prod = ctx.field_op(api.FR, "mul", a, s)
    assert api.bytes_to_ints(ctx.ntt(prod)[0:1].cpu().numpy())[0] == k, "GPU dot product (field_op + NTT) differs from the host's"
    del prod
    # plain bases first (round 4: the two-level (window, bucket) sort, one lane per bucket, nothing precomputed) ...
    plain = api.Bases(ctx, 1, pts, 16, False)
    got_plain = plain.msm(s)
    t0 = time.time()
    got_plain2 = plain.msm(s)
    dt_plain = time.time() - t0
    plain.close()
    ctx.release_scratch()
    assert got_plain[0].tobytes() == want.tobytes() == got_plain2[0].tobytes(), "plain-bases 2^26 MSM differs from the known answer"
    # ... then round 3's per-window tables
    t0 = time.time()
    bases = api.Bases(ctx, 1, pts, 16, True)
    t_tab = time.time() - t0
    del pts
    got = bases.msm(s)
    ctx.profile(True)
    t0 = time.time()
    got2 = bases.msm(s)
    dt = time.time() - t0
    prof = ctx.profile_read()
    ctx.profile(False)
    bases.close()
    assert got[0].tobytes() == want.tobytes() == got2[0].tobytes()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/msm_2_26.json", "w") as f:
        json.dump({"n": n, "msm_seconds_plain_bases": dt_plain, "msm_seconds": dt, "points_per_s": n / dt, "algorithmic_GBps": n * 96 / dt / 1e9,
                   "base_generation_s": t_gen, "window_tables_s": t_tab, "stages_ms": {k2: v[0] for k2, v in prof.items()}}, f, indent=1)


@pytest.mark.parametrize("group", [1, 2])
def test_msm_launch_forms_agree_and_match_c_oracle(ctx_hooks, group, monkeypatch):
    """the persistent bucket-accumulation kernels (default; 1 resident workgroup per CU = every wave walks hundreds of work
    items) against the one-workgroup-per-64-buckets launch of rounds 1-2 (OG_ACC_WAVES_* = 0) and the C restatement:
    2^15 points, 16-bit windows with precomputed tables, a batch of 6 scalar vectors with zero / one / boolean runs"""
    ctx = ctx_hooks  # (OG_ACC_WAVES_*: hooks build)
    from owshen_amd import api
    from oracle.c import binding as oc
    n, batch = 1 << 15, 6
    rng = np.random.default_rng(100 + group)
    ks = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x1F
    gen = np.frombuffer(g1_to_bytes(G1_GEN) if group == 1 else g2_to_bytes(G2_GEN), dtype=np.uint8)
    pts = (oc.fixed_base_g1 if group == 1 else oc.fixed_base_g2)(gen, ks)
    sc = rng.integers(0, 256, (batch, n, 32), dtype=np.uint8)
    sc[:, :, 31] &= 0x1F
    sc[:, ::9] = 0
    sc[:, 1::9, 1:] = 0
    sc[:, 1::9, 0] &= 1                       # boolean wires: one heavy bucket per vector
    bases = api.Bases(ctx, group, ctx.to_device(pts), 16, True)
    sc_d = ctx.to_device(sc)
    b17 = api.Bases(ctx, group, ctx.to_device(pts), 17, True)     # the window the prover picks for its large queries
    out17 = b17.msm(sc_d)
    b17.close()
    for w in (15,):                                               # ... and for queries of 8 k .. 72 k points (this one: 2^15)
        bw = api.Bases(ctx, group, ctx.to_device(pts), w, True)
        assert bw.msm(sc_d).tobytes() == out17.tobytes(), w
        bw.close()
    outs = {}
    for waves in ("0", "1", None):
        for var in ("OG_ACC_WAVES_G1", "OG_ACC_WAVES_G2"):
            if waves is None:
                monkeypatch.delenv(var, raising=False)
            else:
                monkeypatch.setenv(var, waves)
        outs[waves] = bases.msm(sc_d)
    assert outs["0"].tobytes() == outs["1"].tobytes() == outs[None].tobytes() == out17.tobytes()
    ref = oc.msm_g1 if group == 1 else oc.msm_g2
    for g in (0, batch - 1):
        assert outs[None][g].tobytes() == ref(pts, sc[g]).tobytes()
    bases.close()


@pytest.mark.parametrize("group,window", [(1, 15), (1, 16), (1, 17), (2, 15), (2, 16), (2, 17)])
def test_scan_shaped_reduction_equals_segmented(ctx_hooks, group, window, monkeypatch):
    """the two bucket reductions (k_seg_runacc / k_seg_carry and k_scan_reduce) on the same bucket sets: 2^15 and 2^16 buckets,
    G1 and G2, three scalar vectors -- and the C restatement as the referee"""
    ctx = ctx_hooks  # (OG_SCAN_REDUCE: hooks build)
    from owshen_amd import api, groth16
    from oracle.c import binding as oc
    n = 3000 if group == 1 else 600
    rng = np.random.default_rng(window + group)
    ks = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x1F
    gen = groth16.G1_GEN_BYTES if group == 1 else groth16.G2_GEN_BYTES
    bases_np = ctx.scalar_mul(group, gen, ctx.to_device(ks)).cpu().numpy()
    sc = rng.integers(0, 256, (3, n, 32), dtype=np.uint8)
    sc[:, :, 31] &= 0x1F
    sc[1, ::3] = 0
    sc[2] = 0
    sc[2, :, 0] = 1
    b = api.Bases(ctx, group, ctx.to_device(bases_np), window, True)
    monkeypatch.setenv("OG_SCAN_REDUCE", "1")
    scan = b.msm(ctx.to_device(sc))
    monkeypatch.setenv("OG_SCAN_REDUCE", "0")
    seg = b.msm(ctx.to_device(sc))
    b.close()
    assert scan.tobytes() == seg.tobytes()
    for g in range(3):
        want = oc.msm_g1(bases_np, sc[g]) if group == 1 else oc.msm_g2(bases_np, sc[g])
        assert scan[g].tobytes() == want.tobytes()


@pytest.mark.parametrize("window", [16, 17])
def test_g2_batched_affine_accumulation_equals_default(ctx_hooks, window, monkeypatch):
    """OG_G2_AFFINE=1 (k_accumulate_affine: the G2 buckets summed by batched affine additions, 128 buckets per lane and one
    inversion per lane and round; measured, not the default -- DESIGN.md 4.4): same bytes as the XYZZ kernel and as the C
    restatement, with repeated bases (the tangent case), a base and its negative, bases at infinity, zero / one scalars"""
    ctx = ctx_hooks  # (OG_G2_AFFINE: hooks build -- the kernel is not in the shipped library)
    from owshen_amd import api, groth16
    from oracle.c import binding as oc
    n = 900
    rng = np.random.default_rng(window + 40)
    ks = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x1F
    bases_np = ctx.scalar_mul(2, groth16.G2_GEN_BYTES, ctx.to_device(ks)).cpu().numpy()
    bases_np[1:40:2] = bases_np[0:40:2]                     # the same base twice
    neg = bases_np[40:80:2].copy()                          # ... and a base with its negative: y -> q - y, limb-wise on the two halves
    q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    for row in neg:
        for off in (64, 96):
            y = int.from_bytes(row[off:off + 32].tobytes(), "little")
            row[off:off + 32] = np.frombuffer(((q - y) % q).to_bytes(32, "little"), dtype=np.uint8)
    bases_np[41:80:2] = neg
    bases_np[100:104] = 0                                   # bases at infinity
    sc = rng.integers(0, 256, (3, n, 32), dtype=np.uint8)
    sc[:, :, 31] &= 0x1F
    sc[:, 1:80:2] = sc[:, 0:80:2]                           # equal scalars: the pairs meet in every window's bucket
    sc[1, ::3] = 0
    sc[2] = 0
    sc[2, :, 0] = 1
    b = api.Bases(ctx, 2, ctx.to_device(bases_np), window, True)
    monkeypatch.setenv("OG_G2_AFFINE", "1")
    aff = b.msm(ctx.to_device(sc))
    monkeypatch.setenv("OG_G2_AFFINE", "0")
    ref = b.msm(ctx.to_device(sc))
    b.close()
    assert aff.tobytes() == ref.tobytes()
    for g in range(3):
        assert aff[g].tobytes() == oc.msm_g2(bases_np, sc[g]).tobytes()


@pytest.mark.parametrize("v1", [False, True])
def test_msm_lone_sort_with_skewed_bins_vs_c_oracle(ctx_hooks, v1, monkeypatch):
    """the lone-MSM sort (plain bases, n >= 2^18) on scalars that put most of their digits into a few bins: half of the 2^19
    scalars share one 32-bit value (two bins of 2^18 entries: above SB_SLICE, so the second level cuts them into slices that
    claim their runs with global atomics, and every wave of the first level hammers one LDS counter -- the wave-aggregated
    increment, which the CPU interpreter does not model), a run of ones, zeros, r - 1.  Both first levels -- round 5's digit
    arrays with LDS-staged runs (default) and round 4's direct scatter (OG_LONE_SORT_V1, hooks build) -- against the C
    restatement's MSM over the same points."""
    from owshen_amd import api
    from oracle.c import binding as oc
    ctx = ctx_hooks
    n = 1 << 19
    rng = np.random.default_rng(519)
    ks = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    ks[:, 31] &= 0x1F
    pts = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), ks)
    sc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sc[:, 31] &= 0x1F
    sc[::2, :] = 0
    sc[::2, :4] = np.frombuffer((0x7ABC1234).to_bytes(4, "little"), dtype=np.uint8)   # 2^18 equal scalars: digits (0x1234, 0x7ABC)
    sc[1::16] = 0
    sc[1::16, 0] = 1
    sc[3::64] = 0
    sc[5] = np.frombuffer((fields.R - 1).to_bytes(32, "little"), dtype=np.uint8)
    if v1:
        monkeypatch.setenv("OG_LONE_SORT_V1", "1")
    bases = api.Bases(ctx, 1, ctx.to_device(pts), 16, False)
    got = bases.msm(ctx.to_device(sc))
    bases.close()
    assert got[0].tobytes() == oc.msm_g1(pts, sc).tobytes()
