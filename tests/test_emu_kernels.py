"""Kernel logic on the CPU interpreter (tests/hipemu) against the oracles.  These are NOT the
parity tests proper (those are `-m gpu` and run the gfx950 binary); they keep the kernels' index
arithmetic, digit decomposition and reduction trees checked where no GPU exists."""
import random

import numpy as np
import pytest

from oracle.py import fields, mimc7
from oracle.py.curve import G1, G2, G1_GEN, G2_GEN, g1_to_bytes, g2_to_bytes, g1_from_bytes, g2_from_bytes


@pytest.fixture(scope="module")
def ectx():
    from tests import emu
    c = emu.Ctx()
    yield c
    c.close()


def _tob(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint8).reshape(-1, 32).copy()


def _toi(arr):
    return [int.from_bytes(r.tobytes(), "little") for r in np.asarray(arr).reshape(-1, 32)]


def _rand_fr_np(rng, *shape):
    a = rng.integers(0, 256, (*shape, 32), dtype=np.uint8)
    a[..., 31] &= 0x1F
    return a


def test_emu_field_and_mimc7(ectx):
    rnd = random.Random(3)
    a = [rnd.randrange(fields.R) for _ in range(20)] + [0, fields.R - 1]
    b = [rnd.randrange(fields.R) for _ in range(20)] + [fields.R - 1, fields.R - 1]
    assert _toi(ectx.field_op(0, "mul", _tob(a), _tob(b))) == [x * y % fields.R for x, y in zip(a, b)]
    assert _toi(ectx.field_op(1, "sub", _tob(a), _tob(b))) == [(x - y) % fields.P for x, y in zip(a, b)]
    assert _toi(ectx.mimc7_hash2(_tob(a[:4]), _tob(b[:4]))) == [mimc7.hash2(x, y) for x, y in zip(a[:4], b[:4])]


@pytest.mark.parametrize("log_n", [0, 3, 11])
def test_emu_ntt(ectx, log_n):
    from oracle.c import binding as oc
    rng = np.random.default_rng(log_n)
    x = _rand_fr_np(rng, 1 << log_n)
    for inverse in (False, True):
        for coset in (False, True):
            assert ectx.ntt(x, inverse, coset).tobytes() == oc.ntt(x, inverse=inverse, coset=coset).tobytes()


@pytest.mark.parametrize("log_d", [4, 10, 11])
def test_emu_quotient_radix4_equals_radix2_on_extreme_inputs(ectx, log_d, monkeypatch):
    """k_ntt_block4's lazy butterflies (sums grow by up to 3N per stage, limbs reach 2^31 as product operands) against
    the radix-2 kernel (OG_NTT_RADIX4=0: every butterfly reduced) and the C restatement, through the quotient pipeline (the only
    user of the stage-block kernels), on the inputs that maximise every intermediate sum: all r - 1, alternating 0 / r - 1, all 1"""
    from oracle.c import binding as oc
    d = 1 << log_d
    rm1 = np.frombuffer((fields.R - 1).to_bytes(32, "little"), dtype=np.uint8)
    full, alt, ones = np.tile(rm1, (d, 1)), np.tile(rm1, (d, 1)), np.zeros((d, 32), np.uint8)
    alt[::2] = 0
    ones[:, 0] = 1
    for a, b, c in ((full, full, full), (alt, full, ones), (ones, alt, full), (full, alt[::-1].copy(), alt)):
        monkeypatch.delenv("OG_NTT_RADIX4", raising=False)
        r4 = ectx.h_poly(a, b, c).tobytes()
        monkeypatch.setenv("OG_NTT_RADIX4", "0")
        r2 = ectx.h_poly(a, b, c).tobytes()
        assert r4 == r2 == oc.h_poly(a, b, c).tobytes(), log_d


def test_emu_h_poly(ectx):
    from oracle.c import binding as oc
    rng = np.random.default_rng(5)
    a, b, c = (_rand_fr_np(rng, 1 << 11) for _ in range(3))
    assert ectx.h_poly(a, b, c).tobytes() == oc.h_poly(a, b, c).tobytes()


@pytest.mark.parametrize("log_d", [0, 1, 2, 3, 6, 9, 10, 12, 13, 17])
def test_emu_h_poly_block_shapes(ectx, log_d):
    """the radix-4 stage blocks with lazy butterflies (k_ntt_block4): every shape of the two blocks -- one block of 0..10
    stages (even / odd: the leftover radix-2 stage, the twiddle-free global stage 0 alone and inside a double stage), and a
    second block of 1, 2, 3 and 7 stages (2^17: the prover's domain) -- against the C restatement; extreme inputs (0, 1, r - 1) ride along"""
    from oracle.c import binding as oc
    d = 1 << log_d
    rng = np.random.default_rng(40 + log_d)
    a, b, c = (_rand_fr_np(rng, d) for _ in range(3))
    rm1 = np.frombuffer((fields.R - 1).to_bytes(32, "little"), dtype=np.uint8)
    a[0] = rm1
    b[-1] = rm1
    c[d // 2] = rm1
    if d >= 4:
        a[1] = 0
        b[2] = 0
        b[2, 0] = 1
        c[3] = rm1
        a[3] = rm1
        b[3] = rm1
    assert ectx.h_poly(a, b, c).tobytes() == oc.h_poly(a, b, c).tobytes()


@pytest.mark.parametrize("group,window,precomp", [(1, 8, False), (1, 12, True), (2, 8, False), (2, 8, True)])
def test_emu_msm_small(ectx, group, window, precomp):
    rnd = random.Random(40 + group + window)
    n = 23
    G, GEN, tob, fromb = (G1, G1_GEN, g1_to_bytes, g1_from_bytes) if group == 1 else (G2, G2_GEN, g2_to_bytes, g2_from_bytes)
    ks = [rnd.randrange(1, fields.R) for _ in range(n)]
    ks[3] = 0
    ks[9] = ks[8]
    sc = [rnd.randrange(fields.R) for _ in range(n)]
    sc[0], sc[1], sc[2] = 0, 1, fields.R - 1
    sc[8] = sc[9] = 5
    pts = [G.mul(GEN, k) if k else None for k in ks]
    pb = 64 if group == 1 else 128
    bases = np.frombuffer(b"".join(tob(q) for q in pts), dtype=np.uint8).reshape(-1, pb).copy()
    from owshen_amd import api
    got = api.Bases(ectx, group, bases, window, precomp).msm(_tob(sc))
    assert fromb(got[0].tobytes()) == G.msm_naive(sc, pts)


def test_emu_msm_17_bit_windows(ectx):
    """17-bit windows (15 windows, 2^16 buckets, the 8 + 24 bit split of the two-level radix sort): what the prover
    picks for the large proving-key queries.  G1 and G2, zero / one runs and r - 1; plain bases are refused."""
    from owshen_amd import api
    from owshen_amd._lib import OwshenGpuError
    from oracle.c import binding as oc
    rng = np.random.default_rng(17)
    n = 500
    ks = _rand_fr_np(rng, n)
    b1 = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), ks)
    sc = _rand_fr_np(rng, 3, n)
    sc[0, :4] = 0
    sc[1, :50] = 0
    sc[1, :50, 0] = 1
    sc[2, 5] = _tob([fields.R - 1])[0]
    got = api.Bases(ectx, 1, b1, 17, True).msm(sc)
    for g in range(3):
        assert got[g].tobytes() == oc.msm_g1(b1, sc[g]).tobytes()
    n2 = 40
    b2 = oc.fixed_base_g2(np.frombuffer(g2_to_bytes(G2_GEN), dtype=np.uint8), ks[:n2])
    sc2 = np.ascontiguousarray(sc[1:, :n2])
    got = api.Bases(ectx, 2, b2, 17, True).msm(sc2)
    for g in range(2):
        assert got[g].tobytes() == oc.msm_g2(b2, sc2[g]).tobytes()
    with pytest.raises(OwshenGpuError, match="precomputed"):
        api.Bases(ectx, 1, b1, 17, False)


@pytest.mark.parametrize("window", [15])
def test_emu_msm_15_bit_windows(ectx, window, monkeypatch):
    """15-bit windows (17 windows, 2^14 buckets, the 6 + 26 bit split of the two-level radix sort): what the prover picks for
    queries of 8 k .. 72 k points -- the natural depth-32 statement's.  G1 and G2, staged and direct sort kernels, zero / one
    runs and r - 1; plain bases are refused."""
    from owshen_amd import api
    from owshen_amd._lib import OwshenGpuError
    from oracle.c import binding as oc
    rng = np.random.default_rng(window)
    n = 500
    ks = _rand_fr_np(rng, n)
    b1 = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), ks)
    sc = _rand_fr_np(rng, 3, n)
    sc[0, :4] = 0
    sc[1, :50] = 0
    sc[1, :50, 0] = 1
    sc[2, 5] = _tob([fields.R - 1])[0]
    want = [oc.msm_g1(b1, sc[g]).tobytes() for g in range(3)]
    bases = api.Bases(ectx, 1, b1, window, True)
    for direct in ("1", "0"):
        monkeypatch.setenv("OG_SORT_DIRECT", direct)
        got = bases.msm(sc)
        assert [got[g].tobytes() for g in range(3)] == want, direct
    monkeypatch.delenv("OG_SORT_DIRECT")
    n2 = 40
    b2 = oc.fixed_base_g2(np.frombuffer(g2_to_bytes(G2_GEN), dtype=np.uint8), ks[:n2])
    sc2 = np.ascontiguousarray(sc[1:, :n2])
    got = api.Bases(ectx, 2, b2, window, True).msm(sc2)
    for g in range(2):
        assert got[g].tobytes() == oc.msm_g2(b2, sc2[g]).tobytes()
    with pytest.raises(OwshenGpuError, match="precomputed"):
        api.Bases(ectx, 1, b1, window, False)


def test_emu_msm_batch_heavy(ectx):
    from oracle.c import binding as oc
    n = 3000
    rng = np.random.default_rng(5)
    ks = _rand_fr_np(rng, n)
    gen = np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8)
    bases_np = oc.fixed_base_g1(gen, ks)
    sc = _rand_fr_np(rng, 2, n)
    sc[1] = 0
    sc[1, :, 0] = 1
    from owshen_amd import api
    got = api.Bases(ectx, 1, bases_np, 12, False).msm(sc)
    for g in range(2):
        assert got[g].tobytes() == oc.msm_g1(bases_np, sc[g]).tobytes()


@pytest.mark.parametrize("waves", ["0", "1", "12"])
def test_emu_msm_grid_and_persistent_launches_agree(ectx, waves, monkeypatch):
    """OG_ACC_WAVES_G1 / _G2 = 0 selects the one-workgroup-per-64-buckets launch of rounds 1-2, anything else the persistent
    kernels with that many resident workgroups per CU (1: a single wave per CU walks many work items): same points"""
    from owshen_amd import api
    from oracle.c import binding as oc
    monkeypatch.setenv("OG_ACC_WAVES_G1", waves)
    monkeypatch.setenv("OG_ACC_WAVES_G2", waves)
    n = 700
    rng = np.random.default_rng(int(waves) + 9)
    ks = _rand_fr_np(rng, n)
    bases_np = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), ks)
    sc = _rand_fr_np(rng, 3, n)
    sc[0, :4] = 0
    for window, precomp in ((8, False), (12, True)):
        got = api.Bases(ectx, 1, bases_np, window, precomp).msm(sc)
        for g in range(3):
            assert got[g].tobytes() == oc.msm_g1(bases_np, sc[g]).tobytes()
    n2 = 120
    b2 = np.frombuffer(b"".join(g2_to_bytes(G2.mul(G2_GEN, k)) for k in _toi(ks[:n2])), dtype=np.uint8).reshape(-1, 128).copy()
    sc2 = np.ascontiguousarray(sc[:2, :n2])
    got = api.Bases(ectx, 2, b2, 8, True).msm(sc2)
    for g in range(2):
        assert got[g].tobytes() == oc.msm_g2(b2, sc2[g]).tobytes()


@pytest.mark.parametrize("cap", [1, 5, 1000])
def test_emu_msm_heavy_list_overflow(ectx, cap, monkeypatch):
    """more heavy buckets than the heavy list holds: the overflowing ones are accumulated inline by k_accumulate
    (they used to be dropped silently).  OG_HEAVY / OG_HEAVY_CAP make this reachable at toy sizes."""
    from owshen_amd import api
    from oracle.c import binding as oc
    monkeypatch.setenv("OG_HEAVY", "4")
    monkeypatch.setenv("OG_HEAVY_CAP", str(cap))
    n = 1200
    rng = np.random.default_rng(77 + cap)
    ks = _rand_fr_np(rng, n)
    bases_np = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), ks)
    sc = _rand_fr_np(rng, 2, n)       # 8-bit windows: 128 buckets per window, ~9 entries each -> all "heavy"
    for precomp in (False, True):
        got = api.Bases(ectx, 1, bases_np, 8, precomp).msm(sc)
        for g in range(2):
            assert got[g].tobytes() == oc.msm_g1(bases_np, sc[g]).tobytes()


@pytest.mark.parametrize("group,window,precomp,world", [(1, 8, False, 3), (1, 12, True, 2), (1, 16, False, 8), (2, 8, False, 2),
                                                        (1, 8, False, 40)])
def test_emu_msm_window_sharded(ectx, group, window, precomp, world):
    """rank g accumulates the windows k = g (mod world); the ranks' per-window points are concatenated (the
    all-gather) and combined by Horner.  All ranks are played in turn by one interpreter ctx.  world = 40 > nwin leaves
    ranks without any window."""
    from owshen_amd import api
    from oracle.c import binding as oc
    n = 150 if group == 1 else 40
    rng = np.random.default_rng(window + world)
    ks = _rand_fr_np(rng, n)
    if group == 1:
        bases_np = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), ks)
    else:
        kint = _toi(ks)
        bases_np = np.frombuffer(b"".join(g2_to_bytes(G2.mul(G2_GEN, k)) for k in kint), dtype=np.uint8).reshape(-1, 128).copy()
    sc = _rand_fr_np(rng, n)
    sc[0] = 0
    sc[1] = 0
    sc[1, 0] = 1
    sc[2] = np.frombuffer((fields.R - 1).to_bytes(32, "little"), dtype=np.uint8)
    b = api.Bases(ectx, group, bases_np, window, precomp)
    want = b.msm(sc)[0]
    parts = [b.msm_windows(sc, r, world) for r in range(world)]
    assert all(p.shape[0] == b.partial_bytes() for p in parts)
    got = b.msm_combine(np.concatenate(parts), world)
    assert got.tobytes() == want.tobytes()
    if group == 1:
        assert got.tobytes() == oc.msm_g1(bases_np, sc).tobytes()


@pytest.mark.parametrize("window,nblk", [(8, 3), (12, 7), (16, 5)])
def test_emu_msm_multi_block_scan(ectx, window, nblk, monkeypatch):
    """the slice-sum / slice-base / slice-scan path a lone 2^26 MSM takes, forced on a small instance"""
    from owshen_amd import api
    from oracle.c import binding as oc
    monkeypatch.setenv("OG_SCAN_NBLK", str(nblk))
    n = 700
    rng = np.random.default_rng(window)
    ks = _rand_fr_np(rng, n)
    bases_np = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), ks)
    sc = _rand_fr_np(rng, 2, n)
    sc[1, ::3] = 0
    got = api.Bases(ectx, 1, bases_np, window, True).msm(sc)
    for g in range(2):
        assert got[g].tobytes() == oc.msm_g1(bases_np, sc[g]).tobytes()


@pytest.mark.parametrize("direct,rank,groups", [("0", 7, None), ("0", 7, "1"), ("0", 7, "runs"), ("0", 3, "runs"), ("0", 5, "tiles"), ("0", 6, "vector")])   # (the direct second level is k_sort_lo_direct<5>, the prover's kernel at another width)
def test_emu_msm_lone_plain_bases_sort(ectx, direct, rank, groups, monkeypatch):
    """the two-level (window, bucket) radix sort of a lone big MSM over plain bases (msm.hip, k_lone_hist / k_lone_scatter /
    k_sort_lo<5>), forced on a small instance and -- to keep the interpreter's 2^15-bucket reductions few -- on ONE rank's two
    windows of an 8-way window-sharded MSM: its share (Horner over its own window points) must equal the share the legacy
    global-atomic sort gives.  Staged and direct second level, zero / one / r - 1 scalars, a heavy bucket.  (A second chunk
    needs n > LN_CHUNK = 32768: the whole MSM is covered on hardware at 2^18 and 2^26.)"""
    from owshen_amd import api
    from oracle.c import binding as oc
    n = {"tiles": 2600, "vector": 20000}.get(groups, 600)   # "vector": one default tile of 16 384 digits (two 16-byte loads per lane) + a ragged one
    rng = np.random.default_rng(26)
    ks = _rand_fr_np(rng, n)
    bases_np = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), ks)
    sc = _rand_fr_np(rng, n)
    sc[:40] = 0
    sc[40:200] = 0
    sc[40:200, 0] = 1
    sc[200] = _tob([fields.R - 1])[0]
    b = api.Bases(ectx, 1, bases_np, 16, False)
    want = b.msm_combine(b.msm_windows(sc, rank, 8), 1)      # n < 2^18: the legacy sort
    monkeypatch.setenv("OG_LONE_MIN", "1")
    monkeypatch.setenv("OG_SORT_DIRECT", direct)
    monkeypatch.setenv("OG_SCAN_NBLK", "3")
    # round 5's first level (per-window digit arrays, runs staged in LDS: k_lone_digits / k_lone_scatter_runs) is the default;
    # "runs" forces three chunks of 256 scalars and tiles of 64 digits (several tiles per chunk, a ragged last one), the other
    # cases keep round 4's direct scatter (OG_LONE_SORT_V1): window groups, or ONE pass with every run open
    if groups == "runs":
        monkeypatch.setenv("OG_LONE_CHUNK", "256")
        monkeypatch.setenv("OG_LONE_TILE", "64")
    elif groups == "tiles":                                  # two chunks of 2048 digits, tiles of 1024 (one digit per lane): full, full | ragged
        monkeypatch.setenv("OG_LONE_CHUNK", "2048")
        monkeypatch.setenv("OG_LONE_TILE", "1024")
    elif groups == "vector":                                 # the defaults
        pass
    else:
        monkeypatch.setenv("OG_LONE_SORT_V1", "1")
        if groups:
            monkeypatch.setenv("OG_LONE_WGROUPS", groups)
    got = b.msm_combine(b.msm_windows(sc, rank, 8), 1)
    assert got.tobytes() == want.tobytes() and got.any()


@pytest.mark.parametrize("pieces,heavy", [("5", "20")])
def test_emu_msm_lone_position_major_pieces(ectx, pieces, heavy, monkeypatch):
    """k_accumulate_pieces / k_pieces_combine (a lone big MSM over plain bases: every bucket cut into pieces by entry count,
    pieces handed out position-major), forced on a small instance, one rank's two windows of an 8-way window-sharded MSM:
    with and without buckets that go to the heavy path (OG_HEAVY=20: the run of ones and the zero-digit... buckets)"""
    from owshen_amd import api
    from oracle.c import binding as oc
    n = 900
    rng = np.random.default_rng(27)
    ks = _rand_fr_np(rng, n)
    bases_np = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), ks)
    sc = _rand_fr_np(rng, n)
    sc[:300, 2:] = 0                      # small scalars: a few well-filled buckets in window 0
    sc[:300, 1] &= 0x01
    sc[300:420] = 0
    sc[300:420, 0] = 1                    # a run of ones: one long bucket
    sc[420] = _tob([fields.R - 1])[0]
    b = api.Bases(ectx, 1, bases_np, 16, False)
    want = b.msm_combine(b.msm_windows(sc, 0, 8), 1)       # whole buckets, legacy sort
    monkeypatch.setenv("OG_LONE_MIN", "1")
    monkeypatch.setenv("OG_LONE_AVG", "0.0001")
    monkeypatch.setenv("OG_LONE_PIECES", pieces)
    if heavy:
        monkeypatch.setenv("OG_HEAVY", heavy)
    got = b.msm_combine(b.msm_windows(sc, 0, 8), 1)
    assert got.tobytes() == want.tobytes() and got.any()


@pytest.mark.parametrize("scan", ["0", "1"])
@pytest.mark.parametrize("group,window,precomp", [(1, 12, True), (2, 8, False)])
def test_emu_msm_both_reduction_forms(ectx, scan, group, window, precomp, monkeypatch):
    """the bucket reduction in both forms whatever the launch size would pick: segmented running sums (k_seg_runacc / k_seg_carry:
    the throughput form) and the scan-shaped one (k_scan_reduce x 2 + k_scan_reduce_final: launches of a few bucket sets)"""
    from owshen_amd import api
    from oracle.c import binding as oc
    monkeypatch.setenv("OG_SCAN_REDUCE", scan)
    rng = np.random.default_rng(60 + window)
    n = 90 if group == 1 else 30
    ks = _rand_fr_np(rng, n)
    if group == 1:
        bases_np = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), ks)
    else:
        bases_np = oc.fixed_base_g2(np.frombuffer(g2_to_bytes(G2_GEN), dtype=np.uint8), ks)
    sc = _rand_fr_np(rng, 3, n)
    sc[0, :5] = 0
    sc[1] = 0
    sc[1, :, 0] = 1
    sc[2, 7] = _tob([fields.R - 1])[0]
    got = api.Bases(ectx, group, bases_np, window, precomp).msm(sc)
    for g in range(3):
        want = oc.msm_g1(bases_np, sc[g]) if group == 1 else oc.msm_g2(bases_np, sc[g])
        assert got[g].tobytes() == want.tobytes(), (scan, g)


@pytest.mark.parametrize("window,precomp,heavy", [(8, True, None), (8, False, "12"), (16, True, None)])
def test_emu_msm_g2_batched_affine_accumulation(ectx, window, precomp, heavy, monkeypatch):
    """OG_G2_AFFINE=1: the G2 buckets are summed by batched affine additions (k_accumulate_affine: 128 buckets per lane, one
    inversion per lane and round) instead of XYZZ mixed additions.  Same bytes as the C restatement and as the default kernel
    -- with everything an affine formula has to special-case: a base met twice in one bucket (the tangent), a base and its
    negative in one bucket (the sum vanishes, then the bucket starts again), bases at infinity, empty buckets, buckets
    handed to the heavy path, 2^15 buckets spread over four waves"""
    from owshen_amd import api
    from oracle.c import binding as oc
    rnd = random.Random(300 + window)
    rng = np.random.default_rng(300 + window)
    n = 60
    ks = [rnd.randrange(1, fields.R) for _ in range(n)]
    pts = [G2.mul(G2_GEN, k) for k in ks]
    for i in range(0, 12, 2):
        pts[i + 1] = pts[i]                  # the same base twice
    for i in range(12, 24, 2):
        pts[i + 1] = G2.neg(pts[i])          # a base and its negative
    pts[30] = pts[41] = None                 # bases at infinity
    bases = np.frombuffer(b"".join(g2_to_bytes(q) for q in pts), dtype=np.uint8).reshape(-1, 128).copy()
    sc = _rand_fr_np(rng, 4, n)
    for i in range(0, 24, 2):
        sc[0, i + 1] = sc[0, i]              # equal scalars: the pair meets in every window's bucket
        sc[1, i + 1] = sc[1, i]
    sc[1, 24:] = 0
    sc[2] = 0
    sc[2, :, 0] = 1                          # every base in the bucket of digit 1
    sc[3, 5] = _tob([fields.R - 1])[0]
    sc[3, 40:] = 0
    if heavy:
        monkeypatch.setenv("OG_HEAVY", heavy)
    b = api.Bases(ectx, 2, bases, window, precomp)
    monkeypatch.setenv("OG_G2_AFFINE", "1")
    got = b.msm(sc)
    monkeypatch.setenv("OG_G2_AFFINE", "0")
    plain = b.msm(sc)
    b.close()
    assert got.tobytes() == plain.tobytes()
    for g in range(4):
        assert got[g].tobytes() == oc.msm_g2(bases, sc[g]).tobytes(), g


def test_emu_lone_msm_in_two_window_halves(ectx, monkeypatch):
    """og_msm_d's two-half form of a lone big G1 MSM over plain bases (even / odd windows on the context's two lanes, each the
    window-sharded rank path, joined by the Horner combine), forced at toy size together with the lone sort and the
    position-major pieces: the same point as the single launch set (OG_LONE_HALVES=0) and as the C restatement"""
    from owshen_amd import api
    from oracle.c import binding as oc
    n = 700
    rng = np.random.default_rng(22)
    ks = _rand_fr_np(rng, n)
    bases_np = oc.fixed_base_g1(np.frombuffer(g1_to_bytes(G1_GEN), dtype=np.uint8), ks)
    sc = _rand_fr_np(rng, n)
    sc[:30] = 0
    sc[30:150] = 0
    sc[30:150, 0] = 1
    sc[150] = _tob([fields.R - 1])[0]
    want = oc.msm_g1(bases_np, sc).tobytes()
    b = api.Bases(ectx, 1, bases_np, 16, False)
    monkeypatch.setenv("OG_LONE_MIN", "1")
    monkeypatch.setenv("OG_LONE_AVG", "0.0001")
    monkeypatch.setenv("OG_LONE_PIECES", "3")
    monkeypatch.setenv("OG_LONE_HALVES_MIN", "1")
    halves = b.msm(sc)[0].tobytes()
    monkeypatch.setenv("OG_LONE_HALVES", "0")
    single = b.msm(sc)[0].tobytes()
    assert halves == single == want
    # and a second call reuses both lanes' scratch
    monkeypatch.delenv("OG_LONE_HALVES")
    assert b.msm(sc)[0].tobytes() == want
