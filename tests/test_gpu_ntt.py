"""GPU parity: radix-2 Fr NTT and the H-polynomial pipeline through the C ABI,
bit-exact against the oracles, plus size-independent properties (8c-iii)."""
import random

import numpy as np
import pytest
import torch

from oracle.py import fields
from oracle.py import ntt as pntt

pytestmark = pytest.mark.gpu


def _rand_fr_np(rng, *shape):
    a = rng.integers(0, 256, (*shape, 32), dtype=np.uint8)
    a[..., 31] &= 0x1F  # < 2^253 < r
    return a


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 10, 11, 13, 17])   # 17: the headline size -- the full 11-pass tile schedule, k_ntt_block4's lazy bounds
def test_ntt_all_modes_vs_oracle(ctx, log_n):
    from owshen_amd import api
    from oracle.c import binding as oc
    n = 1 << log_n
    rng = np.random.default_rng(100 + log_n)
    x = _rand_fr_np(rng, n)
    if n >= 4:
        x[1] = np.frombuffer((fields.R - 1).to_bytes(32, "little"), dtype=np.uint8)
        x[2] = 0
    xd = ctx.to_device(x)
    for inverse in (False, True):
        for coset in (False, True):
            got = ctx.ntt(xd, inverse=inverse, coset=coset).cpu().numpy()
            want = oc.ntt(x, inverse=inverse, coset=coset)
            assert got.tobytes() == want.tobytes(), (log_n, inverse, coset)
    if log_n <= 5:  # the C oracle itself against the definition
        xi = api.bytes_to_ints(x)
        assert api.bytes_to_ints(ctx.ntt(xd).cpu().numpy()) == pntt.dft_naive(xi, fields.fr_root_of_unity(log_n))


def test_ntt_batched_roundtrip_2_17(ctx):
    """iNTT(NTT(x)) = x and coset variants at the BASELINE size 2^17, batch 3."""
    rng = np.random.default_rng(7)
    x = ctx.to_device(_rand_fr_np(rng, 3, 1 << 17))
    y = ctx.ntt(x)
    assert not torch.equal(x, y)
    assert torch.equal(ctx.ntt(y, inverse=True), x)
    assert torch.equal(ctx.ntt(ctx.ntt(x, coset=True), inverse=True, coset=True), x)
    # batch items are independent: item 1 alone gives the same answer
    assert torch.equal(ctx.ntt(x[1].contiguous()), y[1])


def test_ntt_evaluation_property_2_17(ctx):
    """NTT output i equals the polynomial evaluated at w^i (Horner on the host, 2 points)."""
    from owshen_amd import api
    rng = np.random.default_rng(8)
    log_n = 17
    n = 1 << log_n
    x = _rand_fr_np(rng, n)
    y = api.bytes_to_ints(ctx.ntt(ctx.to_device(x)).cpu().numpy())
    coeffs = api.bytes_to_ints(x)
    w = fields.fr_root_of_unity(log_n)
    for i in (1, 77777):
        pt = pow(w, i, fields.R)
        acc = 0
        for c in reversed(coeffs):
            acc = (acc * pt + c) % fields.R
        assert y[i] == acc


@pytest.mark.parametrize("log_d", [3, 10, 12, 17])
def test_h_poly_vs_oracle(ctx, log_d):
    from oracle.c import binding as oc
    d = 1 << log_d
    rng = np.random.default_rng(200 + log_d)
    a, b, c = (_rand_fr_np(rng, d) for _ in range(3))
    got = ctx.h_poly(ctx.to_device(a), ctx.to_device(b), ctx.to_device(c)).cpu().numpy()
    assert got.tobytes() == oc.h_poly(a, b, c).tobytes()


def test_h_poly_satisfied_has_zero_top_coefficient(ctx):
    """For c = a*b on the domain, (A*B - C) vanishes on it and h has degree <= d-2."""
    from owshen_amd import api
    log_d = 12
    d = 1 << log_d
    rng = np.random.default_rng(9)
    a, b = _rand_fr_np(rng, 2, d), _rand_fr_np(rng, 2, d)
    ad, bd = ctx.to_device(a), ctx.to_device(b)
    cd = ctx.field_op(0, "mul", ad.reshape(-1, 32), bd.reshape(-1, 32)).reshape(2, d, 32)
    h = ctx.h_poly(ad, bd, cd).cpu().numpy()
    assert not h[:, d - 1].any()
    assert h[:, : d - 1].any()
