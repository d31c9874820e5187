"""Committed golden vectors (tests/golden/golden.json): the python and C oracles, the CPU-interpreted kernels, and
(-m gpu) the gfx950 binary all reproduce them."""
import pytest

from tests import golden_cases


def test_oracles_reproduce_golden():
    golden_cases.check_oracles()


def test_interpreted_kernels_reproduce_golden():
    from tests import emu
    c = emu.Ctx()
    try:
        golden_cases.check_device(c)
    finally:
        c.close()


@pytest.mark.gpu
def test_gpu_reproduces_golden(ctx):
    golden_cases.check_device(ctx)
