"""The deposit statement on the CPU interpreter (tests/hipemu); cases in tests/deposit_cases.py."""
import pytest

from tests import deposit_cases as cases


@pytest.fixture(scope="module")
def ectx():
    from tests import emu
    c = emu.Ctx()
    yield c
    c.close()


def test_emu_deposit_r1cs_and_witness_match_spec(ectx):
    cases.case_r1cs_and_witness_match_spec(ectx, n=3)


def test_emu_deposit_end_to_end(ectx):
    cases.case_deposit_end_to_end(ectx, n=4)
