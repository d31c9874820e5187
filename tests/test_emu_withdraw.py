"""Withdraw circuit + witness generator on the CPU interpreter (tests/hipemu); cases in tests/withdraw_cases.py."""
import pytest

from tests import withdraw_cases as cases


@pytest.fixture(scope="module")
def ectx():
    from tests import emu
    c = emu.Ctx()
    yield c
    c.close()


@pytest.mark.parametrize("depth,n_pad3,n_pad2", [(1, 0, 0), (3, 7, 130), (2, 0, 64), (32, 0, 0)])
def test_emu_r1cs_and_witness_match_spec(ectx, depth, n_pad3, n_pad2):
    cases.case_r1cs_and_witness_match_spec(ectx, depth, n_pad3, n_pad2, n_proofs=2 if depth == 32 else 3)


def test_emu_withdraw_end_to_end(ectx):
    cases.case_withdraw_end_to_end(ectx, 1, 5, 70)


def test_emu_dense_rows(ectx):
    cases.case_dense_rows(ectx, 1, 3, 66)


def test_emu_withdraw_end_to_end_dense(ectx):
    cases.case_withdraw_end_to_end(ectx, 1, 2, 5, dense=True)


@pytest.mark.parametrize("depth,n_pad3,n_pad2,dense", [(1, 0, 0, False), (2, 5, 130, True), (3, 0, 64, False)])
def test_emu_native_builder(ectx, depth, n_pad3, n_pad2, dense):
    cases.case_native_builder_equals_python_builder(ectx, depth, n_pad3, n_pad2, dense)
