"""EdDSA batch verification on the CPU interpreter (tests/hipemu); cases in tests/eddsa_cases.py."""
import pytest


@pytest.fixture(scope="module")
def ectx():
    from tests import emu
    c = emu.Ctx()
    yield c
    c.close()


def test_emu_eddsa_batch(ectx):
    from tests import eddsa_cases
    eddsa_cases.case_eddsa_batch(ectx, n_valid=2)


def test_oracle_mimc7_eddsa_self_consistency():
    """the three reference tests' structure (babyjubjub/tests.rs:3-51) on the MiMC7 variant of sign / verify"""
    from oracle.py import babyjubjub as bj
    pk = bj.multiply(bj.BASE, 123456)
    sig = bj.sign_mimc7(123456, 2345, 123456)
    assert bj.verify_mimc7(pk, 123456, sig) and not bj.verify_mimc7(pk, 123457, sig)
    assert bj.decompress(bj.compress(pk)) == pk
