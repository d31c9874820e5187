"""GPU parity: batched BabyJubJub EdDSA verification with the MiMC7 sponge; cases in tests/eddsa_cases.py."""
import pytest

pytestmark = pytest.mark.gpu


def test_eddsa_batch(ctx):
    from tests import eddsa_cases
    eddsa_cases.case_eddsa_batch(ctx, n_valid=40, seed=7)
