"""GPU parity: Groth16 key generation + batched proving through the C ABI, byte-identical
to the oracles for injected (r, s); shared cases in tests/groth16_cases.py."""
import pytest

from tests import groth16_cases as cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_constraints,n_pub", [(6, 1), (57, 3)])
def test_setup_matches_oracle_setup(ctx, n_constraints, n_pub):
    cases.case_setup_matches_oracle_setup(ctx, n_constraints, n_pub)


def test_prove_batch_matches_oracle_and_verifies(ctx):
    cases.case_prove_batch_matches_oracle_and_verifies(ctx)


def test_unsatisfied_witness_is_rejected(ctx):
    cases.case_unsatisfied_witness_is_rejected(ctx)


def test_noncanonical_witness_is_rejected(ctx):
    cases.case_noncanonical_witness_is_rejected(ctx)


def test_pk_load_rejects_malformed_blobs(ctx):
    cases.case_pk_load_rejects_malformed_blobs(ctx)


@pytest.mark.parametrize("n_constraints,n_proofs,sub_batch", [(600, 3, 2), (5000, 5, 2), (20000, 3, None)])
def test_medium_circuit_vs_c_oracle(ctx, n_constraints, n_proofs, sub_batch):
    cases.case_medium_circuit_vs_c_oracle(ctx, n_constraints, n_proofs, sub_batch)


def test_degenerate_circuits(ctx):
    cases.case_degenerate_circuits(ctx)


def test_random_shapes(ctx):
    cases.case_random_shapes(ctx, range(2000, 2030))
