"""GPU parity: batched append to the incremental MiMC7 commitment tree; cases in tests/tree_cases.py."""
import pytest

from tests import tree_cases as cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("depth,batches", [(3, [1, 1, 2, 3, 1]), (10, [1, 100, 411, 512]), (32, [1, 2, 1000, 3])])
def test_append(ctx, depth, batches):
    cases.case_append_matches_incremental_tree(ctx, depth, batches, seed=100 + depth)


@pytest.mark.parametrize("depth,batches", [(3, [1, 1, 2, 1, 3]), (10, [1, 16, 1, 7]), (32, [1, 1, 3, 16, 1])])
def test_host_append_equals_the_kernels(ctx, depth, batches):
    """og_set_host_chains: a handful of leaves appended on the host CPU (one leaf: 8.3 ms of level launches -> ~0.4 ms)"""
    cases.case_host_append_equals_the_kernels(ctx, depth, batches, seed=140 + depth)


def test_one_and_two_lanes_per_hash_agree(ctx_hooks, monkeypatch):
    cases.case_one_and_two_lanes_per_hash_agree(ctx_hooks, monkeypatch, n_hash=301, n_paths=5, depth=32, n_leaves=512, witness_depth=32)
